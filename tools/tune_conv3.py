"""Same-box sweep of the 3x3 kernel's (schedule, n-tiles) choices on the 1080p network_base layers, against the cost model's pick."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
hip_ops = importlib.import_module("atm-vfi_amd.hip_ops")
dev = torch.device("cuda:0")
ops = hip_ops.HipOps(dev)
g = torch.Generator().manual_seed(0)
r4 = lambda c: (c + 3) // 4 * 4
LAYERS = [(2, 1088, 1920, 24, 24), (2, 544, 960, 48, 48), (2, 136, 240, 192, 192), (1, 68, 120, 1352, 768), (1, 68, 120, 768, 768),
          (1, 136, 240, 776, 576), (1, 136, 240, 576, 576), (1, 272, 480, 389, 389), (1, 544, 960, 197, 197), (1, 1088, 1920, 101, 101),
          (1, 1088, 1920, 116, 64), (1, 136, 240, 256, 256), (1, 544, 960, 128, 64), (1, 1088, 1920, 128, 64), (1, 1088, 1920, 64, 3)]
def timed(fn, n=12):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (N, H, W, cin, cout) in LAYERS:
    x = (torch.rand(N, H, W, r4(cin), generator=g) * 2 - 1).to(dev)[..., :cin]
    w = ((torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) / (3 * cin ** 0.5)).to(dev)
    b = torch.zeros(cout, device=dev)
    pw = ops.pack_weight(0, w)
    y = torch.empty(N, H, W, r4(cout), device=dev)[..., :cout]
    run = lambda: ops.conv(x, pw, y, 1, 1, 1, b, b)
    nt = (cout + 15) // 16
    res, t_auto = {}, 1e9
    for _ in range(30): run()          # clocks up after the host-side set-up
    for rep in range(2):               # two interleaved passes, minimum of each
        for sched in (0, 1):
            for wn in range(1, 9):
                if wn > nt and wn != 1: continue
                ops.conv3_instance = (sched, wn)
                res[(sched, wn)] = min(res.get((sched, wn), 1e9), timed(run, 8))
        ops.conv3_instance = None
        t_auto = min(t_auto, timed(run, 8))
    best = min(res, key=res.get)
    row = " ".join(f"{'rh'[s]}{wn}:{t:.3f}" for (s, wn), t in sorted(res.items(), key=lambda kv: kv[1])[:5])
    print(f"N{N} {H}x{W} {cin}->{cout}: auto {t_auto:.3f} ms | best {'row' if best[0] == 0 else 'half'} wn{best[1]} {res[best]:.3f} ({100 * (t_auto / res[best] - 1):+.1f} %) | {row}", flush=True)
