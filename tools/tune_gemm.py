"""Same-box sweep of the f16x3 GEMM engine's tile width on the 1080p network_base deconv / strided-conv layers."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
hip_ops = importlib.import_module("atm-vfi_amd.hip_ops")
dev = torch.device("cuda:0")
ops = hip_ops.HipOps(dev)
g = torch.Generator().manual_seed(0)
r4 = lambda c: (c + 3) // 4 * 4
def timed(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def sweep(tag, run, nt):
    res, t_auto = {}, 1e9
    for _ in range(30): run()
    for rep in range(2):
        for wn in range(1, 9):
            if wn > nt: continue
            ops.gemm_tile_wn = wn
            res[wn] = min(res.get(wn, 1e9), timed(run))
        ops.gemm_tile_wn = 0
        t_auto = min(t_auto, timed(run))
    best = min(res, key=res.get)
    print(f"{tag}: auto {t_auto:.3f} ms | best wn{best} {res[best]:.3f} ({100 * (t_auto / res[best] - 1):+.1f} %) | " +
          " ".join(f"{wn}:{t:.3f}" for wn, t in sorted(res.items())), flush=True)
for (N, H, W, cin, cout, inpr) in [(1, 136, 240, 773, 389, False), (1, 272, 480, 389, 197, True), (1, 544, 960, 197, 101, True),
                                   (1, 136, 240, 256, 128, False), (1, 272, 480, 256, 128, False), (1, 544, 960, 128, 64, False)]:
    x = (torch.rand(N, H, W, r4(cin), generator=g) * 2 - 1).to(dev)[..., :cin]
    wt = ((torch.rand(cin, cout, 2, 2, generator=g) * 2 - 1) / cin ** 0.5).to(dev)
    b = torch.zeros(cout, device=dev); ipr = torch.full(((cin + 31) // 32 * 32,), 0.25, device=dev)
    pw = ops.pack_weight(hip_ops.GEMM_DECONV, wt)
    y = torch.empty(N, 2 * H, 2 * W, r4(cout), device=dev)[..., :cout]
    sweep(f"deconv {H}x{W} {cin}->{cout}", lambda: ops.deconv(x, pw, y, bias=b, prelu=b, in_prelu=ipr if inpr else None), (4 * cout + 15) // 16)
for (N, H, W, cin, cout, k, s_) in [(1, 544, 960, 256, 128, 3, 2), (1, 272, 480, 512, 256, 3, 2), (1, 1088, 1920, 64, 64, 3, 2), (2, 1088, 1920, 24, 48, 3, 2),
                                    (2, 544, 960, 48, 96, 3, 2), (2, 272, 480, 96, 192, 3, 2)]:
    x = (torch.rand(N, H, W, r4(cin), generator=g) * 2 - 1).to(dev)[..., :cin]
    wt = ((torch.rand(cout, cin, k, k, generator=g) * 2 - 1) / (cin * k * k) ** 0.5).to(dev)
    b = torch.zeros(cout, device=dev)
    pw = ops.pack_weight(hip_ops.GEMM_CONV, wt)
    y = torch.empty(N, H // s_, W // s_, r4(cout), device=dev)[..., :cout]
    sweep(f"conv s{s_} {H}x{W} {cin}->{cout}", lambda: ops.conv(x, pw, y, stride=s_, pad=1, bias=b, prelu=b), (cout + 15) // 16)
