"""Same-process A/B of the two 3x3 kernels on the network's 1080p layer shapes: fp32-input row/half kernel (atmvfi_conv3x3_f16x3)
against the split-plane ping-pong kernel (atmvfi_conv3x3_planes).  Interleaved rounds, median of the per-round times."""
import importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
H = importlib.import_module("atm-vfi_amd.hip_ops")
if os.environ.get("ATMVFI_LIB"):          # A/B of two builds on one box: ATMVFI_LIB=tools/lib/libatmvfi_hip_base.so
    H.LIB_PATH = os.path.join(ROOT, os.environ["ATMVFI_LIB"])
    H.load_library.__defaults__ = (H.LIB_PATH,)
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
ops = H.HipOps(dev)
LAYERS = [  # N, H, W, Cin, Cout
    (1, 1088, 1920, 101, 101), (1, 544, 960, 197, 197), (1, 272, 480, 389, 389), (1, 1088, 1920, 116, 64), (1, 1088, 1920, 128, 64),
    (1, 136, 240, 776, 576), (1, 136, 240, 576, 576), (1, 68, 120, 1352, 768), (1, 68, 120, 768, 768),
    (2, 1088, 1920, 24, 24), (2, 544, 960, 48, 48), (2, 272, 480, 96, 96), (2, 136, 240, 192, 192), (1, 544, 960, 128, 64),
    (1, 272, 480, 128, 128), (1, 136, 240, 256, 256), (1, 1088, 1920, 64, 3),
]
only = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 and sys.argv[1] else None
wns = [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
g = torch.Generator().manual_seed(1)
r4 = lambda c: (c + 3) // 4 * 4
tot_old = tot_new = 0.0
for li, (n, h, w, cin, cout) in enumerate(LAYERS):
    if only is not None and li not in only:
        continue
    x = (torch.rand(n, h, w, r4(cin), generator=g) * 2 - 1).to(dev)[..., :cin]
    wt = ((torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) / np.sqrt(9 * cin)).to(dev)
    bias = torch.rand(cout, generator=g).to(dev)
    slope = (torch.rand(cout, generator=g) * 0.4).to(dev)
    pw = ops.pack_weight(H.GEMM_CONV, wt)
    y0 = torch.empty(n, h, w, r4(cout), device=dev)[..., :cout]
    y1 = torch.empty(n, h, w, r4(cout), device=dev)[..., :cout]
    xp = H.Planes.alloc(n * h * w, cin, dev)
    ops.split_planes(x.flatten(0, 2), xp)
    sink = H.Planes.alloc(n * h * w, cout, dev)
    variants = {"old": lambda: ops.conv(x, pw, y0, 1, 1, 1, bias, slope)}
    for wn in wns:
        variants[f"new{wn}"] = (lambda wn=wn: ops.conv3x3_planes(xp, n, h, w, pw, out=y1, bias=bias, prelu=slope, wn=wn))
        variants[f"newP{wn}"] = (lambda wn=wn: ops.conv3x3_planes(xp, n, h, w, pw, out=None, bias=bias, prelu=slope, planes=sink, wn=wn))
    for f in variants.values():
        f()
    torch.cuda.synchronize()
    same = torch.equal(y0, y1)
    times = {k: [] for k in variants}
    for rnd in range(7):
        for k, f in variants.items():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(3):
                f()
            e.record()
            torch.cuda.synchronize()
            times[k].append(s.elapsed_time(e) / 3)
    med = {k: float(np.median(v)) for k, v in times.items()}
    fl = 2.0 * n * h * w * cout * cin * 9
    best_new = min(v for k, v in med.items() if k.startswith("new") and not k.startswith("newP"))
    tot_old += med["old"]; tot_new += best_new
    print(f"[{li:2d}] N{n} {h}x{w} {cin}->{cout}: bit-identical {same} | " +
          " ".join(f"{k} {v:.3f} ms ({fl / v / 1e9:.0f} TF/s)" for k, v in med.items()), flush=True)
print(f"sum old {tot_old:.3f} ms, sum best-new {tot_new:.3f} ms")
