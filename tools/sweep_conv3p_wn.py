"""Tile-width sweep of atmvfi_conv3x3_planes on the network's layer shapes (plane sink only, as the forward runs them): time per wn 1..8
against the launcher's own choice (wn = 0).  usage: python tools/sweep_conv3p_wn.py [c4|c3]"""
import importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
H = importlib.import_module("atm-vfi_amd.hip_ops")
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
ops = H.HipOps(dev)
cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
HH, WW = (1088, 1920) if cfg == "c4" else (576, 960)
LAYERS = [  # N, H/div, Cin, Cout, name
    (1, 1, 101, 101, "dec2"), (1, 2, 197, 197, "dec1"), (1, 4, 389, 389, "dec0"), (1, 1, 119, 64, "proj"), (1, 1, 128, 64, "head0"),
    (1, 8, 776, 576, "lmlp0"), (1, 8, 576, 576, "lmlp1"), (1, 16, 1352, 768, "gmlp0"), (1, 16, 768, 768, "gmlp1"),
    (2, 2, 48, 48, "e1"), (2, 4, 96, 96, "e2"), (2, 8, 192, 192, "e3"), (2, 16, 288, 288, "last"), (1, 2, 128, 64, "up2.1"),
    (1, 4, 128, 128, "down2.1/up1.1"), (1, 8, 256, 256, "down3.x"),
]
g = torch.Generator().manual_seed(1)
for n, div, cin, cout, name in LAYERS:
    h, w = HH // div, WW // div
    xp = H.Planes.alloc(n * h * w, cin, dev)
    xp.t.copy_(((torch.rand(xp.t.shape, generator=g) - 0.5)).half())
    xp.t[:, :, xp.rows:] = 0
    wt = ((torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) / np.sqrt(9 * cin)).to(dev)
    bias = torch.rand(cout, generator=g).to(dev)
    slope = (torch.rand(cout, generator=g) * 0.4).to(dev)
    pw = ops.pack_weight(H.GEMM_CONV, wt)
    sink = H.Planes.alloc(n * h * w, cout, dev)
    res = {}
    ntile = (cout + 15) // 16
    wns = [0] + [k for k in range(1, 9)]
    for wn in wns:
        f = lambda: ops.conv3x3_planes(xp, n, h, w, pw, out=None, bias=bias, prelu=slope, planes=sink, wn=wn)
        f(); torch.cuda.synchronize()
        ts = []
        for rnd in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(3):
                f()
            e.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / 3)
        res[wn] = float(np.median(ts))
    fl = 2.0 * n * h * w * cout * cin * 9
    best = min((v, k) for k, v in res.items() if k)
    print(f"{name:14s} N{n} {h}x{w} {cin}->{cout} ({ntile} n-tiles): auto {res[0]:.3f} ms ({fl / res[0] / 1e9:.0f} TF/s) | best wn={best[1]} {best[0]:.3f} ms ({100 * (res[0] / best[0] - 1):+.1f} %) | " +
          " ".join(f"{k}:{v:.3f}" for k, v in res.items() if k), flush=True)
