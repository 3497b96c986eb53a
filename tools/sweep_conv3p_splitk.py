"""Split-K of atmvfi_conv3x3_planes on the under-filled long-K layer shapes of the small configurations: unsplit (auto width) against the
split launch at every width 1..8 (wn forced; the launcher picks the split count).  python tools/sweep_conv3p_splitk.py"""
import importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
H = importlib.import_module("atm-vfi_amd.hip_ops")
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
ops = H.HipOps(dev)
LAYERS = [(1, 16, 16, 712, 352, "c1 gmlp0"), (1, 16, 16, 352, 352, "c1 gmlp1"), (1, 32, 32, 456, 224, "c1 lmlp0"), (1, 32, 32, 224, 224, "c1 lmlp1"),
          (1, 32, 56, 456, 224, "c2 lmlp0"), (1, 32, 56, 224, 224, "c2 lmlp1"), (1, 36, 60, 1352, 768, "c3 gmlp0"), (1, 36, 60, 768, 768, "c3 gmlp1"),
          (1, 72, 120, 776, 576, "c3 lmlp0"), (2, 36, 60, 288, 288, "c3 last"), (1, 72, 120, 256, 256, "c3 down3.x")]
g = torch.Generator().manual_seed(1)


def timeit(f):
    f(); torch.cuda.synchronize()
    ts = []
    for rnd in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            f()
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 5)
    return 1e3 * float(np.median(ts))


for n, h, w, cin, cout, name in LAYERS:
    xp = H.Planes.alloc(n * h * w, cin, dev)
    xp.t.copy_((torch.rand(xp.t.shape, generator=g) - 0.5).half()); xp.t[:, :, xp.rows:] = 0
    wt = ((torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) / np.sqrt(9 * cin)).to(dev)
    bias = torch.rand(cout, generator=g).to(dev); slope = (torch.rand(cout, generator=g) * 0.4).to(dev)
    pw = ops.pack_weight(H.GEMM_CONV, wt)
    sink = H.Planes.alloc(n * h * w, cout, dev)
    need = ops.conv3x3_workspace_floats(n, h, w, cin, cout)
    ws = torch.empty(max(need, 8 * n * h * w * ((cout + 15) // 16 * 16)), device=dev)
    warm = timeit(lambda: ops.conv3x3_planes(xp, n, h, w, pw, out=None, bias=bias, prelu=slope, planes=sink))
    base = timeit(lambda: ops.conv3x3_planes(xp, n, h, w, pw, out=None, bias=bias, prelu=slope, planes=sink))
    auto = timeit(lambda: ops.conv3x3_planes(xp, n, h, w, pw, out=None, bias=bias, prelu=slope, planes=sink, workspace=ws))
    per = {k: timeit(lambda k=k: ops.conv3x3_planes(xp, n, h, w, pw, out=None, bias=bias, prelu=slope, planes=sink, wn=k, workspace=ws)) for k in range(1, 9)}
    print(f"{name:12s} N{n} {h}x{w} {cin}->{cout}: unsplit {base:6.1f} us | split auto {auto:6.1f} us (need {need}) | split by forced width " +
          " ".join(f"{k}:{v:.1f}" for k, v in per.items()), flush=True)
