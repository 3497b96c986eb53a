"""The plane-input GEMM launches of at most 64 columns under each engine: gemm_pp.hip (tile_wn -3), gemm_duo.hip with 128-column tiles (-2)
and with 64-column tiles (-4), and the automatic choice (0).   python tools/narrow_ab.py"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
hip_ops = importlib.import_module("atm-vfi_amd.hip_ops")
dev = torch.device("cuda:0")
ops = hip_ops.HipOps(dev)
g = torch.Generator().manual_seed(0)


def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


cases = {"down1.0 64 -> 64 s2 (1088x1920)": (1088, 1920, 64, 64, 2, 1), "fusion 48 -> 48 s4 (544x960)": (544, 960, 48, 48, 4, 1),
         "fusion 48 -> 48 s4 d2": (544, 960, 48, 48, 4, 2), "64 -> 64 s2 (544x960)": (544, 960, 64, 64, 2, 1),
         "128 -> 64 s2 (272x480, K 1152)": (272, 480, 128, 64, 2, 1), "256 -> 64 s1 (136x240, K 2304)": (136, 240, 256, 64, 1, 1),
         "lite 32 -> 32 s2 (256x448)": (256, 448, 32, 32, 2, 1)}
for name, (h, wd, cin, cout, stride, dil) in cases.items():
    w = ((torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) / (9 * cin) ** 0.5).to(dev)
    pw = ops.pack_weight(hip_ops.GEMM_CONV, w)
    x = (torch.rand(h * wd, cin, generator=g) * 2 - 1).to(dev)
    pl = hip_ops.Planes.alloc(h * wd, cin, dev)
    ops.split_planes(x, pl)
    pad = dil
    ho, wo = (h + 2 * pad - 2 * dil - 1) // stride + 1, (wd + 2 * pad - 2 * dil - 1) // stride + 1
    sink = hip_ops.Planes.alloc(ho * wo, cout, dev)
    res, ref = {}, None
    for eng in (-3, -2, -4, 0):
        ops.gemm_tile_wn = eng
        sink.t.zero_()
        res[eng] = timed(lambda: ops.conv_planes(pl, 1, h, wd, pw, stride=stride, pad=pad, dil=dil, sink=sink))
        cur = sink.t.clone()
        assert ref is None or torch.equal(cur, ref), "engines differ"
        ref = cur
    fl = 2.0 * ho * wo * cout * 9 * cin
    print(f"{name:34s} M{ho * wo} N{cout} K{9 * cin}: pp {res[-3]:7.1f} us  duo128 {res[-2]:7.1f}  duo64 {res[-4]:7.1f}  auto {res[0]:7.1f} us ({fl / res[0] / 1e6:4.0f} TF/s)", flush=True)
    del x, pl, sink
