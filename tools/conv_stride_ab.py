"""What the scattered source rows of a strided convolution cost the LDS-DMA GEMM's CONV mode: the same (M, N, K) launch once with stride 2
(16 source rows of a DMA piece 128 B apart) and once with stride 1 on a map of the output's size (contiguous KiB per piece), plus the plain
LINEAR launch of that shape.   python tools/conv_stride_ab.py"""
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


for name, (ho, wo, cin, cout) in {"down1.0 64 -> 64 s2 (1088x1920)": (544, 960, 64, 64), "down2.0 256 -> 128 s2": (272, 480, 256, 128),
                                   "down3.0 512 -> 256 s2": (136, 240, 512, 256), "encoder 48 -> 96 s2": (272, 480, 48, 96)}.items():
    w = ((torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) / (9 * cin) ** 0.5).to(dev)
    pw = ops.pack_weight(hip_ops.GEMM_CONV, w)
    res = {}
    for stride in (2, 1):
        h, wd = ho * stride, wo * stride
        x = (torch.rand(h * wd, cin, generator=g) * 2 - 1).to(dev)
        pl = hip_ops.Planes.alloc(h * wd, cin, dev)
        ops.split_planes(x, pl)
        sink = hip_ops.Planes.alloc(ho * wo, cout, dev)
        del x
        res[stride] = timed(lambda: ops.conv_planes(pl, 1, h, wd, pw, stride=stride, pad=1, sink=sink))
        del pl, sink
    m, k = ho * wo, 9 * cin
    xl = (torch.rand(m, k, generator=g) * 2 - 1).to(dev)
    pll = hip_ops.Planes.alloc(m, k, dev); ops.split_planes(xl, pll)
    wl = ops.pack_weight(1, ((torch.rand(cout, k, generator=g) * 2 - 1) / k ** 0.5).to(dev))
    y = torch.empty(m, cout, device=dev)
    lin = timed(lambda: ops.linear(pll, wl, y))
    fl = 2.0 * m * cout * k
    print(f"{name:34s} M{m} N{cout} K{k}: stride 2 {res[2]:7.1f} us ({fl / res[2] / 1e6:5.0f} TF/s)   stride 1 {res[1]:7.1f} us ({fl / res[1] / 1e6:5.0f})   "
          f"linear of that shape {lin:7.1f} us ({fl / lin / 1e6:5.0f})", flush=True)
    del xl, pll, y
