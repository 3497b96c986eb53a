"""Fixed cost against per-k-step cost of the two contraction kernels at small-frame sizes: the same launch with K growing, timed inside a
captured HIP graph of 50 identical launches (device time per launch, no host cost).   python tools/small_kernel_floor.py"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
hip_ops = importlib.import_module("atm-vfi_amd.hip_ops")
dev = torch.device("cuda:0")
ops = hip_ops.HipOps(dev)
g = torch.Generator().manual_seed(1)


def graph_time(fn, n=50):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(n):
                fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        gr.replay()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / n * 1e6)
    return best


print("plane-input GEMM (linear), M = 2048 rows, N = 224 columns: us per launch by K")
for K in (32, 64, 128, 224, 448, 896):
    x = torch.rand(2048, K, generator=g).to(dev) - 0.5
    w = (torch.rand(224, K, generator=g).to(dev) - 0.5) / K ** 0.5
    pw = ops.pack_weight(hip_ops.GEMM_LINEAR, w)
    xp = hip_ops.Planes.alloc(2048, K, dev); ops.split_planes(x, xp)
    out = torch.empty(2048, 224, device=dev)
    print(f"  K {K:4d} ({(K + 31) // 32:2d} k-steps): {graph_time(lambda: ops.linear(xp, pw, out)):6.2f} us")
print("3x3 plane conv, 32 x 32 pixels (4 tiles), Cout = 224: us per launch by Cin (split-K off)")
for cin in (32, 64, 128, 256, 456):
    x = torch.rand(1, 32, 32, cin, generator=g).to(dev) - 0.5
    w = (torch.rand(224, cin, 3, 3, generator=g).to(dev) - 0.5) / (3 * cin ** 0.5)
    pw = ops.pack_weight(hip_ops.GEMM_CONV, w)
    xp = hip_ops.Planes.alloc(1024, cin, dev); ops.split_planes(x.flatten(0, 2), xp)
    sink = hip_ops.Planes.alloc(1024, 224, dev)
    print(f"  Cin {cin:4d} ({9 * ((cin + 31) // 32):3d} k-steps): {graph_time(lambda: ops.conv3x3_planes(xp, 1, 32, 32, pw, planes=sink)):6.2f} us")
print("pointwise floor: resize 8x8 -> 16x16:", f"{graph_time(lambda: ops.resize(a, b, 2.0)):.2f} us" if (a := torch.rand(1, 2, 8, 8, device=dev)) is not None and (b := torch.empty(1, 2, 16, 16, device=dev)) is not None else "")
x = torch.rand(2048, 224, device=dev); o = torch.empty(2048, 224, device=dev); wln = torch.ones(224, device=dev); bln = torch.zeros(224, device=dev)
print("layernorm 2048 x 224:", f"{graph_time(lambda: ops.layernorm(x, o, wln, bln)):.2f} us")
