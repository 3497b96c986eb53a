"""What a launch costs: trivial kernels (an 8x8 -> 16x16 resize) issued directly through the ctypes binding, and replayed from a HIP graph.
MI355X: 8.6 us each when issued from Python (host-bound), 1.9 us per kernel from a graph -- so the ~10 us per launch of a network_lite 256x256
forward (116 launches, 1.16-1.2 ms, launch plans = graph rate) are the kernels' own durations, not launch overhead.   python tools/launch_floor.py"""
import importlib, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
hip_ops = importlib.import_module("atm-vfi_amd.hip_ops")
dev = torch.device("cuda:0")
ops = hip_ops.HipOps(dev)
a = torch.rand(1, 2, 8, 8, device=dev); b = torch.empty(1, 2, 16, 16, device=dev)
for n in (200, 2000):
    for _ in range(50): ops.resize(a, b, 2.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): ops.resize(a, b, 2.0)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{n} trivial launches: host issue {1e6*(t1-t0)/n:.2f} us each, complete {1e6*(t2-t0)/n:.2f} us each")
# a graph of 200 trivial kernels
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): ops.resize(a, b, 2.0)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(200): ops.resize(a, b, 2.0)
torch.cuda.synchronize()
for _ in range(3): g.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): g.replay()
torch.cuda.synchronize(); print(f"graph of 200 trivial kernels: {1e6*(time.perf_counter()-t0)/2000:.2f} us per kernel")
