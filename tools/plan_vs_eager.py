#!/usr/bin/env python3
"""Same-process A/B at one configuration: launch plan vs direct launches vs HIP graph; reports wall ms per forward and the host time
spent inside ``net(a, b)`` (a GPU-bound forward shows host << wall).  usage: python tools/plan_vs_eager.py [c4] [steps]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import bench, pairs
pkg = importlib.import_module("atm-vfi_amd")
host_io = importlib.import_module("atm-vfi_amd.host_io")
cname = sys.argv[1] if len(sys.argv) > 1 else "c4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
variant, h, w, g_on, _ = bench.CONFIGS[cname]
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
net = (pkg.NetworkBase if variant == "base" else pkg.NetworkLite)()
net.load_state_dict(pkg.synthetic_state_dict(variant, seed=1), strict=True)
net.to(dev).eval()
net.global_motion = g_on
padder = host_io.InputPadder((1, 3, h, w), divisor=64)
frames = []
for i in range(4):
    a, b = pairs.random_pair(1, h, w, seed=1000 + i)
    a, b = padder.pad(a.to(dev), b.to(dev))
    frames.append((a.contiguous(), b.contiguous()))


def run(label):
    for i in range(5):
        net(*frames[i % 4])
    torch.cuda.synchronize()
    host = 0.0
    t0 = time.perf_counter()
    for i in range(steps):
        th = time.perf_counter()
        net(*frames[i % 4])
        host += time.perf_counter() - th
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"{cname} {label:18s} {1e3 * el / steps:8.3f} ms/forward  ({steps / el:7.2f} fps)   host inside net(): {1e3 * host / steps:7.3f} ms", flush=True)


for rep in range(2):
    net.enable_graphs(False); net.enable_plans(True); run("launch plan")
    net.enable_plans(False); run("direct launches")
    net.enable_graphs(True); run("HIP graph"); net.enable_graphs(False)
