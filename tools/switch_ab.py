#!/usr/bin/env python3
"""Same-process A/B of one host-side switch of Network (an attribute, e.g. use_splitk) on BASELINE.json's configurations: interleaved
rounds, launch plans on.   python tools/switch_ab.py use_splitk c1,c2,c3 [steps]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import bench, pairs
pkg = importlib.import_module("atm-vfi_amd")
host_io = importlib.import_module("atm-vfi_amd.host_io")
attr = sys.argv[1]
cfgs = sys.argv[2].split(",") if len(sys.argv) > 2 else ["c1", "c2", "c3"]
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
for cname in cfgs:
    variant, h, w, g_on, _ = bench.CONFIGS[cname]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else {"c1": 300, "c2": 300, "c3": 60, "c4": 20, "c5": 6}[cname]
    net = (pkg.NetworkBase if variant == "base" else pkg.NetworkLite)()
    net.load_state_dict(pkg.synthetic_state_dict(variant, seed=1), strict=True)
    net.to(dev).eval()
    net.global_motion = g_on
    padder = host_io.InputPadder((1, 3, h, w), divisor=64)
    a, b = pairs.random_pair(1, h, w, seed=1000)
    a, b = [t.contiguous() for t in padder.pad(a.to(dev), b.to(dev))]
    res = {True: [], False: []}
    for rnd in range(3):
        for val in (True, False):
            setattr(net, attr, val)
            for i in range(6):
                net(a, b)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                net(a, b)
            torch.cuda.synchronize()
            res[val].append((time.perf_counter() - t0) / steps)
    on, off = min(res[True]), min(res[False])
    print(f"{cname}: {attr}=True {1e3 * on:.4f} ms ({1 / on:.1f} fps)   {attr}=False {1e3 * off:.4f} ms ({1 / off:.1f} fps)   on/off time {on / off:.3f}", flush=True)
    net.release_workspace(); del net; torch.cuda.empty_cache()
