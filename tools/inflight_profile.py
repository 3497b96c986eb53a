#!/usr/bin/env python3
"""Host-side profile (cProfile) of planned forwards with 4 streams in flight at 256x256 lite: where the issuing thread's time goes."""
import cProfile, importlib, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import pairs
pkg = importlib.import_module("atm-vfi_amd")
host_io = importlib.import_module("atm-vfi_amd.host_io")
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
net = pkg.NetworkLite(); net.load_state_dict(pkg.synthetic_state_dict("lite", seed=1), strict=True); net.to(dev).eval()
frames = [tuple(t.to(dev) for t in pairs.random_pair(1, 256, 256, seed=2000 + i)) for i in range(4)]
ps = host_io.PairStreams(net, 4)
list(ps.map(frames[i % 4] for i in range(24)))
ps.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
n = sum(1 for _ in ps.map((frames[i % 4] for i in range(800)), wait_inputs=False, record_outputs=False))
ps.synchronize()
pr.disable()
el = time.perf_counter() - t0
print(f"{n / el:.0f} frames/s under cProfile; {1e6 * el / n:.0f} us per forward")
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
