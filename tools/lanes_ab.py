"""Small-frame configurations with and without lanes (independent branches of the forward on side streams), launch plans on:
frames/s of the BASELINE configs c1 (network_lite 256x256), c2 (network_lite 256x448, global off), c3 (network_base 540x960 -> 576x960).
  python tools/lanes_ab.py"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import pairs
pkg = importlib.import_module("atm-vfi_amd")
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
for name, variant, H, W, glob in (("c1", "lite", 256, 256, True), ("c2", "lite", 256, 448, False), ("c3", "base", 576, 960, True), ("c4", "base", 1088, 1920, True)):
    net = (pkg.NetworkBase if variant == "base" else pkg.NetworkLite)()
    net.load_state_dict(pkg.synthetic_state_dict(variant, seed=1))
    net.global_motion = glob
    net.to(dev).eval()
    frames = [[t.to(dev) for t in pairs.random_pair(1, H, W, seed=2000 + i)] for i in range(2)]
    res = {}
    for rep in range(2):
        for lanes in (False, True):
            net.use_lanes = lanes
            steps = 300 if variant == "lite" else (100 if H < 1000 else 20)
            for i in range(8):
                net(*frames[i & 1])
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(steps):
                net(*frames[i & 1])
            torch.cuda.synchronize()
            res[lanes] = max(res.get(lanes, 0.0), steps / (time.perf_counter() - t0))
    print(f"{name} {variant} {H}x{W} global {glob}: {res[False]:.1f} -> {res[True]:.1f} frames/s ({100 * (res[True] / res[False] - 1):+.1f} %)", flush=True)
    net.release_workspace()
    torch.cuda.empty_cache()
