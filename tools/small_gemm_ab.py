"""Small frames: the ping-pong GEMM (gemm_pp.hip) against gemm_duo.hip (128 x 128 tiles, two workgroups per CU, tile_wn = -2) for the whole forward."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import pairs
pkg = importlib.import_module("atm-vfi_amd")
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
for variant, h, w, g in (("lite", 256, 256, True), ("lite", 256, 448, False), ("base", 576, 960, True)):
    net = (pkg.NetworkBase if variant == "base" else pkg.NetworkLite)()
    net.load_state_dict(pkg.synthetic_state_dict(variant, seed=1))
    net.to(dev).eval()
    net.global_motion = g
    a, b = [t.to(dev) for t in pairs.random_pair(1, h, w, seed=3)]
    for wn in (-4, -2, 0, -4, -2, 0):
        net(a, b)
        net._ops_obj.gemm_tile_wn = wn
        net._plans.clear()
        for _ in range(10):
            net(a, b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            net(a, b)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 100
        print(f"{variant} {h}x{w} global {g}: tile_wn {wn:2d}: {1e3 * dt:.3f} ms = {1 / dt:.1f} frames/s", flush=True)
