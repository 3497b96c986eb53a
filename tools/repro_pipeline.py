"""Stand-alone replay of tests/test_gpu_e2e.py::test_frame_pipeline_matches_sequential (diagnostic)."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
pkg = importlib.import_module("atm-vfi_amd")
host_io = importlib.import_module("atm-vfi_amd.host_io")
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
net = pkg.NetworkLite()
net.load_state_dict(pkg.synthetic_state_dict("lite", seed=1), strict=True)
net.to(dev).eval()
rng = np.random.default_rng(5)
frames = [rng.integers(0, 256, (100, 150, 3), dtype=np.uint8) for _ in range(6)]
pairs_ = list(zip(frames[:-1], frames[1:]))
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    seq = [host_io.inference_2frame(a, b, net, isBGR=True) for a, b in pairs_]
    pipe = host_io.FramePipeline(net, 100, 150, isBGR=True, divisor=64, depth=2)
    got = list(pipe.run(pairs_))
    ok = len(got) == len(seq) and all(np.array_equal(g, s) for g, s in zip(got, seq))
    print("rep", rep, "ok", ok, flush=True)
print("done")
