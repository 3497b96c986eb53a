"""Where the PCIe-inclusive frame time goes: resident forward / + pre-post kernels / + transfers sequential / overlapped."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
pkg = importlib.import_module("atm-vfi_amd")
host_io = importlib.import_module("atm-vfi_amd.host_io")
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
net = pkg.NetworkBase(); net.load_state_dict(pkg.synthetic_state_dict("base", seed=1)); net.to(dev).eval()
H, W, N = 1080, 1920, 30
rng = np.random.default_rng(0)
u8 = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(3)]
pairs = [(u8[i % 3], u8[(i + 1) % 3]) for i in range(N)]
pipe = host_io.FramePipeline(net, H, W, depth=2)
s = pipe.slots[0]
pipe._upload(s, pairs[0]); torch.cuda.synchronize()
def t(fn, n=N):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def resident():
    net(s["f0"], s["f1"])
def with_kernels():
    pipe.ops.frame_u8_to_f32(s["d_in"][0], s["f0"][0], pipe.pad_top, pipe.pad_left, True)
    pipe.ops.frame_u8_to_f32(s["d_in"][1], s["f1"][0], pipe.pad_top, pipe.pad_left, True)
    it = net(s["f0"], s["f1"])["I_t"]
    pipe.ops.frame_f32_to_u8(it[0], s["d_out"], pipe.pad_top, pipe.pad_left, True)
print(f"resident forward            {t(resident):7.2f} ms")
print(f"+ pre/post kernels          {t(with_kernels):7.2f} ms")
t0 = time.perf_counter(); [np.copyto(s["h_in_np"][0], u8[0]) for _ in range(20)]; print(f"host memcpy 6.2 MB -> pinned {(time.perf_counter()-t0)/20*1e3:7.2f} ms")
t0 = time.perf_counter(); [s["h_out"].numpy().copy() for _ in range(20)]; print(f"host copy of the result      {(time.perf_counter()-t0)/20*1e3:7.2f} ms")
t0 = time.perf_counter()
for _ in range(20): s["d_in"].copy_(s["h_in"], non_blocking=True); torch.cuda.synchronize()
print(f"H2D 12.4 MB (sync each)      {(time.perf_counter()-t0)/20*1e3:7.2f} ms")
t0 = time.perf_counter(); net(s["f0"], s["f1"]); print(f"CPU time to enqueue a forward {(time.perf_counter()-t0)*1e3:6.2f} ms"); torch.cuda.synchronize()
for depth in (1, 2, 3):
    p = host_io.FramePipeline(net, H, W, depth=depth)
    list(p.run(pairs[:3])); torch.cuda.synchronize(); t0 = time.perf_counter(); n = sum(1 for _ in p.run(pairs)); torch.cuda.synchronize()
    print(f"pipeline depth {depth}: {(time.perf_counter()-t0)/n*1e3:7.2f} ms per frame")
