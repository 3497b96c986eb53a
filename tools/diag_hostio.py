"""Bisect FramePipeline's per-frame overhead: GPU time of each forward inside the pipeline, and the pipeline with pieces removed."""
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

def run(tag, depth=3, no_h2d=False, no_d2h=False, no_hostcopy=False, time_gpu=False):
    p = host_io.FramePipeline(net, H, W, depth=depth)
    evs = []
    if no_h2d or no_hostcopy:
        def up(slot, pair):
            if not no_hostcopy:
                np.copyto(slot["h_in_np"][0], pair[0]); np.copyto(slot["h_in_np"][1], pair[1])
            with torch.cuda.stream(p.copy_in):
                if not no_h2d: slot["d_in"].copy_(slot["h_in"], non_blocking=True)
                slot["in_ready"].record(p.copy_in)
        p._upload = up
    if no_d2h or time_gpu:
        def comp(slot):
            cur = torch.cuda.current_stream(dev)
            cur.wait_event(slot["in_ready"])
            if time_gpu:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record(cur)
            p.ops.frame_u8_to_f32(slot["d_in"][0], slot["f0"][0], p.pad_top, p.pad_left, p.bgr)
            p.ops.frame_u8_to_f32(slot["d_in"][1], slot["f1"][0], p.pad_top, p.pad_left, p.bgr)
            it = p.model.forward(slot["f0"], slot["f1"])["I_t"]
            p.ops.frame_f32_to_u8(it[0], slot["d_out"], p.pad_top, p.pad_left, p.bgr)
            if time_gpu:
                e1.record(cur); evs.append((e0, e1))
            slot["done"].record(cur)
            p.copy_out.wait_event(slot["done"])
            with torch.cuda.stream(p.copy_out):
                if not no_d2h: slot["h_out"].copy_(slot["d_out"], non_blocking=True)
                slot["out_ready"].record(p.copy_out)
        p._compute = comp
    list(p.run(pairs[:4])); evs.clear(); torch.cuda.synchronize(); t0 = time.perf_counter(); n = sum(1 for _ in p.run(pairs)); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    extra = ""
    if evs:
        g = sorted(a.elapsed_time(b) for a, b in evs)
        gaps = sorted(evs[i][1].elapsed_time(evs[i + 1][0]) for i in range(len(evs) - 1))
        extra = f"   GPU forward median {g[len(g)//2]:.2f} max {g[-1]:.2f}; gap between forwards median {gaps[len(gaps)//2]:.2f} max {gaps[-1]:.2f}"
    print(f"{tag:34s} {dt:7.2f} ms per frame{extra}", flush=True)

print("torch threads", torch.get_num_threads(), "cpus", len(os.sched_getaffinity(0)))
try: print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e: print("cpu.max n/a", e)
run("depth 3 as shipped")
run("depth 3 + GPU timing", time_gpu=True)
run("depth 3 no H2D", no_h2d=True)
run("depth 3 no D2H", no_d2h=True)
run("depth 3 no H2D no D2H", no_h2d=True, no_d2h=True)
run("depth 3 no host memcpy", no_hostcopy=True)
torch.set_num_threads(1)
run("depth 3, 1 torch thread")
run("depth 3, 1 thread + GPU timing", time_gpu=True)
run("depth 1, 1 thread + GPU timing", depth=1, time_gpu=True)
