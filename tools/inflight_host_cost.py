#!/usr/bin/env python3
"""Where the host time of a planned forward goes (c1/c2): enqueue-only rate on one stream, PairStreams, and worker threads."""
import importlib, os, sys, time, threading, queue
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import pairs
pkg = importlib.import_module("atm-vfi_amd")
host_io = importlib.import_module("atm-vfi_amd.host_io")
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "c1"
v, h, w, g = {"c1": ("lite", 256, 256, True), "c2": ("lite", 256, 448, False), "c3": ("base", 576, 960, True)}[name]
net = (pkg.NetworkBase if v == "base" else pkg.NetworkLite)()
net.load_state_dict(pkg.synthetic_state_dict(v, seed=1), strict=True)
net.to(dev).eval(); net.global_motion = g
frames = [tuple(t.to(dev) for t in pairs.random_pair(1, h, w, seed=2000 + i)) for i in range(4)]
for i in range(6): net(*frames[i % 4])
torch.cuda.synchronize()
N = 300
t0 = time.perf_counter()
for i in range(N): net(*frames[i % 4])
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"{name}: single stream: enqueue {1e6*(t1-t0)/N:.0f} us/forward, total {1e6*(t2-t0)/N:.0f} us/forward")
# the pieces of forward()
ops = net._ops_obj
t0 = time.perf_counter()
for i in range(N): net._prepare(ops)
print(f"  _prepare {1e6*(time.perf_counter()-t0)/N:.0f} us", end="")
t0 = time.perf_counter()
for i in range(N): net._mode_key(ops, *frames[0])
print(f"  _mode_key {1e6*(time.perf_counter()-t0)/N:.0f} us", end="")
key = net._mode_key(ops, *frames[0]); ent = net._plans[key]
a, b = frames[0]
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(N): ent.run((a, b), ops.device, ops._stream())
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"  plan.run enqueue {1e6*(t1-t0)/N:.0f} us")
# threads: K workers, each its own replica + stream
for K in (2, 3, 4, 6):
    reps = [net.replica() for _ in range(K)]
    streams = [torch.cuda.Stream(dev) for _ in range(K)]
    for r, st in zip(reps, streams):
        with torch.cuda.stream(st):
            for i in range(4): r(*frames[i % 4])
    torch.cuda.synchronize()
    qs = [queue.Queue() for _ in range(K)]
    done = queue.Queue()
    def worker(i):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(streams[i]):
            while True:
                item = qs[i].get()
                if item is None: break
                out = reps[i](*item)
                done.put(out["I_t"])
    th = [threading.Thread(target=worker, args=(i,), daemon=True) for i in range(K)]
    for t in th: t.start()
    M = 600
    t0 = time.perf_counter()
    for i in range(M): qs[i % K].put(frames[i % 4])
    for i in range(M): done.get()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    for q in qs: q.put(None)
    for t in th: t.join()
    print(f"  threads K={K}: {M/(t2-t0):.0f} frames/s (enqueue done at {1e3*(t1-t0):.0f} ms of {1e3*(t2-t0):.0f})")
    for r in reps: r.release_workspace()
    del reps; torch.cuda.empty_cache()
