#!/usr/bin/env python3
"""K independent forwards in flight (host_io.PairStreams) against the single-stream forward: frames/s per BASELINE configuration and K,
and bit-identity of every result.  Usage: python tools/inflight_ab.py [c1 c2 c3 c4] [--ks 1,2,3,4] [--threads]"""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import pairs  # noqa: E402

pkg = importlib.import_module("atm-vfi_amd")
host_io = importlib.import_module("atm-vfi_amd.host_io")
CFG = {"c1": ("lite", 256, 256, True, 400), "c2": ("lite", 256, 448, False, 400), "c3": ("base", 576, 960, True, 100),
       "c4": ("base", 1088, 1920, True, 30)}


def main():
    names = [a for a in sys.argv[1:] if a in CFG] or ["c1", "c2", "c3"]
    ks = [1, 2, 3, 4]
    graphs = "--graphs" in sys.argv[1:]      # every replica replays a captured HIP graph (static outputs per replica) instead of its launch plan
    for a in sys.argv[1:]:
        if a.startswith("--ks="):
            ks = [int(x) for x in a[5:].split(",")]
    torch.set_grad_enabled(False)
    dev = torch.device("cuda:0")
    nets = {}
    for name in names:
        v, h, w, g, steps = CFG[name]
        if v not in nets:
            net = (pkg.NetworkBase if v == "base" else pkg.NetworkLite)()
            net.load_state_dict(pkg.synthetic_state_dict(v, seed=1), strict=True)
            nets[v] = net.to(dev).eval()
        net = nets[v]
        net.global_motion = g
        frames = [tuple(t.to(dev) for t in pairs.random_pair(1, h, w, seed=2000 + i)) for i in range(4)]
        for i in range(6):
            net(*frames[i % 4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            net(*frames[i % 4])
        torch.cuda.synchronize()
        base = steps / (time.perf_counter() - t0)
        want = [net(*f)["I_t"].clone() for f in frames]
        line = f"{name} {v} {h}x{w} global {'on' if g else 'off'}: single stream {base:8.1f} frames/s |"
        for k, thr in [(k, t) for t in ((False,) if graphs else (False, True)) for k in ks]:
            ps = host_io.PairStreams(net, k, threads=thr)
            if graphs:
                for r in ps.replicas:
                    r.enable_graphs(True)
            ok = True
            for i, o in enumerate(ps.map(frames[i % 4] for i in range(4 * k + 4))):           # workspaces, plans
                ok = ok and torch.equal(o["I_t"], want[i % 4])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 0
            for o in ps.map((frames[i % 4] for i in range(steps)), wait_inputs=False, record_outputs=False):
                n += 1
            ps.synchronize()
            fps = n / (time.perf_counter() - t0)
            for i, o in enumerate(ps.map(frames[i % 4] for i in range(8))):
                ok = ok and torch.equal(o["I_t"], want[i % 4])
            line += f" K={k}{'t' if thr else ''}: {fps:7.1f}{'' if ok else ' DIFFERS'}"
            ps.release()
            del ps
            torch.cuda.empty_cache()
        print(line + "   (t = one issuing thread per stream; every result compared bit for bit with the single-stream forward)", flush=True)
        net.release_workspace()


if __name__ == "__main__":
    main()
