"""Determinism soak: forwards of several shapes interleaved for a minute with launch plans on; every output of a shape must be bit-identical
to its first occurrence (fresh tensors every time, workspaces evicted and rebuilt on the way).   python tools/soak.py [seconds]"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import pairs
pkg = importlib.import_module("atm-vfi_amd")
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
nets = {}
for v, cls in (("lite", pkg.NetworkLite), ("base", pkg.NetworkBase)):
    n = cls(); n.load_state_dict(pkg.synthetic_state_dict(v, seed=1)); nets[v] = n.to(dev).eval()
shapes = [("lite", 1, 256, 256, True), ("lite", 2, 256, 448, False), ("base", 1, 576, 960, True), ("base", 1, 1088, 1920, True),
          ("base", 2, 320, 512, True), ("lite", 1, 128, 192, True)]
inputs = {s: [t.to(dev) for t in pairs.smooth_pair(s[1], s[2], s[3], seed=7 + i)] for i, s in enumerate(shapes)}
first, count = {}, {s: 0 for s in shapes}
t0 = time.time(); it = 0
while time.time() - t0 < budget:
    s = shapes[(it * 7 + it // 5) % len(shapes)]; it += 1
    net = nets[s[0]]; net.global_motion = s[4]
    out = net(*inputs[s])
    sig = tuple(float(out[k].double().sum()) for k in ("I_t", "opt_flow_0", "occ_mask1")) + (float(out["im_t_list"][-1].double().sum()),)
    if s not in first: first[s] = (sig, out["I_t"].clone())
    else:
        assert sig == first[s][0] and torch.equal(out["I_t"], first[s][1]), f"shape {s} changed at iteration {it}: {sig} vs {first[s][0]}"
    count[s] += 1
    if it % 200 == 0: print(f"{it} forwards, {time.time() - t0:.0f} s", flush=True)
torch.cuda.synchronize()
print(f"{it} forwards in {time.time() - t0:.0f} s, every shape bit-identical to its first output:", {f"{s[0]} {s[1]}x{s[2]}x{s[3]}": c for s, c in count.items()},
      f"plans: lite {len(nets['lite']._plans)}, base {len(nets['base']._plans)}; peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
