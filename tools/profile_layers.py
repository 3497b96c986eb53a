"""Per-launch timing table of one forward (HIP events on the launch stream): which layer shapes are slow."""
import importlib, sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import pairs
pkg = importlib.import_module("atm-vfi_amd")
if os.environ.get("ATMVFI_LIB"):          # A/B of two builds on one box: ATMVFI_LIB=tools/lib/libatmvfi_hip_base.so
    _h = importlib.import_module("atm-vfi_amd.hip_ops")
    _h.LIB_PATH = os.path.join(ROOT, os.environ["ATMVFI_LIB"])
    _h.load_library.__defaults__ = (_h.LIB_PATH,)
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1088, 1920)
variant = sys.argv[3] if len(sys.argv) > 3 else "base"
net = (pkg.NetworkBase if variant == "base" else pkg.NetworkLite)()
net.load_state_dict(pkg.synthetic_state_dict(variant, seed=1))
net.to(dev).eval()
a, b = pairs.random_pair(1, H, W, seed=3)
a, b = a.to(dev), b.to(dev)
for _ in range(2):
    net(a, b)
ops = net._ops_obj
best = None
for rep in range(3):
    ops.profile = []
    net(a, b)
    torch.cuda.synchronize()
    prof = [(n, m, s.elapsed_time(e)) for n, m, s, e in ops.profile]
    ops.profile = None
    if best is None:
        best = prof
    else:
        best = [(n, m, min(t, t2)) for (n, m, t), (_, _, t2) in zip(best, prof)]
tot = sum(t for _, _, t in best)
print(f"total {tot:.2f} ms over {len(best)} launches")
for n, m, t in best:
    if t < float(os.environ.get("ATMVFI_PROFILE_MIN_MS", "0.15")): continue
    tf = m.get("flops", 0) / (t * 1e-3) / 1e12
    gb = m.get("bytes", 0) / (t * 1e-3) / 1e9
    print(f"{n:18s} {m.get('shape', ''):28s} {t:8.3f} ms {tf:7.1f} TF/s {gb:8.1f} GB/s")
fam = {}
for n, m, t in best:
    f = fam.setdefault(n, [0, 0.0, 0.0, 0.0])
    f[0] += 1; f[1] += t; f[2] += m.get("flops", 0); f[3] += m.get("bytes", 0)
print("--- by family")
for n, (c, t, fl, by) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:18s} x{c:3d} {t:8.3f} ms {100 * t / tot:5.1f} %  {fl / (t * 1e-3) / 1e12:7.1f} TF/s {by / (t * 1e-3) / 1e9:8.1f} GB/s")
