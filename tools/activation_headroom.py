"""fp16-range headroom of the f16x3 engines (deviation 2): the largest |operand| every contraction layer sees
on the synthetic "stress" weights, as a fraction of the saturation point 65504 * (1 + 2^-10).  Diagnostic only (torch reductions
on the layer outputs); run on the GPU box: python tools/activation_headroom.py [H W [variant]]."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import pairs
pkg = importlib.import_module("atm-vfi_amd")
H = importlib.import_module("atm-vfi_amd.hip_ops")
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1088, 1920)
variant = sys.argv[3] if len(sys.argv) > 3 else "base"
net = (pkg.NetworkBase if variant == "base" else pkg.NetworkLite)()
net.load_state_dict(pkg.synthetic_state_dict(variant, seed=1))
net.to(dev).eval()
rows = []


def amax(t):
    if t is None:
        return None
    if isinstance(t, H.Planes):
        t = t.t[:, :, :t.rows].float()
        return float((t[0].abs() + t[1].abs() / 1024.0).max())
    return float(t.abs().max())


def wrap(ops, name, in_idx, out_idx, out_kw=()):
    fn = getattr(ops, name)

    def f(*a, **k):
        r = fn(*a, **k)
        xin = a[in_idx] if len(a) > in_idx else None
        xout = a[out_idx] if out_idx is not None and len(a) > out_idx else None
        cands = [xout] + [k.get(n) for n in out_kw]
        mo = max([m for m in (amax(c) for c in cands) if m is not None], default=float("nan"))
        wobj = next((x for x in a if isinstance(x, H.PackedWeight)), None)
        wmax = float(wobj.orig.abs().max()) if wobj is not None else float("nan")
        rows.append((name, tuple(wobj.orig.shape) if wobj is not None else (), amax(xin) if xin is not None else float("nan"), wmax, mo))
        return r
    setattr(ops, name, f)


for kind, seed in (("random", 3), ("smooth", 4)):
    a, b = (pairs.random_pair if kind == "random" else pairs.smooth_pair)(1, h, w, seed=seed)
    a, b = a.to(dev), b.to(dev)
    net(a, b)
    if kind == "random":
        ops = net._ops_obj
        wrap(ops, "conv", 0, 2, ("planes",))
        wrap(ops, "linear", 0, 2, ("sink",))
        wrap(ops, "deconv", 0, 2, ("sink",))
        wrap(ops, "conv3x3_planes", 0, None, ("out", "planes"))
    rows.clear()
    net(a, b)
    torch.cuda.synchronize()
    SAT = 65504.0 * (1 + 2.0 ** -10)
    worst = max(rows, key=lambda r: max(x for x in (r[2], r[4]) if x == x))
    big = max(max(x for x in (r[2], r[4]) if x == x) for r in rows)
    print(f"network_{variant} {h}x{w}, {kind} frames, stress weights seed 1: {len(rows)} contraction launches")
    print(f"  largest |activation| entering or leaving a contraction: {big:.3f} = {100 * big / SAT:.4f} % of the f16x3 saturation point "
          f"({SAT:.0f}); headroom x{SAT / big:.0f}")
    print(f"  largest |weight|: {max(r[3] for r in rows if r[3] == r[3]):.3f}")
    print("  ten largest layers (op, weight shape, max|in|, max|w|, max|out|):")
    for r in sorted(rows, key=lambda r: -max(x for x in (r[2], r[4]) if x == x))[:10]:
        print(f"    {r[0]:16s} {str(r[1]):24s} {r[2]:10.3f} {r[3]:8.3f} {r[4]:10.3f}")
