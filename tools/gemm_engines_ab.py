"""Per-launch A/B of the three plane-input GEMM schedules inside a forward -- gemm_pp.hip (tile_wn -3), gemm_duo.hip 128 x 128 (-2) and
128 x 64 (-4) -- against the launcher's own choice (0): the data behind launch_gemm_split's rule.
    python tools/gemm_engines_ab.py [H W [variant [global(0/1)]]]"""
import importlib, os, re, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import pairs
pkg = importlib.import_module("atm-vfi_amd")
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1088, 1920)
variant = sys.argv[3] if len(sys.argv) > 3 else "base"
net = (pkg.NetworkBase if variant == "base" else pkg.NetworkLite)()
net.load_state_dict(pkg.synthetic_state_dict(variant, seed=1))
net.to(dev).eval()
net.global_motion = (sys.argv[4] != "0") if len(sys.argv) > 4 else True
a, b = [t.to(dev) for t in pairs.random_pair(1, H, W, seed=3)]
net(a, b)
ops = net._ops_obj
res = {}
ENG = (-3, -2, -4, 0)
for wn in ENG:
    ops.gemm_tile_wn = wn
    best = None
    for rep in range(5):
        ops.profile = []
        net(a, b)
        torch.cuda.synchronize()
        t = [(n, m.get("shape", ""), s.elapsed_time(e)) for n, m, s, e in ops.profile if n.endswith("_split")]
        ops.profile = None
        best = t if best is None else [(n, sh, min(x, y[2])) for (n, sh, x), y in zip(t, best)]
    res[wn] = best
tot = {k: 0.0 for k in ENG}
tot["best"] = 0.0
print(f"network_{variant} {H}x{W} global {net.global_motion}: {len(res[-3])} plane-input GEMM launches (us; * = fastest)")
names = {-3: "pp", -2: "duo128", -4: "duo64", 0: "auto"}
for i, (n, sh, _) in enumerate(res[-3]):
    ts = {k: res[k][i][2] for k in ENG}
    m_, n_, k_ = (int(x) for x in re.match(r"M(\d+) N(\d+) K(\d+)", sh).groups())
    for k in ENG:
        tot[k] += ts[k]
    bk = min((-3, -2, -4), key=lambda k: ts[k])
    tot["best"] += ts[bk]
    print(f"  {n:16s} {sh:28s} t256 {-(-m_ // 256) * -(-n_ // 128):5d} t128x64 {-(-m_ // 128) * -(-n_ // 64):6d} nk {-(-k_ // 32):4d}  " +
          "  ".join(f"{names[k]} {ts[k] * 1e3:7.1f}{'*' if k == bk else ' '}" for k in ENG) + (f"   <-- auto loses {100 * (ts[0] / ts[bk] - 1):.0f} %" if ts[0] > 1.04 * ts[bk] and ts[0] - ts[bk] > 0.0015 else ""))
print("sum: " + ", ".join(f"{names[k]} {tot[k]:.3f} ms" for k in ENG) + f", per-launch best {tot['best']:.3f} ms")
