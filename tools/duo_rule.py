"""Per-launch A/B of gemm_pp.hip (tile_wn -3) and gemm_duo.hip (-2) inside a forward: which launches does each win?  (the data behind
the selection rule in gemm_split.hip::launch_gemm_split)   python tools/duo_rule.py [H W [variant]]"""
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
a, b = [t.to(dev) for t in pairs.random_pair(1, H, W, seed=3)]
net(a, b)
ops = net._ops_obj
res = {}
for wn in (-3, -2):
    ops.gemm_tile_wn = wn
    best = None
    for rep in range(4):
        ops.profile = []
        net(a, b)
        torch.cuda.synchronize()
        t = [(n, m.get("shape", ""), s.elapsed_time(e)) for n, m, s, e in ops.profile if n.endswith("_split")]
        ops.profile = None
        best = t if best is None else [(n, sh, min(x, y[2])) for (n, sh, x), y in zip(t, best)]
    res[wn] = best
tot = {-3: 0.0, -2: 0.0, "best": 0.0, "rule": 0.0}
print(f"network_{variant} {H}x{W}: {len(res[-3])} plane-input GEMM launches")
for (n, sh, tp), (_, _, td) in zip(res[-3], res[-2]):
    m_, n_, k_ = (int(x) for x in re.match(r"M(\d+) N(\d+) K(\d+)", sh).groups())
    tiles = -(-m_ // 256) * -(-n_ // 128)
    nk = -(-k_ // 32)
    rule_duo = tiles <= 128 or (n_ <= 64 and nk <= 36)
    tot[-3] += tp; tot[-2] += td; tot["best"] += min(tp, td); tot["rule"] += td if rule_duo else tp
    print(f"  {n:16s} {sh:28s} tiles {tiles:5d} k-steps {nk:4d}  pp {tp * 1e3:7.1f} us  duo {td * 1e3:7.1f} us  duo/pp {td / tp:5.2f}  rule: {'duo' if rule_duo else 'pp'}"
          + ("   <-- rule loses" if (td < tp * 0.97) != rule_duo and abs(td - tp) > 0.002 else ""))
print(f"sum: pp {tot[-3]:.3f} ms, duo {tot[-2]:.3f} ms, per-launch best {tot['best']:.3f} ms, rule {tot['rule']:.3f} ms")
