"""A/B: nn.Linear rows through the fp32-input f16x3 GEMM vs the split-plane (LDS-DMA) GEMM, interleaved in one process."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
hip_ops = importlib.import_module("atm-vfi_amd.hip_ops")
dev = torch.device("cuda:0")
ops = hip_ops.HipOps(dev)
shapes = [(65280, 1536, 384), (65280, 384, 1536), (65280, 1152, 384), (16320, 2688, 672), (16320, 672, 2688), (17280, 2016, 672),
          (65280, 384, 384), (1000, 200, 100)]
USE_RES = os.environ.get("RES", "1") == "1"
g = torch.Generator().manual_seed(0)
for m, n, k in shapes:
    x = (torch.rand(m, k, generator=g) * 2 - 1).to(dev)
    w = ((torch.rand(n, k, generator=g) * 2 - 1) / k ** 0.5).to(dev)
    b = (torch.rand(n, generator=g) - 0.5).to(dev)
    res = (torch.rand(m, n, generator=g) - 0.5).to(dev) if USE_RES else None
    pw = ops.pack_weight(1, w)
    y0 = torch.empty(m, n, device=dev)
    y1 = torch.empty(m, n, device=dev)
    pl = hip_ops.Planes.alloc(m, k, dev)
    ops.split_planes(x, pl)
    ops.linear(x, pw, y0, b, res)
    ops.linear(pl, pw, y1, b, res)
    torch.cuda.synchronize()
    ref = (x.double() @ w.double().t() + b.double() + (res.double() if USE_RES else 0))
    e0 = (y0.double() - ref).abs().max().item()
    e1 = (y1.double() - ref).abs().max().item()
    same = torch.equal(y0, y1)
    ts = {"f32in": [], "planes": [], "split": []}
    for rep in range(5):
        for name, fn in (("f32in", lambda: ops.linear(x, pw, y0, b, res)), ("planes", lambda: ops.linear(pl, pw, y1, b, res)),
                         ("split", lambda: ops.split_planes(x, pl))):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(3):
                fn()
            e.record(); torch.cuda.synchronize()
            ts[name].append(s.elapsed_time(e) / 3)
    fl = 2.0 * m * n * k
    t0, t1, t2 = (min(ts[k_]) for k_ in ("f32in", "planes", "split"))
    print(f"M{m} N{n} K{k}: f32in {t0:.3f} ms {fl/t0/1e9:6.1f} TF/s | planes {t1:.3f} ms {fl/t1/1e9:6.1f} TF/s | split pass {t2:.3f} ms"
          f" | err {e0:.2e} {e1:.2e} identical={same}", flush=True)
