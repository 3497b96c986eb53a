"""Deferred-epilogue 3x3 plane kernel against the two-accumulator one (wn | 16): every output kind, ragged sizes, timing."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
hip_ops = importlib.import_module("atm-vfi_amd.hip_ops")
if os.environ.get("ATMVFI_LIB"):          # an experiment build (wrong results allowed: ATMVFI_DE_NOCHECK=1)
    hip_ops.LIB_PATH = os.path.join(ROOT, os.environ["ATMVFI_LIB"])
    hip_ops.load_library.__defaults__ = (hip_ops.LIB_PATH,)
dev = torch.device("cuda:0")
ops = hip_ops.HipOps(dev)
g = torch.Generator().manual_seed(3)
r4 = lambda c: (c + 3) // 4 * 4
cases = [(1, 272, 480, 101, 101, 0), (1, 300, 500, 64, 64, 0), (2, 150, 250, 37, 101, 7), (1, 290, 490, 104, 48, 0), (1, 272, 480, 197, 197, 0),
         (1, 280, 480, 40, 21, 2), (1, 136, 240, 1352, 768, 0), (1, 272, 480, 389, 389, 0), (1, 544, 960, 197, 197, 0), (1, 1088, 1920, 101, 101, 0),
         (1, 1088, 1920, 64, 64, 0), (1, 1088, 1920, 116, 64, 0), (2, 544, 960, 48, 48, 0)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in sys.argv[1:7])]
for (N, H, W, cin, cout, wn) in cases:
    x = ((torch.rand(N, H, W, r4(cin), generator=g) * 2 - 1) * 1.5).to(dev)
    wt = ((torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) / (3 * cin ** 0.5)).to(dev)
    bias = ((torch.rand(cout, generator=g) * 2 - 1) * 0.2).to(dev)
    slope = (torch.rand(cout, generator=g) * 0.4).to(dev)
    pslope = torch.zeros((cout + 31) // 32 * 32, device=dev); pslope[:cout] = torch.rand(cout, generator=g).to(dev)
    pw = ops.pack_weight(0, wt)
    xp = hip_ops.Planes.alloc(N * H * W, cin, dev)
    ops.split_planes(x[..., :cin].flatten(0, 2), xp)
    cmin = (cout - 5) // 4 * 4 if cout > 8 else 0
    res = {}
    for defer in (False, True):
        s1 = hip_ops.Planes.alloc(N * H * W, 8 + cout, dev); s3 = hip_ops.Planes.alloc(N * H * W, cout, dev); s4 = hip_ops.Planes.alloc(N * H * W, cout, dev)
        ops.conv3x3_planes(xp, N, H, W, pw, out=None, bias=bias, prelu=slope, planes=s1, planes_c0=8, wn=wn, defer=defer)     # at a channel offset
        ops.conv3x3_planes(xp, N, H, W, pw, out=None, bias=None, prelu=slope, planes=s3, wn=wn, defer=defer)                   # no bias
        ops.conv3x3_planes(xp, N, H, W, pw, out=None, bias=bias, prelu=None, planes=s4, wn=wn, defer=defer)                    # no activation
        torch.cuda.synchronize()
        res[defer] = (s1, s3, s4)
    (a0, d0, e0), (a1, d1, e1) = res[False], res[True]
    scale = max(1.0, float(a0.to_float().abs().max()))
    e = [float((a0.to_float() - a1.to_float()).abs().max()), float((d0.to_float() - d1.to_float()).abs().max()), float((e0.to_float() - e1.to_float()).abs().max())]
    pads = bool((a1.to_rows()[:, :, :8] == 0).all()) and bool((a1.to_rows()[:, :, 8 + cout:] == 0).all()) and bool((a1.t[:, :, N * H * W:] == 0).all()) \
        and bool((d1.to_rows()[:, :, cout:] == 0).all()) and bool((e1.t[:, :, N * H * W:] == 0).all())
    same = torch.equal(d0.t, d1.t)
    tt = {}
    s1 = hip_ops.Planes.alloc(N * H * W, cout, dev)
    for rep in range(2):
        for defer in (False, True):
            for _ in range(3):
                ops.conv3x3_planes(xp, N, H, W, pw, out=None, bias=bias, prelu=slope, planes=s1, wn=wn, defer=defer)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20):
                ops.conv3x3_planes(xp, N, H, W, pw, out=None, bias=bias, prelu=slope, planes=s1, wn=wn, defer=defer)
            torch.cuda.synchronize(); tt[defer] = min(tt.get(defer, 1e9), (time.perf_counter() - t0) / 20 * 1e3)
    print(f"N{N} {H}x{W} {cin}->{cout} wn{wn}: max|diff| offset sink {e[0]:.2e} no bias {e[1]:.2e} no activation {e[2]:.2e} (scale {scale:.2f}) "
          f"pads untouched {pads} bit-identical {same} | ms two-acc / deferred: {tt[False]:.3f} / {tt[True]:.3f}", flush=True)
    assert os.environ.get("ATMVFI_DE_NOCHECK") or (max(e) <= 2e-5 * scale and pads)
