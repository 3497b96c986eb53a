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
         (1, 280, 480, 40, 21, 2), (1, 1088, 1920, 101, 101, 0), (1, 1088, 1920, 64, 64, 0)]
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
        y = torch.full((N, H, W, r4(cout)), 7.0, device=dev)
        yc = torch.full((N, H, W, 8), 7.0, device=dev)
        s1 = hip_ops.Planes.alloc(N * H * W, 8 + cout, dev); s2 = hip_ops.Planes.alloc(N * H * W, cout, dev); s3 = hip_ops.Planes.alloc(N * H * W, cout, dev)
        ops.conv3x3_planes(xp, N, H, W, pw, out=y[..., :cout], bias=bias, prelu=slope, wn=wn, defer=defer)
        ops.conv3x3_planes(xp, N, H, W, pw, out=yc[..., :cout - cmin], bias=bias, prelu=None, planes=s1, planes_c0=8, planes_prelu=pslope,
                           planes2=s2, out_cmin=cmin, wn=wn, defer=defer)
        ops.conv3x3_planes(xp, N, H, W, pw, out=None, bias=None, prelu=slope, planes=s3, wn=wn, defer=defer)
        torch.cuda.synchronize()
        res[defer] = (y, yc, s1, s2, s3)
    (y0, c0, a0, b0, d0), (y1, c1, a1, b1, d1) = res[False], res[True]
    scale = max(1.0, float(y0[..., :cout].abs().max()))
    e = [float((y0 - y1).abs().max()), float((c0 - c1).abs().max()), float((a0.to_float() - a1.to_float()).abs().max()),
         float((b0.to_float() - b1.to_float()).abs().max()), float((d0.to_float() - d1.to_float()).abs().max())]
    pads = bool((y1[..., cout:] == 7.0).all()) and bool((c1[..., cout - cmin:] == 7.0).all()) and bool((a1.to_rows()[:, :, :8] == 0).all()) \
        and bool((a1.to_rows()[:, :, 8 + cout:] == 0).all()) and bool((b1.to_rows()[:, :, cout:] == 0).all()) and bool((a1.t[:, :, N * H * W:] == 0).all())
    same = torch.equal(y0, y1)
    if os.environ.get("ATMVFI_DE_WHERE"):
        dd = (y0 - y1).abs()[..., :cout]
        bad = (dd > 2e-5).nonzero()
        print("elements off by > 2e-5:", bad.shape[0], "of", dd.numel())
        if bad.shape[0]:
            ys, xs, cs = bad[:, 1], bad[:, 2], bad[:, 3]
            ty, tx = ys // 16, xs // 16
            tiles = torch.unique(ty * 1000 + tx)
            print("tiles touched:", tiles.numel(), tiles[:40].tolist())
            print("rows in tile:", torch.unique(ys % 16).tolist(), "cols in tile:", torch.unique(xs % 16).tolist(), "channels:", torch.unique(cs).tolist()[:40])
            for k in range(min(8, bad.shape[0])):
                n_, y_, x_, c_ = bad[k].tolist()
                print("  ", (y_, x_, c_), float(y0[n_, y_, x_, c_]), float(y1[n_, y_, x_, c_]))
    # timing: sink + prelu launch, 20 reps
    tt = {}
    for defer in (False, True):
        s1 = hip_ops.Planes.alloc(N * H * W, cout, dev)
        for _ in range(3):
            ops.conv3x3_planes(xp, N, H, W, pw, out=None, bias=bias, prelu=slope, planes=s1, planes_prelu=pslope, wn=wn, defer=defer)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            ops.conv3x3_planes(xp, N, H, W, pw, out=None, bias=bias, prelu=slope, planes=s1, planes_prelu=pslope, wn=wn, defer=defer)
        torch.cuda.synchronize(); tt[defer] = (time.perf_counter() - t0) / 20 * 1e3
    print(f"N{N} {H}x{W} {cin}->{cout} wn{wn}: max|diff| rows {e[0]:.2e} compact {e[1]:.2e} sink1 {e[2]:.2e} sink2 {e[3]:.2e} sink-only {e[4]:.2e} (scale {scale:.2f}) "
          f"pads untouched {pads} bit-identical {same} | two-acc {tt[False]:.3f} ms  deferred {tt[True]:.3f} ms", flush=True)
    assert os.environ.get("ATMVFI_DE_NOCHECK") or (max(e) <= 2e-5 * scale and pads)
