"""Same-process A/B of the LDS-DMA GEMM (gemm_split.hip) between two builds of the library, interleaved per shape:
    python tools/bench_split_ab.py [tools/lib/libatmvfi_hip_old.so]        (B = the product build: gemm_pp.hip; A = another
    build, or without an argument the reference schedule gemm_split.hip of the same build, atmvfi_gemm_params.tile_wn = -1)
Shapes: the network's linears (with / without bias + residual + row map), deconvs (plane sink) and strided convs at 1080p."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
hip_ops = importlib.import_module("atm-vfi_amd.hip_ops")
dev = torch.device("cuda:0")
opsB = hip_ops.HipOps(dev)
opsA = hip_ops.HipOps(dev)
if len(sys.argv) > 1 and sys.argv[1] == "duo":
    opsA.gemm_tile_wn = -3          # A = gemm_pp.hip, B = gemm_duo.hip (128 x 128 tiles, two workgroups per CU) of the same library
    opsB.gemm_tile_wn = -2
elif len(sys.argv) > 1:
    opsA.lib = hip_ops.load_library(os.path.join(ROOT, sys.argv[1]))
else:
    opsA = hip_ops.HipOps(dev, lib_path=os.path.join(ROOT, "tools", "lib", "libatmvfi_hip_ref.so"))
    opsA.gemm_tile_wn = -1          # A = the reference schedule (gemm_split.hip; in the diagnostic library since round 6)
g = torch.Generator().manual_seed(0)


def timeit(fn, reps=5, inner=3):
    best = 1e9
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(inner):
            fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / inner)
    return best


def report(tag, fl, fa, fb, ya, yb):
    fa(); fb(); torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(ya, yb))
    ta, tb = [], []
    for _ in range(3):
        ta.append(timeit(fa)); tb.append(timeit(fb))
    ta, tb = min(ta), min(tb)
    print(f"{tag:44s} A {ta:.3f} ms {fl / ta / 1e9:6.1f} TF/s | B {tb:.3f} ms {fl / tb / 1e9:6.1f} TF/s | B/A {tb / ta:.3f} identical={same}", flush=True)


# ---- linears
for m, n, k, res in [(65280, 1152, 384, False), (65280, 1536, 384, False), (65280, 384, 1536, True), (65280, 384, 384, True),
                     (16320, 2688, 672, False), (16320, 672, 2688, True), (17280, 2016, 672, False)]:
    x = (torch.rand(m, k, generator=g) * 2 - 1).to(dev)
    w = ((torch.rand(n, k, generator=g) * 2 - 1) / k ** 0.5).to(dev)
    b = (torch.rand(n, generator=g) - 0.5).to(dev)
    r = (torch.rand(m, n, generator=g) - 0.5).to(dev) if res else None
    pw = opsB.pack_weight(1, w)
    pl = hip_ops.Planes.alloc(m, k, dev)
    opsB.split_planes(x, pl)
    ya, yb = torch.empty(m, n, device=dev), torch.empty(m, n, device=dev)
    report(f"linear M{m} N{n} K{k} res={int(res)}", 2.0 * m * n * k, lambda: opsA.linear(pl, pw, ya, b, r), lambda: opsB.linear(pl, pw, yb, b, r), [ya], [yb])

# ---- deconvs into a plane sink (decoder stages; network_base.py:27-32)
for (bn, h, w_, cin, cout) in [(1, 136, 240, 773, 389), (1, 272, 480, 389, 197), (1, 544, 960, 197, 101), (1, 272, 480, 256, 128), (1, 544, 960, 128, 64)]:
    x = (torch.rand(bn * h * w_, cin, generator=g) * 2 - 1).to(dev)
    wt = ((torch.rand(cin, cout, 2, 2, generator=g) * 2 - 1) / cin ** 0.5).to(dev)
    b = (torch.rand(cout, generator=g) - 0.5).to(dev)
    pr = torch.rand(cout, generator=g).to(dev)
    pw = opsB.pack_weight(2, wt)
    pl = hip_ops.Planes.alloc(bn * h * w_, cin, dev)
    opsB.split_planes(x, pl)
    sa = hip_ops.Planes.alloc(bn * 4 * h * w_, cout, dev)
    sb = hip_ops.Planes.alloc(bn * 4 * h * w_, cout, dev)
    fl = 2.0 * bn * h * w_ * cin * 4 * cout
    report(f"deconv {h}x{w_} {cin}->{cout} (plane sink)", fl,
           lambda: opsA.deconv(None, pw, None, bias=b, prelu=pr, planes=pl, sink=sa, in_shape=(bn, h, w_, cin)),
           lambda: opsB.deconv(None, pw, None, bias=b, prelu=pr, planes=pl, sink=sb, in_shape=(bn, h, w_, cin)), [sa.t], [sb.t])

# ---- strided 3x3 convs from planes into a plane sink (encoder / fusion; network_base.py:20-25, 73-85)
for (bn, h, w_, cin, cout, stride) in [(2, 544, 960, 48, 96, 2), (2, 272, 480, 96, 192, 2), (2, 136, 240, 192, 384, 2), (2, 272, 480, 96, 96, 4),
                                       (1, 1088, 1920, 64, 64, 2), (1, 544, 960, 256, 128, 2), (1, 272, 480, 512, 256, 2)]:
    x = (torch.rand(bn * h * w_, cin, generator=g) * 2 - 1).to(dev)
    wt = ((torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) / (9 * cin) ** 0.5).to(dev)
    b = (torch.rand(cout, generator=g) - 0.5).to(dev)
    pw = opsB.pack_weight(0, wt)
    pl = hip_ops.Planes.alloc(bn * h * w_, cin, dev)
    opsB.split_planes(x, pl)
    ho, wo = (h + 2 - 3) // stride + 1, (w_ + 2 - 3) // stride + 1
    sa = hip_ops.Planes.alloc(bn * ho * wo, cout, dev)
    sb = hip_ops.Planes.alloc(bn * ho * wo, cout, dev)
    fl = 2.0 * bn * ho * wo * 9 * cin * cout
    report(f"conv s{stride} {h}x{w_} {cin}->{cout} (plane sink)", fl,
           lambda: opsA.conv_planes(pl, bn, h, w_, pw, out=None, stride=stride, pad=1, dil=1, bias=b, sink=sa),
           lambda: opsB.conv_planes(pl, bn, h, w_, pw, out=None, stride=stride, pad=1, dil=1, bias=b, sink=sb), [sa.t], [sb.t])
