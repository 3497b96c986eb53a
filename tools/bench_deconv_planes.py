"""Deconvs of the 1080p network on the fp32-input engine vs on the LDS-DMA GEMM from pre-split planes (split pass timed apart)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
hip_ops = importlib.import_module("atm-vfi_amd.hip_ops")
dev = torch.device("cuda:0")
ops = hip_ops.HipOps(dev)
g = torch.Generator().manual_seed(0)
r4 = lambda c: (c + 3) // 4 * 4
def timed(fn, n=10):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
tot = [0.0, 0.0, 0.0]
for (H, W, cin, cout, inpr) in [(136, 240, 773, 389, False), (272, 480, 389, 197, True), (544, 960, 197, 101, True),
                                (136, 240, 256, 128, False), (272, 480, 256, 128, False), (544, 960, 128, 64, False)]:
    x = (torch.rand(1, H, W, r4(cin), generator=g) * 2 - 1).to(dev)[..., :cin]
    wt = ((torch.rand(cin, cout, 2, 2, generator=g) * 2 - 1) / cin ** 0.5).to(dev)
    b = torch.zeros(cout, device=dev); ipr = torch.full(((cin + 31) // 32 * 32,), 0.25, device=dev)
    pw = ops.pack_weight(hip_ops.GEMM_DECONV, wt)
    y = torch.empty(1, 2 * H, 2 * W, r4(cout), device=dev)[..., :cout]
    y2 = torch.empty(1, 2 * H, 2 * W, r4(cout), device=dev)[..., :cout]
    xp = hip_ops.Planes.alloc(H * W, cin, dev)
    rows = x.reshape(H * W, -1)[:, :cin] if x.is_contiguous() else x.reshape(H * W, cin)
    xr = x.as_strided((H * W, cin), (r4(cin), 1))
    t_f = timed(lambda: ops.deconv(x, pw, y, bias=b, prelu=b, in_prelu=ipr if inpr else None))
    t_s = timed(lambda: ops.split_planes(xr, xp, ipr if inpr else None))
    t_p = timed(lambda: ops.deconv(x, pw, y2, bias=b, prelu=b, planes=xp))
    err = (y - y2).abs().max().item()
    tot[0] += t_f; tot[1] += t_s; tot[2] += t_p
    print(f"deconv {H}x{W} {cin}->{cout}: fp32-input engine {t_f:.3f} ms | split pass {t_s:.3f} + LDS-DMA GEMM {t_p:.3f} ms | max diff {err:.1e}", flush=True)
print(f"total: {tot[0]:.3f} vs split {tot[1]:.3f} + gemm {tot[2]:.3f}")
