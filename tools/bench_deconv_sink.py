"""Deconvs of the 1080p network on the LDS-DMA GEMM: fp32 NHWC output vs plane-sink output (same-process A/B)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
hip_ops = importlib.import_module("atm-vfi_amd.hip_ops")
if os.environ.get("ATMVFI_LIB"):          # A/B of two builds on one box: ATMVFI_LIB=tools/lib/libatmvfi_hip_base.so
    hip_ops.LIB_PATH = os.path.join(ROOT, os.environ["ATMVFI_LIB"])
    hip_ops.load_library.__defaults__ = (hip_ops.LIB_PATH,)
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
for (H, W, cin, cout) in [(136, 240, 773, 389), (272, 480, 389, 197), (544, 960, 197, 101), (136, 240, 256, 128), (272, 480, 256, 128), (544, 960, 128, 64)]:
    x = (torch.rand(H * W, r4(cin), generator=g) * 2 - 1).to(dev)[:, :cin]
    wt = ((torch.rand(cin, cout, 2, 2, generator=g) * 2 - 1) / cin ** 0.5).to(dev)
    b = torch.zeros(cout, device=dev)
    pw = ops.pack_weight(hip_ops.GEMM_DECONV, wt)
    y = torch.empty(1, 2 * H, 2 * W, r4(cout), device=dev)[..., :cout]
    xp = hip_ops.Planes.alloc(H * W, cin, dev)
    ops.split_planes(x, xp)
    sink = hip_ops.Planes.alloc(4 * H * W, cout, dev)
    t_f = timed(lambda: ops.deconv(None, pw, y, bias=b, prelu=b, planes=xp, in_shape=(1, H, W, cin)))
    t_s = timed(lambda: ops.deconv(None, pw, None, bias=b, prelu=b, planes=xp, sink=sink, in_shape=(1, H, W, cin)))
    fl = 2.0 * H * W * 4 * cout * cin
    print(f"deconv {H}x{W} {cin}->{cout}: fp32 out {t_f:.3f} ms ({fl / t_f / 1e9:.0f} TF/s) | plane sink {t_s:.3f} ms ({fl / t_s / 1e9:.0f} TF/s)", flush=True)
