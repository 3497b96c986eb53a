"""Phase ticks of gemm_f16x3_kernel on the decoder's deconv and a strided conv (diagnostic ATMVFI_STAMP library)."""
import ctypes, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
hip_ops = importlib.import_module("atm-vfi_amd.hip_ops")
hip_ops.LIB_PATH = os.path.join(ROOT, "tools", "lib", "libatmvfi_hip_stamp.so")
hip_ops.load_library.__defaults__ = (hip_ops.LIB_PATH,)
dev = torch.device("cuda:0")
ops = hip_ops.HipOps(dev)
g = torch.Generator().manual_seed(0)
names = ["first setup+load issue", "chunk-0 convert+ds_write+barrier", "next-chunk load issue", "LDS reads + MFMA", "convert + ds_write",
         "barrier", "out rows + next tile setup/load issue", "epilogue (bias/prelu/stores)"]
ops.lib.atmvfi_debug_set_gemm_stamp_buffer.argtypes = [ctypes.c_void_p]
def report(tag, fn):
    buf = torch.zeros(256 * 8 * 8, dtype=torch.int64, device=dev)
    ops.lib.atmvfi_debug_set_gemm_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    t = buf.reshape(-1, 8).double(); t = t[t.sum(1) > 0]; tot = t.sum(1).mean().item()
    print(f"{tag}: {e0.elapsed_time(e1) / 20:.3f} ms, waves {t.shape[0]}, mean ticks per wave {tot:.0f}")
    for i in range(8):
        print(f"  {names[i]:40s} {t[:, i].mean().item():10.0f}  {100 * t[:, i].mean().item() / tot:5.1f} %", flush=True)
for (h, w, cin, cout) in [(544, 960, 197, 101), (272, 480, 389, 197), (544, 960, 128, 64)]:
    x = (torch.rand(1, h, w, (cin + 3) // 4 * 4, generator=g) * 2 - 1).to(dev)[..., :cin]
    wt = ((torch.rand(cin, cout, 2, 2, generator=g) * 2 - 1) / cin ** 0.5).to(dev)
    b = (torch.rand(cout, generator=g) - 0.5).to(dev); pr = torch.full((cout,), 0.25, device=dev)
    ipr = torch.full(((cin + 31) // 32 * 32,), 0.25, device=dev)
    pw = ops.pack_weight(hip_ops.GEMM_DECONV, wt)
    y = torch.empty(1, 2 * h, 2 * w, (cout + 3) // 4 * 4, device=dev)[..., :cout]
    report(f"deconv {h}x{w} {cin}->{cout} in_prelu", lambda: ops.deconv(x, pw, y, bias=b, prelu=pr, in_prelu=ipr))
    report(f"deconv {h}x{w} {cin}->{cout}", lambda: ops.deconv(x, pw, y, bias=b, prelu=pr))
for (h, w, cin, cout, k, s) in [(1088, 1920, 64, 64, 3, 2), (544, 960, 64, 128, 3, 2), (1088, 1920, 24, 48, 3, 2)]:
    x = (torch.rand(1, h, w, cin, generator=g) * 2 - 1).to(dev)
    wt = ((torch.rand(cout, cin, k, k, generator=g) * 2 - 1) / (cin * k * k) ** 0.5).to(dev)
    b = (torch.rand(cout, generator=g) - 0.5).to(dev); pr = torch.full((cout,), 0.25, device=dev)
    pw = ops.pack_weight(hip_ops.GEMM_CONV, wt)
    y = torch.empty(1, h // s, w // s, cout, device=dev)
    report(f"conv {h}x{w} {cin}->{cout} k{k} s{s}", lambda: ops.conv(x, pw, y, stride=s, pad=1, bias=b, prelu=pr))
