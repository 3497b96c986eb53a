"""Phase ticks of the persistent gemm_f16x3_kernel from the diagnostic (ATMVFI_STAMP) library (per-wave sums over all tiles)."""
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
for m, n, k in [(65280, 1536, 384), (65280, 384, 1536), (16320, 2688, 672)]:
    x = (torch.rand(m, k, generator=g) * 2 - 1).to(dev)
    w = ((torch.rand(n, k, generator=g) * 2 - 1) / k ** 0.5).to(dev)
    b = (torch.rand(n, generator=g) - 0.5).to(dev)
    pw = ops.pack_weight(1, w)
    y = torch.empty(m, n, device=dev)
    buf = torch.zeros(256 * 8 * 8, dtype=torch.int64, device=dev)
    ops.lib.atmvfi_debug_set_gemm_stamp_buffer.argtypes = [ctypes.c_void_p]
    ops.lib.atmvfi_debug_set_gemm_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
    for _ in range(50):
        ops.linear(x, pw, y, b)
    torch.cuda.synchronize()
    t = buf.reshape(-1, 8).double()
    t = t[t.sum(1) > 0]
    tot = t.sum(1).mean().item()
    print(f"M{m} N{n} K{k}: waves {t.shape[0]}, mean ticks per wave {tot:.0f}")
    for i in range(8):
        print(f"  {names[i]:40s} {t[:, i].mean().item():10.0f}  {100 * t[:, i].mean().item() / tot:5.1f} %", flush=True)
