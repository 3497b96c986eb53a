"""Phase timing of conv3x3_f16x3_row_kernel from the diagnostic (ATMVFI_STAMP) library: per-wave cycle sums."""
import ctypes, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
hip_ops = importlib.import_module("atm-vfi_amd.hip_ops")
hip_ops.LIB_PATH = os.path.join(ROOT, "tools", "lib", "libatmvfi_hip_stamp.so")
hip_ops.load_library.__defaults__ = (hip_ops.LIB_PATH,)
dev = torch.device("cuda:0")
ops = hip_ops.HipOps(dev)
H, W, cin, cout = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (1088, 1920, 101, 101)
r4 = lambda c: (c + 3) // 4 * 4
g = torch.Generator().manual_seed(0)
x = (torch.rand(1, H, W, r4(cin), generator=g) * 2 - 1).to(dev)
w = ((torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) / (3 * cin ** 0.5)).to(dev)
b = torch.zeros(cout, device=dev)
pw = ops.pack_weight(0, w)
y = torch.empty(1, H, W, r4(cout), device=dev)
nblk = ((H + 15) // 16) * ((W + 15) // 16) * 8
buf = torch.zeros(nblk * 8 * 8, dtype=torch.int64, device=dev)
ops.lib.atmvfi_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
ops.lib.atmvfi_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
for _ in range(3):
    ops.conv(x[..., :cin], pw, y[..., :cout], 1, 1, 1, b, b)
torch.cuda.synchronize()
t = buf.reshape(-1, 8).double()
t = t[t.sum(1) > 0]
names = ["prologue", "issue loads (DMA + halo loads)", "LDS reads + MFMA", "vmcnt(0) wait", "stage barrier", "halo convert+write+barrier", "epilogue", "loop top (stage decode)"]
tot = t.sum(1).mean().item()
print(f"{H}x{W} {cin}->{cout}: waves {t.shape[0]}, mean cycles per wave {tot:.0f} (s_memtime ticks)")
for k in range(8):
    print(f"  {names[k]:28s} {t[:, k].mean().item():10.0f}  {100 * t[:, k].mean().item() / tot:5.1f} %")
