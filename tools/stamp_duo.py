"""Phase cycles of gemm_duo_kernel (128 x 128 tiles, two workgroups per CU) from the diagnostic (ATMVFI_STAMP) library: per wave and
tile, sums over the k-steps of: DMA wait + barrier, fragment reads, second barrier, DMA issue, MFMA issue; then the epilogue."""
import ctypes, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
hip_ops = importlib.import_module("atm-vfi_amd.hip_ops")
hip_ops.LIB_PATH = os.path.join(ROOT, "tools", "lib", "libatmvfi_hip_stamp.so")
hip_ops.load_library.__defaults__ = (hip_ops.LIB_PATH,)
dev = torch.device("cuda:0")
ops = hip_ops.HipOps(dev)
ops.gemm_tile_wn = -2
NW = 4
g = torch.Generator().manual_seed(0)
for m, n, k in [(65280, 1536, 384), (65280, 384, 1536), (16320, 2688, 672)]:
    x = (torch.rand(m, k, generator=g) * 2 - 1).to(dev)
    w = ((torch.rand(n, k, generator=g) * 2 - 1) / k ** 0.5).to(dev)
    pw = ops.pack_weight(1, w)
    y = torch.empty(m, n, device=dev)
    pl = hip_ops.Planes.alloc(m, k, dev)
    ops.split_planes(x, pl)
    nblk = ((m + 127) // 128 + 7) // 8 * 8 * ((n + 127) // 128)
    buf = torch.zeros(nblk * NW * 8, dtype=torch.int64, device=dev)
    ops.lib.atmvfi_debug_set_duo_stamp_buffer.argtypes = [ctypes.c_void_p]
    ops.lib.atmvfi_debug_set_duo_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
    for _ in range(100):
        ops.linear(pl, pw, y)
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        ops.linear(pl, pw, y)
    e.record(); torch.cuda.synchronize()
    t = buf.reshape(-1, 8).double()
    tv = t[t[:, 7] > 0]
    nk = (k + 31) // 32
    med = [tv[:, i].median().item() for i in range(8)]
    print(f"M{m} N{n} K{k}: {s.elapsed_time(e) / 20 * 1e3:.1f} us per launch (stamped build), {nk} k-steps; median cycles per tile and wave: prologue issue {med[0]:.0f}; "
          f"per k-step: DMA wait + barrier {med[1] / nk:.0f}, fragment reads {med[2] / nk:.0f}, barrier {med[3] / nk:.0f}, DMA issue {med[4] / nk:.0f}, "
          f"MFMA phase {med[5] / nk:.0f} (sum {sum(med[1:6]) / nk:.0f}); epilogue {med[6]:.0f}; tile {med[7]:.0f}", flush=True)
