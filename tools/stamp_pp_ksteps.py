"""Cycles of every k-step of a gemm_pp tile by its index inside the tile (diagnostic ATMVFI_STAMP library): where do the tile-boundary
cycles of a short-K launch go?   make -C atm-vfi_amd/csrc stamp && python tools/stamp_pp_ksteps.py"""
import ctypes, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
hip_ops = importlib.import_module("atm-vfi_amd.hip_ops")
hip_ops.LIB_PATH = os.path.join(ROOT, "tools", "lib", "libatmvfi_hip_stamp.so")
hip_ops.load_library.__defaults__ = (hip_ops.LIB_PATH,)
dev = torch.device("cuda:0")
ops = hip_ops.HipOps(dev)
ops.gemm_tile_wn = -3
g = torch.Generator().manual_seed(0)
for m, n, k in [(65280, 1536, 384), (65280, 1152, 384), (16320, 2688, 672)]:
    x = (torch.rand(m, k, generator=g) * 2 - 1).to(dev)
    w = ((torch.rand(n, k, generator=g) * 2 - 1) / k ** 0.5).to(dev)
    pw = ops.pack_weight(1, w)
    y = torch.empty(m, n, device=dev)
    pl = hip_ops.Planes.alloc(m, k, dev)
    ops.split_planes(x, pl)
    buf = torch.zeros(256 * 64 + 256 * 128 + 4096, dtype=torch.int64, device=dev)
    ops.lib.atmvfi_debug_set_pp_stamp_buffer.argtypes = [ctypes.c_void_p]
    ops.lib.atmvfi_debug_set_pp_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
    for _ in range(100):
        ops.linear(pl, pw, y)
    torch.cuda.synchronize()
    ks = buf[256 * 64:256 * 64 + 256 * 128].reshape(256, 8, 16).double()
    ph = buf[:256 * 64].reshape(256, 8, 8).double()
    nk = int(ph[0, 0, 3].item())
    for grp in (0, 1):
        v = ks[:, 4 * grp:4 * grp + 4].reshape(-1, 16).median(0).values
        lp = ph[:, 4 * grp:4 * grp + 4, 1].median().item()
        ep = ph[:, 4 * grp:4 * grp + 4, 2].median().item()
        print(f"M{m} N{n} K{k} group {grp}: loop {lp:.0f} epilogue {ep:.0f} cycles per tile; per k-step index: " + " ".join(f"{int(c)}" for c in v[:nk].tolist()), flush=True)
