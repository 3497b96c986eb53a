"""In-kernel clock and phase cycles of gemm_split_kernel from the diagnostic (ATMVFI_STAMP) library."""
import ctypes, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
hip_ops = importlib.import_module("atm-vfi_amd.hip_ops")
hip_ops.LIB_PATH = os.path.join(ROOT, "tools", "lib", "libatmvfi_hip_stamp.so")
hip_ops.load_library.__defaults__ = (hip_ops.LIB_PATH,)
dev = torch.device("cuda:0")
ops = hip_ops.HipOps(dev)
ops.gemm_tile_wn = -3          # the ping-pong kernel, whatever the launcher would choose
g = torch.Generator().manual_seed(0)
for m, n, k in [(65280, 1536, 384), (2048, 1536, 384), (512, 1536, 384), (65280, 384, 1536)]:
    x = (torch.rand(m, k, generator=g) * 2 - 1).to(dev)
    w = ((torch.rand(n, k, generator=g) * 2 - 1) / k ** 0.5).to(dev)
    pw = ops.pack_weight(1, w)
    y = torch.empty(m, n, device=dev)
    pl = hip_ops.Planes.alloc(m, k, dev)
    ops.split_planes(x, pl)
    nblk = ((m + 255) // 256 + 7) // 8 * 8 * ((n + 127) // 128)
    buf = torch.zeros(max(nblk * 8 * 8, 256 * 64 + 256 * 128 + 4096), dtype=torch.int64, device=dev)
    ops.lib.atmvfi_debug_set_pp_stamp_buffer.argtypes = [ctypes.c_void_p]
    ops.lib.atmvfi_debug_set_pp_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
    for _ in range(200):           # keep the chip loaded so the clock settles
        ops.linear(pl, pw, y)
    torch.cuda.synchronize()
    t = buf[:256 * 64].reshape(-1, 8).double()          # (behind them: the per-k-step sums of tools/stamp_pp_ksteps.py)
    tv = t[t[:, 3] > 0]
    nk = tv[0, 3].item()
    pro, loop, epi = (tv[:, i].median().item() for i in range(3))
    rloop = tv[:, 4].median().item()
    clk = loop / rloop * 100.0 if rloop else float("nan")     # s_memrealtime ticks at 100 MHz
    for grp, rows in (("group 0", t.reshape(-1, 8, 8)[:, :4].reshape(-1, 8)), ("group 1", t.reshape(-1, 8, 8)[:, 4:].reshape(-1, 8))):
        rows = rows[rows[:, 3] > 0]
        lp, ep, s0, s1, s2 = (rows[:, i].median().item() for i in (1, 2, 5, 6, 7))
        print(f"M{m} N{n} K{k} {grp}: k-steps {nk:.0f}; cycles per tile: prologue (first tile) {pro:.0f}  loop {lp:.0f} ({lp / nk:.0f}/step)  epilogue {ep:.0f}"
              f" = fold + transpose {s0:.0f} + constants / rows / residual {s1:.0f} + stores {s2:.0f}; in-kernel clock {clk:.0f} MHz", flush=True)
