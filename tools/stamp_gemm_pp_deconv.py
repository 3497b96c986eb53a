"""Phase cycles of gemm_pp_kernel on the decoder's ConvTranspose2d 2x2 / stride 2 launches (plane in, plane sink) and the strided
convs, from the diagnostic (ATMVFI_STAMP) library: python tools/stamp_gemm_pp_deconv.py"""
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
ops.lib.atmvfi_debug_set_pp_stamp_buffer.argtypes = [ctypes.c_void_p]


def report(tag, nblk, run):
    buf = torch.zeros(nblk * 8 * 8, dtype=torch.int64, device=dev)
    ops.lib.atmvfi_debug_set_pp_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
    for _ in range(100):
        run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); run(); e.record(); torch.cuda.synchronize()
    t = buf.reshape(-1, 8).double()
    tv = t[t[:, 3] > 0]
    nk = tv[0, 3].item()
    rloop = tv[:, 4].median().item()
    clk = tv[:, 1].median().item() / rloop * 100.0 if rloop else float("nan")
    for grp, rows in (("group 0", t.reshape(-1, 8, 8)[:, :4].reshape(-1, 8)), ("group 1", t.reshape(-1, 8, 8)[:, 4:].reshape(-1, 8))):
        rows = rows[rows[:, 3] > 0]
        lp, ep, s0, s1, s2 = (rows[:, i].median().item() for i in (1, 2, 5, 6, 7))
        print(f"{tag} {grp}: {s.elapsed_time(e) * 1e3:.0f} us; k-steps {nk:.0f}; cycles per tile: loop {lp:.0f} ({lp / nk:.0f}/step)  epilogue {ep:.0f}"
              f" = fold + transpose {s0:.0f} + constants / rows {s1:.0f} + stores {s2:.0f}; clock {clk:.0f} MHz", flush=True)


for (bn, h, w_, cin, cout) in [(1, 544, 960, 197, 101), (1, 272, 480, 389, 197), (1, 544, 960, 128, 64)]:
    x = (torch.rand(bn * h * w_, cin, generator=g) * 2 - 1).to(dev)
    wt = ((torch.rand(cin, cout, 2, 2, generator=g) * 2 - 1) / cin ** 0.5).to(dev)
    b = (torch.rand(cout, generator=g) - 0.5).to(dev)
    pr = torch.rand(cout, generator=g).to(dev)
    pw = ops.pack_weight(2, wt)
    pl = hip_ops.Planes.alloc(bn * h * w_, cin, dev)
    ops.split_planes(x, pl)
    sk = hip_ops.Planes.alloc(bn * 4 * h * w_, cout, dev)
    m, n = bn * h * w_, 4 * ((cout + 3) // 4 * 4)
    nblk = ((m + 255) // 256 + 7) // 8 * 8 * ((n + 127) // 128)
    report(f"deconv {h}x{w_} {cin}->{cout}", nblk,
           lambda: ops.deconv(None, pw, None, bias=b, prelu=pr, planes=pl, sink=sk, in_shape=(bn, h, w_, cin)))
