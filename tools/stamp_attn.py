"""Phase cycles of the f16x3 window-attention kernel (diagnostic build `make -C atm-vfi_amd/csrc stamp`): workgroup 0, wave 0,
averaged over the items it walks.   python tools/stamp_attn.py"""
import ctypes, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
hip_ops = importlib.import_module("atm-vfi_amd.hip_ops")
hip_ops.LIB_PATH = os.path.join(ROOT, "tools", "lib", "libatmvfi_hip_stamp.so")
hip_ops.load_library.__defaults__ = (hip_ops.LIB_PATH,)
windows = importlib.import_module("atm-vfi_amd.windows")
dev = torch.device("cuda:0")
ops = hip_ops.HipOps(dev)
buf = torch.zeros(16, dtype=torch.int64, device=dev)
ops.lib.atmvfi_debug_set_attn_stamp_buffer.argtypes = [ctypes.c_void_p]
ops.lib.atmvfi_debug_set_attn_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
g = torch.Generator().manual_seed(0)
NAMES = ["stage K/V (wait for rows, convert, LDS)", "labels + split Q", "barrier", "request next rows", "S = K Q^T", "softmax + split P",
         "O = P V + stores", "end barrier"]
for ws, hd, frames, h, w, shift in [(8, 48, 2, 136, 240, 4), (12, 84, 2, 68, 120, 6)]:
    heads = 8
    C = heads * hd
    geo = windows.build_window_geometry(frames, h, w, ws, shift)
    bw, n = frames * geo.n_windows, ws * ws
    qkv = ((torch.rand(bw * n, 3 * C, generator=g) * 2 - 1) * 1.5).to(dev)
    labels = None if geo.labels is None else geo.labels.to(dev)
    m = torch.empty(bw * n, heads, 2, device=dev)
    pl = hip_ops.Planes.alloc(bw * n, C, dev)
    for _ in range(3):
        buf.zero_()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        ops.window_attention(qkv, None, m, labels, bw, geo.n_windows, ws, heads, hd, bw // 2, planes=pl)
        e.record(); torch.cuda.synchronize()
    t = buf.cpu().tolist()
    items = max(t[8], 1)
    print(f"ws{ws} hd{hd} {h}x{w}: {s.elapsed_time(e) * 1e3:.1f} us, {items} items per workgroup, {sum(t[:8]) / items:.0f} shader-clock cycles (s_memtime) per item")
    for nme, c in zip(NAMES, t[:8]):
        print(f"    {nme:42s} {c / items:9.0f}")
