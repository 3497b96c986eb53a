"""Same-process A/B of the two window-attention kernels (exact fp32 MFMA / f16x3 MFMA) at the 1080p shapes of network_base:
local blocks (ws 8, hd 48, 2 x 136 x 240 tokens) and global blocks (ws 12, hd 84, 2 x 68 x 120 padded to 72 x 120).
    python tools/bench_attn_ab.py"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
hip_ops = importlib.import_module("atm-vfi_amd.hip_ops")
windows = importlib.import_module("atm-vfi_amd.windows")
dev = torch.device("cuda:0")
A, B = hip_ops.HipOps(dev), hip_ops.HipOps(dev)
A.attention_f16x3 = False
g = torch.Generator().manual_seed(0)


def timeit(fn, reps=7, inner=5):
    best = 1e9
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(inner):
            fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / inner)
    return best


for ws, hd, frames, h, w, shift, planes in [(8, 48, 2, 136, 240, 0, True), (8, 48, 2, 136, 240, 4, True), (12, 84, 2, 68, 120, 6, True),
                                            (12, 84, 2, 68, 120, 0, False), (8, 48, 2, 68, 120, 4, True), (8, 48, 2, 32, 56, 4, True)]:
    heads = 8
    C = heads * hd
    geo = windows.build_window_geometry(frames, h, w, ws, shift)
    bw, n = frames * geo.n_windows, ws * ws
    qkv = ((torch.rand(bw * n, 3 * C, generator=g) * 2 - 1) * 1.5).to(dev)
    labels = None if geo.labels is None else geo.labels.to(dev)
    outs = []
    for ops in (A, B):
        o = None if planes else torch.empty(bw * n, C, device=dev)
        m = torch.empty(bw * n, heads, 2, device=dev)
        pl = hip_ops.Planes.alloc(bw * n, C, dev) if planes else None
        outs.append((o, m, pl))
    fa = lambda: A.window_attention(qkv, outs[0][0], outs[0][1], labels, bw, geo.n_windows, ws, heads, hd, bw // 2, planes=outs[0][2])
    fb = lambda: B.window_attention(qkv, outs[1][0], outs[1][1], labels, bw, geo.n_windows, ws, heads, hd, bw // 2, planes=outs[1][2])
    fa(); fb(); torch.cuda.synchronize()
    d_o = ((outs[0][2].t.float() - outs[1][2].t.float()).abs().max().item() if planes else (outs[0][0] - outs[1][0]).abs().max().item())
    d_m = (outs[0][1] - outs[1][1]).abs().max().item()
    ta, tb = min(timeit(fa) for _ in range(3)), min(timeit(fb) for _ in range(3))
    gb = 4.0 * bw * n * 4 * C
    print(f"ws{ws} hd{hd} {h}x{w} shift{shift} planes={int(planes)}: fp32 {ta * 1e3:7.1f} us ({gb / ta / 1e9:5.2f} TB/s) | f16x3 {tb * 1e3:7.1f} us "
          f"({gb / tb / 1e9:5.2f} TB/s) | ratio {tb / ta:.3f} | max|dO| (plane halves when planes=1) {d_o:.2e} max|dmotion| {d_m:.2e}", flush=True)
