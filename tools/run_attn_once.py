"""A few launches of the f16x3 window attention at the 1080p local / global shapes (the workload of tools/pmc_attn.sh)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
hip_ops = importlib.import_module("atm-vfi_amd.hip_ops")
windows = importlib.import_module("atm-vfi_amd.windows")
dev = torch.device("cuda:0")
ops = hip_ops.HipOps(dev)
ops.attention_f16x3 = os.environ.get("ATTN", "f16x3") == "f16x3"
g = torch.Generator().manual_seed(0)
which = sys.argv[1] if len(sys.argv) > 1 else "local"
ws, hd, frames, h, w, shift = (8, 48, 2, 136, 240, 4) if which == "local" else (12, 84, 2, 68, 120, 6)
heads = 8
C = heads * hd
geo = windows.build_window_geometry(frames, h, w, ws, shift)
bw, n = frames * geo.n_windows, ws * ws
qkv = ((torch.rand(bw * n, 3 * C, generator=g) * 2 - 1) * 1.5).to(dev)
labels = None if geo.labels is None else geo.labels.to(dev)
m = torch.empty(bw * n, heads, 2, device=dev)
pl = hip_ops.Planes.alloc(bw * n, C, dev)
for _ in range(5):
    ops.window_attention(qkv, None, m, labels, bw, geo.n_windows, ws, heads, hd, bw // 2, planes=pl)
torch.cuda.synchronize()
