"""Where does the round-2 reproducer (attention.hip of commit 40b1371 built WITH the SLP vectorizer, tools/slp_check.py) go wrong?
Runs the failing case of tests/test_gpu_ops.py::test_window_attention (ws 12, hd 84, 12x20 map, shift 6, cross attention) on that
library and on the product library and prints which (window, query, head, component) entries of the motion output differ."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
H = importlib.import_module("atm-vfi_amd.hip_ops")
windows = importlib.import_module("atm-vfi_amd.windows")
from cpu_ops import CpuOps

dev = torch.device("cuda:0")
ws, hd, frames, h, w, shift = 12, 84, 2, 12, 20, 6
heads, C = 8, 8 * 84
g = torch.Generator().manual_seed(ws * 100 + hd + shift)
geo = windows.build_window_geometry(frames, h, w, ws, shift)
bw, n = frames * geo.n_windows, ws * ws
qkv = (torch.rand(bw * n, 3 * C, generator=g) * 2 - 1) * 1.5
oc, mc = torch.empty(bw * n, C), torch.empty(bw * n, heads, 2)
CpuOps().window_attention(qkv, oc, mc, geo.labels, bw, geo.n_windows, ws, heads, hd, bw // 2)
for name in ("libatmvfi_hip.so", "libatmvfi_hip_oldattn_noslp.so", "libatmvfi_hip_oldattn_slp.so"):
    lib = os.path.join(ROOT, "atm-vfi_amd" if name == "libatmvfi_hip.so" else os.path.join("tools", "lib"), name)
    H.load_library.__defaults__ = (lib,)
    ops = H.HipOps(dev)
    og = torch.full((bw * n, C), 9.0, device=dev)
    mg = torch.full((bw * n, heads, 2), 9.0, device=dev)
    ops.window_attention(qkv.to(dev), og, mg, geo.labels.to(dev), bw, geo.n_windows, ws, heads, hd, bw // 2)
    torch.cuda.synchronize()
    d = (mg.cpu() - mc).abs().reshape(bw, n, heads, 2)
    bad = (d > 1e-4).nonzero()
    print(f"== {name}: out max|d| {(og.cpu() - oc).abs().max():.2e}  motion max|d| {d.max():.3e}  wrong entries {len(bad)} of {d.numel()}")
    if len(bad):
        print("   windows", sorted(set(bad[:, 0].tolist())), " heads", sorted(set(bad[:, 2].tolist())), " components (0 = x, 1 = y)", sorted(set(bad[:, 3].tolist())))
        qs = sorted(set(bad[:, 1].tolist()))
        print("   queries", qs[:8], "...", qs[-8:], f"({len(qs)} of {n}; waves {sorted(set(q // 16 for q in qs))})")
        for b_, q_, h_, c_ in bad[:6].tolist():
            print(f"   window {b_} q {q_} head {h_} comp {c_}: got {mg.cpu().reshape(bw, n, heads, 2)[b_, q_, h_, c_]:.5f} want {mc.reshape(bw, n, heads, 2)[b_, q_, h_, c_]:.5f}")
        # is the wrong value the right value of ANOTHER query?  (an indexing / cross-lane fault rather than arithmetic)
        got = mg.cpu().reshape(bw, n, heads, 2)
        want = mc.reshape(bw, n, heads, 2)
        b_, q_, h_, c_ = bad[0].tolist()
        near = ((want[b_, :, h_, c_] - got[b_, q_, h_, c_]).abs() < 1e-4).nonzero().flatten().tolist()
        print(f"   the wrong value of (window {b_}, q {q_}, head {h_}, comp {c_}) equals the reference value of queries {near}")
        diff = got[b_, q_, h_, c_] - want[b_, q_, h_, c_]
        print(f"   difference {diff:.5f}")
