"""Phase cycles of stem_kernel (atmvfi_stem_fused) from the diagnostic (ATMVFI_STAMP) library: per wave, summed over the tiles of a
workgroup -- wait at the tile's first barrier | layer 1 (VALU) | barrier | layer 2 (MFMA) | barrier + write-back + barrier | layer 3 + stores."""
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
c0, c1, f, h, w = 24, 48, 2, 1088, 1920
x = torch.zeros(f, h, w, 4)
x[..., :3] = torch.rand(f, h, w, 3, generator=g)
rnd = lambda *s, sc=1.0: ((torch.rand(*s, generator=g) * 2 - 1) * sc).to(dev)
pk = ops.pack_stem(rnd(c0, 3, 3, 3, sc=0.6), rnd(c0, sc=0.3), 0.25 + rnd(c0, sc=0.2), rnd(c0, c0, 3, 3, sc=0.25), rnd(c0, sc=0.3), 0.25 + rnd(c0, sc=0.2),
                   rnd(c1, c0, 3, 3, sc=0.25), rnd(c1, sc=0.3), 0.25 + rnd(c1, sc=0.2))
out = hip_ops.Planes.alloc(f * (h // 2) * (w // 2), c1, dev)
xd = x.to(dev)
buf = torch.zeros(256 * 8 * 8, dtype=torch.int64, device=dev)
ops.lib.atmvfi_debug_set_stem_stamp_buffer.argtypes = [ctypes.c_void_p]
ops.lib.atmvfi_debug_set_stem_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
for _ in range(30):
    ops.stem_fused(xd, pk, out)
torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20):
    ops.stem_fused(xd, pk, out)
e.record(); torch.cuda.synchronize()
print(f"stem_fused 2x{h}x{w}: {s.elapsed_time(e) / 20:.3f} ms per launch (stamp build)")
t = buf.reshape(-1, 8, 8).double()
t = t[t[:, 0, 6] > 0]
names = ["wait at tile start", "layer 1 (VALU)", "barrier", "layer 2 (MFMA)", "barrier + write-back + barrier", "layer 3 + stores"]
for wv in range(8):
    per = t[:, wv, :6] / t[:, wv, 6:7]
    ex = buf.reshape(-1, 8, 8)[buf.reshape(-1, 8, 8)[:, 0, 6] > 0][:, wv, 7]
    issue = ((ex >> 32).double() / t[:, wv, 6]).median().item()
    land = ((ex & 0xffffffff).double() / t[:, wv, 6]).median().item()
    print(f"wave {wv}: cycles per tile  " + "  ".join(f"{n} {per[:, i].median().item():.0f}" for i, n in enumerate(names)) + f"  | total {per.sum(1).median().item() + issue + land:.0f}"
          f"  [layer 1 before: gathers issued {issue:.0f}, landed {land:.0f}]")
