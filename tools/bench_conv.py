"""Run one conv layer shape repeatedly (for rocprofv3 PMC passes / A-B timing of the contraction kernels)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
hip_ops = importlib.import_module("atm-vfi_amd.hip_ops")
dev = torch.device("cuda:0")
ops = hip_ops.HipOps(dev)
H, W, cin, cout = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (1088, 1920, 101, 101)
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
r4 = lambda c: (c + 3) // 4 * 4
g = torch.Generator().manual_seed(0)
x = (torch.rand(1, H, W, r4(cin), generator=g) * 2 - 1).to(dev)
w = ((torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) / (3 * cin ** 0.5)).to(dev)
b = torch.zeros(cout, device=dev)
pw = ops.pack_weight(0, w)
y = torch.empty(1, H, W, r4(cout), device=dev)
for _ in range(2):
    ops.conv(x[..., :cin], pw, y[..., :cout], 1, 1, 1, b, b)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(reps):
    ops.conv(x[..., :cin], pw, y[..., :cout], 1, 1, 1, b, b)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / reps
print(f"conv3x3 {H}x{W} {cin}->{cout}: {ms:.3f} ms  {2.0*H*W*cin*cout*9/ms/1e9:.1f} TF/s")
