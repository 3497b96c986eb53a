import importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
H = importlib.import_module("atm-vfi_amd.hip_ops")
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
ops = H.HipOps(dev)
g = torch.Generator().manual_seed(1)
# stride-2 conv M4096 N64 K1296 (C1's down2.0: 144 -> 64 at 128x128 -> 64x64) and M1024 N128 K2592
for (hh, cin, cout) in ((128, 144, 64), (64, 288, 128)):
    xp = H.Planes.alloc(hh * hh, cin, dev)
    xp.t.copy_((torch.rand(xp.t.shape, generator=g) - 0.5).half()); xp.t[:, :, xp.rows:] = 0
    w = ops.pack_weight(H.GEMM_CONV, ((torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) / np.sqrt(9 * cin)).to(dev))
    sink = H.Planes.alloc((hh // 2) ** 2, cout, dev)
    for mode in (None, lambda n: torch.empty(n, device=dev)):
        ops.gemm_workspace = mode
        for i in range(10):
            ops.conv_planes(xp, 1, hh, hh, w, stride=2, pad=1, sink=sink)
        torch.cuda.synchronize()
