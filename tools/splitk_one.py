import importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
H = importlib.import_module("atm-vfi_amd.hip_ops")
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
ops = H.HipOps(dev)
n, h, w, cin, cout = [int(x) for x in sys.argv[1:6]]
wn = int(sys.argv[6]) if len(sys.argv) > 6 else 0
g = torch.Generator().manual_seed(1)
xp = H.Planes.alloc(n * h * w, cin, dev)
xp.t.copy_((torch.rand(xp.t.shape, generator=g) - 0.5).half()); xp.t[:, :, xp.rows:] = 0
wt = ((torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) / np.sqrt(9 * cin)).to(dev)
bias = torch.rand(cout, generator=g).to(dev); slope = (torch.rand(cout, generator=g) * 0.4).to(dev)
pw = ops.pack_weight(H.GEMM_CONV, wt)
sink = H.Planes.alloc(n * h * w, cout, dev)
ws = torch.empty(8 * n * h * w * ((cout + 15) // 16 * 16), device=dev)
for i in range(20):
    ops.conv3x3_planes(xp, n, h, w, pw, out=None, bias=bias, prelu=slope, planes=sink, wn=wn)
torch.cuda.synchronize()
for i in range(20):
    ops.conv3x3_planes(xp, n, h, w, pw, out=None, bias=bias, prelu=slope, planes=sink, wn=wn, workspace=ws)
torch.cuda.synchronize()
