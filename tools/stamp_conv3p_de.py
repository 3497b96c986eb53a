"""Per-k-step phase stamps of the deferred-epilogue 3x3 plane kernel (stamp build: make -C atm-vfi_amd/csrc stamp): for each k-step of a
workgroup's second tile, the wave's read phase (fragment reads, DMA issue, waits), its wait at the barrier, its MFMA phase (with the
previous tile's epilogue atoms in the first chunk) and the wait at the closing barrier -- s_memtime ticks, mean over workgroups, for the
first wave of each group.
  python tools/stamp_conv3p_de.py N H W Cin Cout [kind]      kind = lean | mid | last"""
import ctypes, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
hip_ops = importlib.import_module("atm-vfi_amd.hip_ops")
hip_ops.LIB_PATH = os.path.join(ROOT, "tools", "lib", os.environ.get("ATMVFI_STAMP_LIB", "libatmvfi_hip_stamp.so"))
hip_ops.load_library.__defaults__ = (hip_ops.LIB_PATH,)
dev = torch.device("cuda:0")
ops = hip_ops.HipOps(dev)
N, H, W, cin, cout = (int(v) for v in sys.argv[1:6])
kind = sys.argv[6] if len(sys.argv) > 6 else "lean"
g = torch.Generator().manual_seed(3)
r4 = lambda c: (c + 3) // 4 * 4
x = ((torch.rand(N, H, W, r4(cin), generator=g) * 2 - 1) * 1.5).to(dev)
wt = ((torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) / (3 * cin ** 0.5)).to(dev)
bias = ((torch.rand(cout, generator=g) * 2 - 1) * 0.2).to(dev)
slope = (torch.rand(cout, generator=g) * 0.4).to(dev)
pslope = torch.zeros((cout + 31) // 32 * 32, device=dev); pslope[:cout] = torch.rand(cout, generator=g).to(dev)
pw = ops.pack_weight(0, wt)
xp = hip_ops.Planes.alloc(N * H * W, cin, dev)
ops.split_planes(x[..., :cin].flatten(0, 2), xp)
cmin = (cout - 5) // 4 * 4 if cout > 8 else 0
s1 = hip_ops.Planes.alloc(N * H * W, cout, dev); s2 = hip_ops.Planes.alloc(N * H * W, cout, dev)
yc = torch.zeros((N, H, W, 8), device=dev)
nblk = 256
buf = torch.zeros(nblk * 8 * 128, dtype=torch.int32, device=dev)
ops.lib.atmvfi_debug_set_planes_stamp_buffer.argtypes = [ctypes.c_void_p]
def go(defer=True):
    if kind == "lean":
        ops.conv3x3_planes(xp, N, H, W, pw, out=None, bias=bias, prelu=slope, planes=s1, defer=defer)
    elif kind == "mid":
        ops.conv3x3_planes(xp, N, H, W, pw, out=yc[..., :cout - cmin], bias=bias, prelu=None, planes=s1, planes_prelu=pslope, planes2=s2, out_cmin=cmin, defer=defer)
    else:
        ops.conv3x3_planes(xp, N, H, W, pw, out=yc[..., :cout - cmin], bias=bias, prelu=None, planes=s1, out_cmin=cmin, defer=defer)
ops.lib.atmvfi_debug_set_planes_stamp_buffer(ctypes.c_void_p(0))
for _ in range(3):
    go()
torch.cuda.synchronize()
ops.lib.atmvfi_debug_set_planes_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
go()
torch.cuda.synchronize()
ops.lib.atmvfi_debug_set_planes_stamp_buffer(ctypes.c_void_p(0))
st = buf.view(nblk, 8, 128).cpu().to(torch.int64) & 0xffffffff
nfull = (cin - cin % 32) // 32 if 1 <= cin % 32 <= 8 else (cin + 31) // 32
tail = 1 <= cin % 32 <= 8
print(f"{N}x{H}x{W} {cin}->{cout} {kind}: {nfull} full chunks{' + tail' if tail else ''}; ticks per phase, mean over {nblk} workgroups (second tile of each)")
ksteps = [("first", t, t) for t in range(9)] + ([("loop", t, 9 + t) for t in range(9)] if nfull > 1 else []) + ([("tail", t, 18 + t) for t in range(3)] if tail else [])
for wv, name in ((0, "group A"), (4, "group B")):
    print(f"  {name} (wave {wv}):   read phase | barrier wait | MFMA phase | barrier wait | k-step total")
    for pos, (cname, t, k) in enumerate(ksteps):
        s = st[:, wv, 4 * k:4 * k + 4]
        ok = (s[:, 0] != 0)
        d = [((s[:, i + 1] - s[:, i]) & 0xffffffff)[ok].double().mean().item() for i in range(3)]
        if pos + 1 < len(ksteps) and not (cname == "loop" and t == 8 and nfull > 2):
            s_n = st[:, wv, 4 * ksteps[pos + 1][2]]
            d.append((((s_n - s[:, 3]) & 0xffffffff)[ok]).double().mean().item())
        else:
            d.append(float("nan"))
        print(f"    {cname:5s} T={t}:   {d[0]:8.0f}   {d[1]:8.0f}   {d[2]:8.0f}   {d[3]:8.0f}   {sum(v for v in d if v == v):8.0f}")
