"""Phase timing of conv3x3_planes_kernel from the diagnostic (ATMVFI_STAMP) library: per-wave s_memtime sums, by wave group."""
import ctypes, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
hip_ops = importlib.import_module("atm-vfi_amd.hip_ops")
hip_ops.LIB_PATH = os.path.join(ROOT, "tools", "lib", os.environ.get("ATMVFI_STAMP_LIB", "libatmvfi_hip_stamp.so"))
hip_ops.load_library.__defaults__ = (hip_ops.LIB_PATH,)
dev = torch.device("cuda:0")
ops = hip_ops.HipOps(dev)
H, W, cin, cout = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (1088, 1920, 101, 101)
wn = int(sys.argv[5]) if len(sys.argv) > 5 else 0
r4 = lambda c: (c + 3) // 4 * 4
g = torch.Generator().manual_seed(0)
x = (torch.rand(1, H, W, r4(cin), generator=g) * 2 - 1).to(dev)
w = ((torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) / (3 * cin ** 0.5)).to(dev)
b = torch.zeros(cout, device=dev)
pw = ops.pack_weight(0, w)
y = torch.empty(1, H, W, r4(cout), device=dev)
xp = hip_ops.Planes.alloc(H * W, cin, dev)
ops.split_planes(x[..., :cin].flatten(0, 2), xp)
nblk = (((H + 15) // 16) * ((W + 15) // 16) + 7) // 8 * 8 * 8
buf = torch.zeros(nblk * 8 * 10, dtype=torch.int64, device=dev)
ops.lib.atmvfi_debug_set_planes_stamp_buffer.argtypes = [ctypes.c_void_p]
ops.lib.atmvfi_debug_set_planes_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
mode = os.environ.get("ATMVFI_STAMP_OUT", "f32")       # f32: fp32 rows; sink: a plane sink; sink+prelu: through its own PReLU; both: rows + sink
yp = hip_ops.Planes.alloc(H * W, cout, dev)
slope = torch.full(((cout + 31) // 32 * 32,), 0.25, device=dev)
kw = {"f32": dict(out=y[..., :cout]), "sink": dict(planes=yp), "sink+prelu": dict(planes=yp, planes_prelu=slope),
      "both": dict(out=y[..., :cout], planes=yp, planes_prelu=slope)}[mode]
print("outputs:", mode)
for _ in range(3):
    ops.conv3x3_planes(xp, 1, H, W, pw, bias=b, prelu=b, wn=wn, **kw)
torch.cuda.synchronize()
t = buf.reshape(-1, 8, 10).double()
t = t[t[:, 0, 9] > 0]
if os.environ.get("ATMVFI_STAMP_KSTEP"):      # a library built with -DATMVFI_STAMP -DATMVFI_STAMP_KSTEP (spills since the persistent grid)
    names = ["prologue", "fragment address + ds_read issue", "DMA issue", "vmcnt wait", "lgkmcnt(0) wait", "barrier after read phase",
             "MFMA phase", "barrier after MFMA phase", "epilogue"]
else:                                         # the default stamp build: whole tiles only (four s_memtime per tile)
    names = ["prologue (first tile only)", "k-loop", "boundary: partner's last MFMA + DMA wait", "next decode + re-stagger barrier",
             "-", "-", "-", "-", "epilogue (VALU + stores)"]
nk = t[0, 0, 9].item()
# spread over WORKGROUPS of the time per tile (wave 0 of each group; prologue excluded): with the static tile walk every workgroup gets the
# same number of tiles, so the kernel lasts as long as its slowest workgroup -- max / mean - 1 is what a dynamic tile queue could recover
per_wg = t[:, 0, 1:9].sum(1)
xcd = torch.arange(per_wg.numel()) % 8
print(f"per-tile ticks over {per_wg.numel()} workgroups: min {per_wg.min().item():.0f} mean {per_wg.mean().item():.0f} max {per_wg.max().item():.0f} "
      f"(max / mean - 1 = {100 * (per_wg.max() / per_wg.mean() - 1).item():.1f} %, p95 / mean - 1 = {100 * (per_wg.quantile(0.95) / per_wg.mean() - 1).item():.1f} %); "
      "by XCD (block % 8): " + " ".join(f"{per_wg[xcd == k].mean().item():.0f}" for k in range(8)))
for grp in (0, 1):
    tg = t[:, 4 * grp:4 * grp + 4, :9].reshape(-1, 9)
    tot = tg.sum(1).mean().item()
    print(f"{H}x{W} {cin}->{cout} wn {wn} group {grp}: {tg.shape[0]} waves, {nk:.0f} k-steps, mean ticks per wave {tot:.0f} (per k-step {tot / nk:.0f})")
    for k in range(9):
        if names[k] == "-": continue
        m = tg[:, k].mean().item()
        print(f"  {names[k]:34s} {m:10.0f}  {100 * m / tot:5.1f} %   per k-step {m / nk:7.0f}")
