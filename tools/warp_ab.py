"""Time atmvfi_warp_blend_planes in the shape of the forward's full-resolution call (all outputs + the refiner's plane sink) with the library
given on the command line:   python tools/warp_ab.py [path/to/lib.so ...]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
hip_ops = importlib.import_module("atm-vfi_amd.hip_ops")
dev = torch.device("cuda:0")
libs = sys.argv[1:] or [hip_ops.LIB_PATH]
B, H, W = 1, 1088, 1920
g = torch.Generator().manual_seed(0)
im0 = torch.rand(B, 3, H, W, generator=g).to(dev); im1 = torch.rand(B, 3, H, W, generator=g).to(dev)
for amp in (4.0, 24.0):
    # smooth flows of a few pixels (low-pass noise), mask logits
    low = torch.randn(B, 5, H // 32, W // 32, generator=g)
    mot = torch.nn.functional.interpolate(low, size=(H, W), mode="bilinear", align_corners=True) * amp
    motion = mot.permute(0, 2, 3, 1).contiguous().to(dev)
    for path in libs:
        hip_ops.LIB_PATH = path
        hip_ops.load_library.__defaults__ = (path,)
        if hasattr(hip_ops.load_library, "cache_clear"): hip_ops.load_library.cache_clear()
        ops = hip_ops.HipOps(dev)
        for tiles in (False, True):
          ops.warp_tiles = tiles
          outs = [torch.empty(B, 3, H, W, device=dev) for _ in range(3)]
          f0, f1 = (torch.empty(B, 2, H, W, device=dev) for _ in range(2))
          m1, m2 = (torch.empty(B, 1, H, W, device=dev) for _ in range(2))
          pl = hip_ops.Planes.alloc(B * H * W, 32, dev)
          def call():
              ops.warp_blend(im0, im1, motion, *outs, f0, f1, m1, m2, im0, im1, None, pack_planes=pl, pack_c0=8)
          for _ in range(5): call()
          torch.cuda.synchronize()
          s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
          s.record()
          for _ in range(50): call()
          e.record(); torch.cuda.synchronize()
          chk = float(outs[2].double().sum()) + float(pl.t[0].float().sum())
          print(f"flow amplitude ~{amp:4.0f} px  {os.path.basename(path):20s} {'LDS-staged tiles' if tiles else 'direct gathers  '} {s.elapsed_time(e) / 50 * 1e3:7.1f} us   checksum {chk:.6f}", flush=True)
