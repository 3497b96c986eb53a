"""Helpers to load the committed reference vectors (tests/golden, made by oracle/gen_golden.py)."""
import json
import os

import numpy as np
import torch

import pairs

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_manifest():
    with open(os.path.join(GOLD, "manifest.json")) as f:
        return json.load(f)


def e2e_cases():
    return [c for c in load_manifest()["cases"] if c["kind"] in pairs.PAIR_KINDS]


def demo_cases():
    return [c for c in load_manifest()["cases"] if c["kind"] == "demo_uint8"]


def load_npz(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def warp_taps(flow):
    """Integer tap origins (x0, y0) of a backward warp by ``flow`` [B,2,H,W] (flow_warp.py:50-60: sample at (x + fx, y + fy), taps
    floor .. floor + 1), as the kernels compute them (fp32 coordinate arithmetic)."""
    b, _, h, w = flow.shape
    xs = np.arange(w, dtype=np.float32)[None, None, :]
    ys = np.arange(h, dtype=np.float32)[None, :, None]
    x0 = np.floor(xs + flow[:, 0].astype(np.float32)).astype(np.int64)
    y0 = np.floor(ys + flow[:, 1].astype(np.float32)).astype(np.int64)
    return x0, y0


def taps_out_of_frame(flow):
    """[B,H,W] bool: pixels none of whose four taps lies inside the frame (they read zero: bilinear_sample's padding, flow_warp.py:26-47)."""
    b, _, h, w = flow.shape
    x0, y0 = warp_taps(flow)
    return (x0 < -1) | (x0 > w - 1) | (y0 < -1) | (y0 > h - 1)


def tiled_warp_fallback_tiles(flow, tile_w=32, tile_h=8, box_w=64, box_h=24):
    """How many 32 x 8 output tiles of a warp by ``flow`` [B,2,H,W] do NOT fit the LDS-staged source box of the tiled warp kernels
    (atm-vfi_amd/csrc/pointwise.hip, box_add / box_get: bounding box of the tile's in-frame taps, x origin rounded down to a multiple
    of 4, at most 64 x 24 pixels) and therefore take the per-tile gather fallback.  Host restatement of the kernel's decision, used to
    certify that a large-motion fixture really exercises that path."""
    b, _, h, w = flow.shape
    x0, y0 = warp_taps(flow)
    live = ~taps_out_of_frame(flow)
    n = 0
    for bi in range(b):
        for ty in range(0, h, tile_h):
            for tx in range(0, w, tile_w):
                m = live[bi, ty:ty + tile_h, tx:tx + tile_w]
                if not m.any():
                    continue
                xx, yy = x0[bi, ty:ty + tile_h, tx:tx + tile_w][m], y0[bi, ty:ty + tile_h, tx:tx + tile_w][m]
                xlo, xhi = max(int(xx.min()), 0), min(int(xx.max()) + 1, w - 1)
                ylo, yhi = max(int(yy.min()), 0), min(int(yy.max()) + 1, h - 1)
                ax0 = xlo & ~3
                nv = ((xhi - ax0) >> 2) + 1
                n += int(nv > box_w // 4 or yhi - ylo + 1 > box_h)
    return n


def case_inputs(c):
    fn = pairs.PAIR_KINDS[c["kind"]]
    im0, im1 = fn(c["B"], c["H"], c["W"], c["seed"])
    return im0, im1


def check_inputs_match(c, gold, im0, im1):
    """Inputs are regenerated from the seed; make sure they are the ones the reference saw."""
    s = gold["in_sums"]
    assert abs(im0.double().sum().item() - s[0]) < 1e-6 * max(1.0, abs(s[0]))
    assert abs(im1.double().sum().item() - s[1]) < 1e-6 * max(1.0, abs(s[1]))


def compare_e2e(out, gold, step, tol, tol_flow=None):
    """max|d| of every stored tensor; returns dict of errors (asserts on tol)."""
    tol_flow = tol if tol_flow is None else tol_flow
    errs = {}
    def sub(t):
        return t[..., ::step, ::step].detach().float().cpu().numpy()
    for key, val, tl in (("I_t", out["I_t"], tol), ("im_t0", out["im_t_list"][0], tol),
                         ("opt_flow_0", out["opt_flow_0"], tol_flow), ("opt_flow_1", out["opt_flow_1"], tol_flow),
                         ("occ_mask1", out["occ_mask1"], tol), ("I_t_0", out["I_t_0"], tol), ("I_t_1", out["I_t_1"], tol)):
        errs[key] = float(np.abs(sub(val) - gold[key]).max())
        assert errs[key] <= tl, f"{key}: max|d| {errs[key]:.3e} > {tl}"
    for key, val in (("im_t_coarse", out["im_t_list"][-1]), ("im0_warped_coarse", out["im0_warped_list"][-1])):
        errs[key] = float(np.abs(val.detach().float().cpu().numpy() - gold[key]).max())
        assert errs[key] <= tol, f"{key}: max|d| {errs[key]:.3e} > {tol}"
    return errs
