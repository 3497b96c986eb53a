"""Host-side window bookkeeping for the shifted-window blocks.

The reference pads, rolls, partitions, un-partitions, un-rolls and de-pads tensors on
every block (attention.py:273-331) and rebuilds its masks on the CPU (:28-62, :282-305).
All of that is a fixed permutation for a given (frames, h, w, window, shift), so it is
computed once here as two small int32 tables that the kernels consume:

* ``row_map[m]``  : image-order row (``f*h*w + y*w + x``) of window-order token ``m``, or -1
  for a zero-padded token.  Used as a *gather* map by LayerNorm and as a *scatter*
  map by the projection GEMM and the motion read-out.
* ``labels[w, n]``: region label per window token; the additive attention mask is
  ``-100 * (label_q != label_k)``.  One integer carries both the 9-region pad labelling
  (taken in UN-rolled window coordinates, exactly as the reference does -- SURVEY.md
  Appendix B.2) and the 9-region Swin shift labelling.

Cached per geometry (the reference caches on ``Hp*Wp`` only; keying
on the full geometry is the same thing for any single resolution).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import torch


@dataclass
class WindowGeometry:
    frames: int
    h: int
    w: int
    ws: int
    shift: int
    hp: int
    wp: int
    n_windows: int          # windows per frame
    tokens: int             # ws*ws
    row_map: torch.Tensor   # int32 [frames*n_windows*tokens]
    labels: Optional[torch.Tensor]   # int32 [n_windows, tokens] or None


def _three_way(n: int, a: int, b: int) -> torch.Tensor:
    i = torch.arange(n)
    return (i >= a).to(torch.int64) + (i >= b).to(torch.int64)


def build_window_geometry(frames: int, h: int, w: int, ws: int, shift: int) -> WindowGeometry:
    pad_h = math.ceil(h / ws) * ws - h
    pad_w = math.ceil(w / ws) * ws - w
    hp, wp = h + pad_h, w + pad_w
    top, left = pad_h // 2, pad_w // 2
    nwh, nww = hp // ws, wp // ws
    # window-order enumeration of the (rolled) canvas positions
    py = (torch.arange(nwh)[:, None] * ws + torch.arange(ws)[None, :])          # [nwh, ws]
    px = (torch.arange(nww)[:, None] * ws + torch.arange(ws)[None, :])          # [nww, ws]
    py = py[:, None, :, None].expand(nwh, nww, ws, ws)
    px = px[None, :, None, :].expand(nwh, nww, ws, ws)
    # roll(-shift): rolled[p] = canvas[(p + shift) % size]; canvas = centre-padded image
    sy = (py + shift) % hp - top
    sx = (px + shift) % wp - left
    ok = (sy >= 0) & (sy < h) & (sx >= 0) & (sx < w)
    spatial = torch.where(ok, sy * w + sx, torch.full_like(sy, -1)).reshape(-1)      # [nW*N]
    offs = torch.arange(frames)[:, None] * (h * w)
    row_map = torch.where(spatial[None, :] >= 0, spatial[None, :] + offs, torch.full((1, 1), -1, dtype=torch.int64))
    labels = None
    if pad_h or pad_w or shift:
        lab = torch.zeros(hp, wp, dtype=torch.int64)
        if pad_h or pad_w:
            lab = lab + _three_way(hp, top, h + top)[:, None] * 3 + _three_way(wp, left, w + left)[None, :]
        if shift:
            lab = lab + 9 * (_three_way(hp, hp - ws, hp - shift)[:, None] * 3 + _three_way(wp, wp - ws, wp - shift)[None, :])
        labels = lab[py, px].reshape(nwh * nww, ws * ws).to(torch.int32).contiguous()
    return WindowGeometry(frames, h, w, ws, shift, hp, wp, nwh * nww, ws * ws,
                          row_map.reshape(-1).to(torch.int32).contiguous(), labels)
