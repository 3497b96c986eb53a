"""CPU oracle for the ATM-VFI forward hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain fp32 PyTorch-CPU restatement of the reference algorithm
(``/root/reference/network/{network_base,network_lite,attention,flow_warp}.py``).
It exists to *check* the HIP path; only ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` may import it.  The product
(``atm-vfi_amd``) never does, and fails loudly when its HIP library is missing.

Parity status: PINNED.  ``oracle/gen_golden.py`` imports the reference itself
in the build container, loads identical weights into both and commits the
reference's outputs under ``tests/golden/``; ``tests/test_oracle_golden.py`` holds
this restatement to those vectors (max|d| <= 2e-5).

Style: purely functional over a ``state_dict`` (no module tree), so it can be fed
the very tensors the HIP path uses.  Each function cites the reference lines it
follows.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

NUM_HEADS = 8
MOTION_OUT = 5


# --------------------------------------------------------------------------- #
# elementary ops
# --------------------------------------------------------------------------- #
def conv_act(sd: SD, p: str, x: Tensor, stride: int = 1) -> Tensor:
    """``conv()`` = Conv2d(k3,p1)+PReLU  (network_base.py:20-25)."""
    y = F.conv2d(x, sd[f"{p}.0.weight"], sd[f"{p}.0.bias"], stride=stride, padding=1)
    return F.prelu(y, sd[f"{p}.1.weight"])


def deconv_act(sd: SD, p: str, x: Tensor) -> Tensor:
    """``deconv()`` = ConvTranspose2d(k2,s2,p0)+PReLU  (network_base.py:27-32)."""
    y = F.conv_transpose2d(x, sd[f"{p}.0.weight"], sd[f"{p}.0.bias"], stride=2)
    return F.prelu(y, sd[f"{p}.1.weight"])


def half_res(x: Tensor) -> Tensor:
    """Bilinear x0.5, align_corners=True (network_base.py:445)."""
    return F.interpolate(x, scale_factor=0.5, mode="bilinear", align_corners=True)


def upsample_flow(flow: Tensor, factor: int = 2) -> Tensor:
    """network_base.py:11-18: bilinear, align_corners=True, values scaled by the factor."""
    return F.interpolate(flow, scale_factor=factor, mode="bilinear", align_corners=True) * factor


def resize_ac_explicit(x: Tensor, oh: int, ow: int, value_scale: float = 1.0) -> Tensor:
    """Explicit align_corners=True bilinear resize: ``src = dst*(in-1)/(out-1)``
    (SURVEY.md E.5).  Kernel-level spec for ``atmvfi_resize_bilinear_ac``; checked
    against F.interpolate in the CPU tests."""
    b, c, ih, iw = x.shape
    def axis(o, i):
        if o == 1:
            src = torch.zeros(1)
        else:
            src = torch.arange(o, dtype=torch.float32) * (float(i - 1) / float(o - 1))
        i0 = src.floor().clamp_(0, i - 1).long()
        i1 = (i0 + 1).clamp_(max=i - 1)
        w1 = src - i0.float()
        return i0, i1, w1
    y0, y1, wy = axis(oh, ih)
    x0, x1, wx = axis(ow, iw)
    top = x[:, :, y0][:, :, :, x0] * (1 - wx) + x[:, :, y0][:, :, :, x1] * wx
    bot = x[:, :, y1][:, :, :, x0] * (1 - wx) + x[:, :, y1][:, :, :, x1] * wx
    return (top * (1 - wy)[:, None] + bot * wy[:, None]) * value_scale


def flow_warp(feature: Tensor, flow: Tensor, mask: bool = False, padding_mode: str = "zeros"):
    """Backward bilinear warp (flow_warp.py:50-60, 26-47, 7-23): pixel grid + flow, normalised with ``2*p/(dim-1)-1`` and sampled
    with ``grid_sample(bilinear, padding_mode, align_corners=True)``; zero padding is the forward's form.  ``mask=True`` also
    returns bilinear_sample's in-range mask (:42-45)."""
    b, c, h, w = feature.shape
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32),
                            torch.arange(w, dtype=torch.float32), indexing="ij")
    px = xs[None] + flow[:, 0]
    py = ys[None] + flow[:, 1]
    gx = 2 * px / (w - 1) - 1
    gy = 2 * py / (h - 1) - 1
    grid = torch.stack([gx, gy], dim=-1)
    out = F.grid_sample(feature, grid, mode="bilinear", padding_mode=padding_mode, align_corners=True)
    if mask:
        return out, (gx >= -1) & (gy >= -1) & (gx <= 1) & (gy <= 1)
    return out


def flow_warp_explicit(feature: Tensor, flow: Tensor) -> Tensor:
    """Same warp written as a direct 4-tap gather in pixel coordinates (what the HIP
    kernel does).  Differs from :func:`flow_warp` only by the normalise/un-normalise
    round trip, ~1e-6."""
    b, c, h, w = feature.shape
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32),
                            torch.arange(w, dtype=torch.float32), indexing="ij")
    px = xs[None] + flow[:, 0]
    py = ys[None] + flow[:, 1]
    x0 = px.floor()
    y0 = py.floor()
    fx = px - x0
    fy = py - y0
    out = torch.zeros_like(feature)
    flat = feature.reshape(b, c, h * w)
    for dy in (0, 1):
        for dx in (0, 1):
            xi = x0 + dx
            yi = y0 + dy
            wgt = (fx if dx else 1 - fx) * (fy if dy else 1 - fy)
            ok = (xi >= 0) & (xi <= w - 1) & (yi >= 0) & (yi <= h - 1)
            idx = (yi.clamp(0, h - 1) * w + xi.clamp(0, w - 1)).long().reshape(b, 1, h * w)
            g = torch.gather(flat, 2, idx.expand(b, c, h * w)).reshape(b, c, h, w)
            out = out + g * (wgt * ok)[:, None]
    return out


# --------------------------------------------------------------------------- #
# window machinery (attention.py:8-71, 275-305)
# --------------------------------------------------------------------------- #
def _region_ids(n: int, a: int, b: int) -> Tensor:
    """Label positions 0..n-1 with 0 for [0,a), 1 for [a,b), 2 for [b,n)."""
    i = torch.arange(n)
    return (i >= a).long() + (i >= b).long()


def window_labels(h: int, w: int, ws: int, shift: int) -> Tuple[Optional[Tensor], int, int, int, int]:
    """Per-token mask labels in window order ``[nW, N]`` such that the additive mask
    of the reference is ``-100 * (label_q != label_k)``.

    * pad part: 9 regions of the centre-padded canvas (attention.py:33-57), taken in
      UN-rolled window coordinates even when the data has been rolled (Appendix B.2);
    * shift part: Swin labelling with slices (0,-ws),(-ws,-s),(-s,) (attention.py:282-299).
    "different in either labelling" == "different combined label", so one integer
    per token carries both.  Returns (labels|None, Hp, Wp, pad_top, pad_left)."""
    pad_h = math.ceil(h / ws) * ws - h
    pad_w = math.ceil(w / ws) * ws - w
    hp, wp = h + pad_h, w + pad_w
    lab = torch.zeros(hp, wp, dtype=torch.long)
    used = False
    if pad_h > 0 or pad_w > 0:
        ry = _region_ids(hp, pad_h // 2, h + pad_h // 2)
        rx = _region_ids(wp, pad_w // 2, w + pad_w // 2)
        lab = lab + (ry[:, None] * 3 + rx[None, :])
        used = True
    if shift:
        sy = _region_ids(hp, hp - ws, hp - shift)
        sx = _region_ids(wp, wp - ws, wp - shift)
        lab = lab + 9 * (sy[:, None] * 3 + sx[None, :])
        used = True
    if not used:
        return None, hp, wp, pad_h // 2, pad_w // 2
    lab = lab.reshape(hp // ws, ws, wp // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    return lab, hp, wp, pad_h // 2, pad_w // 2


def to_windows(x: Tensor, ws: int, shift: int) -> Tuple[Tensor, Tuple[int, int, int, int]]:
    """[F,h,w,C] -> centre zero-pad, roll(-s,-s), window partition -> [F*nW, N, C]."""
    f, h, w, c = x.shape
    pad_h = math.ceil(h / ws) * ws - h
    pad_w = math.ceil(w / ws) * ws - w
    if pad_h or pad_w:
        x = F.pad(x, (0, 0, pad_w // 2, pad_w - pad_w // 2, pad_h // 2, pad_h - pad_h // 2))
    hp, wp = h + pad_h, w + pad_w
    if shift:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    xw = x.reshape(f, hp // ws, ws, wp // ws, ws, c).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, c)
    return xw, (hp, wp, pad_h // 2, pad_w // 2)


def from_windows(xw: Tensor, f: int, h: int, w: int, ws: int, shift: int, geo) -> Tensor:
    """Inverse of :func:`to_windows` (window_reverse, roll back, de-pad)."""
    hp, wp, pt, pl = geo
    c = xw.shape[-1]
    x = xw.reshape(f, hp // ws, wp // ws, ws, ws, c).permute(0, 1, 3, 2, 4, 5).reshape(f, hp, wp, c)
    if shift:
        x = torch.roll(x, shifts=(shift, shift), dims=(1, 2))
    return x[:, pt:pt + h, pl:pl + w, :]


def _attention_core(q: Tensor, k: Tensor, v: Tensor, labels: Optional[Tensor], ws: int,
                    want_motion: bool):
    """softmax(q k^T * hd^-0.5 + mask) v, plus per-head expected key offset
    (attention.py:189-208).  q,k,v: [Bw, heads, N, hd]."""
    bw, heads, n, hd = q.shape
    attn = (q @ k.transpose(-2, -1)) * (hd ** -0.5)
    if labels is not None:
        nw = labels.shape[0]
        mask = (labels[:, :, None] != labels[:, None, :]).float() * -100.0     # [nW,N,N]
        attn = (attn.reshape(bw // nw, nw, heads, n, n) + mask[None, :, None]).reshape(bw, heads, n, n)
    attn = attn.softmax(dim=-1)
    out = attn @ v
    motion = None
    if want_motion:
        idx = torch.arange(n)
        cx = (idx % ws).float()
        cy = (idx // ws).float()
        rel = torch.stack([cx[None, :] - cx[:, None], cy[None, :] - cy[:, None]])   # [2,N,N]: k - q
        motion = (attn[:, :, None] * rel[None, None]).sum(-1)                        # [Bw,heads,2,N]
    return out, motion


def mlp_block(sd: SD, p: str, x: Tensor, h: int, w: int) -> Tensor:
    """Mlp with depth-wise conv (attention.py:116-123, 79-85): fc1, dw3x3, GELU(erf), fc2."""
    f, l, c = x.shape
    y = F.linear(x, sd[f"{p}.fc1.weight"], sd[f"{p}.fc1.bias"])
    hid = y.shape[-1]
    y = y.transpose(1, 2).reshape(f, hid, h, w)
    y = F.conv2d(y, sd[f"{p}.dwconv.dwconv.weight"], sd[f"{p}.dwconv.dwconv.bias"], padding=1, groups=hid)
    y = y.reshape(f, hid, l).transpose(1, 2)
    y = F.gelu(y)
    return F.linear(y, sd[f"{p}.fc2.weight"], sd[f"{p}.fc2.bias"])


def atm_block(sd: SD, p: str, x: Tensor, ws: int, shift: int) -> Tuple[Tensor, Tensor]:
    """ATMFormer.forward (attention.py:265-334).  x: [2B,h,w,C] with frame-0 samples in
    the first half of dim 0.  Returns (x [2B,h*w,C], motion [2B,h*w,2])."""
    f, h, w, c = x.shape
    hd = c // NUM_HEADS
    labels, _, _, _, _ = window_labels(h, w, ws, shift)
    xw, geo = to_windows(x, ws, shift)
    xn = F.layer_norm(xw, (c,), sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"], 1e-5)
    half = xn.shape[0] // 2
    other = torch.cat([xn[half:], xn[:half]])                      # attention.py:318
    bw, n, _ = xn.shape
    q = F.linear(xn, sd[f"{p}.attn.q.weight"]).reshape(bw, n, NUM_HEADS, hd).permute(0, 2, 1, 3)
    kv = F.linear(other, sd[f"{p}.attn.kv.weight"]).reshape(bw, n, 2, NUM_HEADS, hd).permute(2, 0, 3, 1, 4)
    out, motion = _attention_core(q, kv[0], kv[1], labels, ws, True)
    out = out.transpose(1, 2).reshape(bw, n, c)
    out = F.linear(out, sd[f"{p}.attn.proj.weight"], sd[f"{p}.attn.proj.bias"])
    # head read-out of the motion (attention.py:209-211): MLP over the 8 heads, x and y separately
    m = motion.permute(0, 2, 3, 1)                                   # [Bw,2,N,heads]
    m = F.linear(m, sd[f"{p}.attn.mlp.0.weight"], sd[f"{p}.attn.mlp.0.bias"])
    m = F.gelu(m)
    m = F.linear(m, sd[f"{p}.attn.mlp.2.weight"], sd[f"{p}.attn.mlp.2.bias"])   # [Bw,2,N,1]
    m = m[..., 0].permute(0, 2, 1)                                   # [Bw,N,2]  (dx,dy)
    xr = xn + out                                                    # residual onto the NORMALISED tensor (:320)
    xs = from_windows(xr, f, h, w, ws, shift, geo).reshape(f, h * w, c)
    ms = from_windows(m, f, h, w, ws, shift, geo).reshape(f, h * w, 2)
    y = F.layer_norm(xs, (c,), sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"], 1e-5)
    xs = xs + mlp_block(sd, f"{p}.mlp", y, h, w)
    return xs, ms


def swin_block(sd: SD, p: str, x: Tensor, ws: int, shift: int) -> Tensor:
    """RefineBottleneck.forward (attention.py:433-495): self-attention twin of atm_block."""
    f, h, w, c = x.shape
    hd = c // NUM_HEADS
    labels, _, _, _, _ = window_labels(h, w, ws, shift)
    xw, geo = to_windows(x, ws, shift)
    xn = F.layer_norm(xw, (c,), sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"], 1e-5)
    bw, n, _ = xn.shape
    qkv = F.linear(xn, sd[f"{p}.attn.qkv.weight"]).reshape(bw, n, 3, NUM_HEADS, hd).permute(2, 0, 3, 1, 4)
    out, _ = _attention_core(qkv[0], qkv[1], qkv[2], labels, ws, False)
    out = out.transpose(1, 2).reshape(bw, n, c)
    out = F.linear(out, sd[f"{p}.attn.proj.weight"], sd[f"{p}.attn.proj.bias"])
    xs = from_windows(xn + out, f, h, w, ws, shift, geo).reshape(f, h * w, c)
    y = F.layer_norm(xs, (c,), sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"], 1e-5)
    return xs + mlp_block(sd, f"{p}.mlp", y, h, w)


def fusion(sd: SD, p: str, xs: List[Tensor]) -> Tuple[Tensor, int, int]:
    """CrossScaleFeatureFusion.forward (network_base.py:73-85).  xs fine->coarse (3 scales)."""
    ys = [F.conv2d(xs[1], sd[f"{p}.layers.0.weight"], sd[f"{p}.layers.0.bias"], stride=2, padding=1),
          F.conv2d(xs[0], sd[f"{p}.layers.1.weight"], sd[f"{p}.layers.1.bias"], stride=4, padding=1, dilation=1),
          F.conv2d(xs[0], sd[f"{p}.layers.2.weight"], sd[f"{p}.layers.2.bias"], stride=4, padding=2, dilation=2),
          xs[2]]
    x = F.conv2d(torch.cat(ys, 1), sd[f"{p}.proj.weight"], sd[f"{p}.proj.bias"])
    _, c, h, w = x.shape
    x = x.flatten(2).transpose(1, 2)
    return F.layer_norm(x, (c,), sd[f"{p}.norm.weight"], sd[f"{p}.norm.bias"], 1e-5), h, w


# --------------------------------------------------------------------------- #
# network
# --------------------------------------------------------------------------- #
def encoder(sd: SD, x: Tensor) -> Tuple[Tensor, List[Tensor]]:
    """shared_feat_extraction (network_base.py:342-352)."""
    feats = []
    for i in range(4):
        x = conv_act(sd, f"feat_extracts.{i}.0", x, stride=1 if i == 0 else 2)
        x = conv_act(sd, f"feat_extracts.{i}.1", x)
        if i:
            feats.append(x)
    return x, feats


def motion_head(sd: SD, p: str, feat: Tensor, motions: List[Tensor], b: int, h: int, w: int):
    """Tail of estimate_{local,global}_motion (network_base.py:379-389 / 406-415).
    feat [2B,h*w,C]; motions: per block [2B,h*w,2]."""
    c = feat.shape[-1]
    fcat = feat.reshape(2, b, h, w, c).permute(1, 0, 4, 2, 3).reshape(b, 2 * c, h, w)       # '(N B)(H W) C -> B (N C) H W'
    ms = [m.reshape(2, b, h * w, 2).permute(1, 2, 0, 3).reshape(b, h * w, 4) for m in motions]  # '(N B) L K -> B L (N K)'
    mo = torch.cat(ms, dim=2).reshape(b, h, w, 8).permute(0, 3, 1, 2)
    x = torch.cat([mo, fcat], 1)
    x = conv_act(sd, f"{p}.0", x)
    x = conv_act(sd, f"{p}.1", x)
    return F.conv2d(x, sd[f"{p}.2.weight"], sd[f"{p}.2.bias"])


def global_flows(sd: SD, enc_last: Tensor, feats: List[Tensor], ws: int) -> Tensor:
    """estimate_global_motion (network_base.py:391-415) -> raw 5-channel map at H/16."""
    g = conv_act(sd, "last_feat_extract.0", enc_last, stride=2)
    g = conv_act(sd, "last_feat_extract.1", g)
    x, h, w = fusion(sd, "global_feature_fusion", [feats[1], feats[2], g])
    f2 = x.shape[0]
    x = x.reshape(f2, h, w, -1)
    motions = []
    for blk in range(2):
        xs, m = atm_block(sd, f"global_motion_atmformer.{blk}", x, ws, 0 if blk == 0 else ws // 2)
        motions.append(m)
        x = xs.reshape(f2, h, w, -1)
    return motion_head(sd, "global_motion_mlp", xs, motions, f2 // 2, h, w)


def blend(i0: Tensor, i1: Tensor, logit: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    m1 = torch.sigmoid(logit)
    m2 = 1 - m1
    return m1 * i0 + m2 * i1, m1, m2


def refiner(sd: SD, feat: Tensor, im0, it0, im1, it1, it, skips: List[Tensor]) -> Tensor:
    """residual_refinement (network_base.py:417-431)."""
    f0 = conv_act(sd, "proj", torch.cat([feat, im0, it0, im1, it1, it], 1))
    f1 = conv_act(sd, "down1.0", f0, stride=2)
    x = conv_act(sd, "down2.0", torch.cat([f1, skips.pop()], 1), stride=2)
    f2 = conv_act(sd, "down2.1", x)
    x = conv_act(sd, "down3.0", torch.cat([f2, skips.pop()], 1), stride=2)
    x = conv_act(sd, "down3.1", x)
    f3 = conv_act(sd, "down3.2", x)
    u2 = conv_act(sd, "up1.1", deconv_act(sd, "up1.0", f3))
    u1 = conv_act(sd, "up2.1", deconv_act(sd, "up2.0", torch.cat([u2, f2], 1)))
    u0 = deconv_act(sd, "up3.0", torch.cat([u1, f1], 1))
    r = conv_act(sd, "refine_head.1", conv_act(sd, "refine_head.0", torch.cat([u0, f0], 1)))
    return 2 * torch.sigmoid(r) - 1


def _alignment_loss(flow0, flow1, im0, im1):
    """global_alignmentness (network_base.py:548-562)."""
    factor = im0.shape[2] // flow0.shape[2]
    a = flow_warp(im0, upsample_flow(flow0, factor))
    b = flow_warp(im1, upsample_flow(flow1, factor))
    return (a - b).abs().mean(dim=[1, 2, 3])


def ensemble_global_flows(sd: SD, im0: Tensor, im1: Tensor, ws: int, picks: Optional[list] = None) -> Tuple[Tensor, Tensor]:
    """multiscale_global_motion_ensemble (network_base.py:564-605).  ``picks``: a list that receives the level chosen per sample
    (the fixtures' manifest records it, so a test can tell which branches of :593-603 a fixture reaches)."""
    b = im0.shape[0]
    im = torch.cat([im0, im1], 0)
    levels, losses = [], []
    for lvl in range(3):
        if lvl:
            im = half_res(im)
        last, feats = encoder(sd, im)
        out = global_flows(sd, last, feats, ws)
        levels.append((out[:, :2], out[:, 2:4]))
        losses.append(_alignment_loss(out[:, :2], out[:, 2:4], im0, im1))
    f0 = torch.zeros_like(levels[0][0])
    f1 = torch.zeros_like(levels[0][1])
    for i in range(b):
        ls = [losses[k][i] for k in range(3)]
        mn = min(ls)
        pick = 0 if ls[0] == mn else (1 if ls[1] == mn else 2)
        if picks is not None:
            picks.append(pick)
        if pick == 0:
            f0[i], f1[i] = levels[0][0][i], levels[0][1][i]
        else:
            f0[i] = upsample_flow(levels[pick][0][i, None], 2 ** pick)
            f1[i] = upsample_flow(levels[pick][1][i, None], 2 ** pick)
    return f0, f1


@torch.no_grad()
def forward(sd: SD, im0: Tensor, im1: Tensor, global_motion: bool = True,
            ensemble_global_motion: bool = False, local_window: int = 8,
            global_window: int = 12) -> Dict[str, object]:
    """Network.forward (network_base.py:336-340) = forward_normal (:433-546) or
    forward_global_ensemble (:607-712).  Returns the same 10-entry dict."""
    b = im0.shape[0]
    pyr0, pyr1 = [im0], [im1]
    for _ in range(3):
        pyr0.append(half_res(pyr0[-1]))
        pyr1.append(half_res(pyr1[-1]))
    last, feats = encoder(sd, torch.cat([im0, im1], 0))
    feat, h, w = fusion(sd, "cross_scale_feature_fusion", feats)
    c = feat.shape[-1]
    it_list: List[Tensor] = []
    w0_list: List[Tensor] = []
    w1_list: List[Tensor] = []
    if global_motion:
        if ensemble_global_motion:
            gf0, gf1 = ensemble_global_flows(sd, im0, im1, global_window)
        else:
            gout = global_flows(sd, last, feats, global_window)
            gf0, gf1 = gout[:, :2], gout[:, 2:4]
            a = flow_warp(half_res(pyr0[-1]), gf0)
            bb = flow_warp(half_res(pyr1[-1]), gf1)
            it, _, _ = blend(a, bb, gout[:, 4:5])
            w0_list.insert(0, a); w1_list.insert(0, bb); it_list.insert(0, it)
        gf0, gf1 = upsample_flow(gf0), upsample_flow(gf1)
        fm = feat.reshape(2 * b, h, w, c).permute(0, 3, 1, 2)
        fm = torch.cat([flow_warp(fm[:b], gf0), flow_warp(fm[b:], gf1)], 0)
        x = fm.permute(0, 2, 3, 1)
        for i in (3, 2, 1, 0):
            pyr0[i] = flow_warp(pyr0[i], gf0)
            pyr1[i] = flow_warp(pyr1[i], gf1)
            if i:
                gf0, gf1 = upsample_flow(gf0), upsample_flow(gf1)
    else:
        x = feat.reshape(2 * b, h, w, c)
    # local motion (network_base.py:367-389)
    motions = []
    for blk in range(2):
        xs, m = atm_block(sd, f"local_motion_atmformer.{blk}", x, local_window,
                          0 if blk == 0 else local_window // 2)
        motions.append(m)
        x = xs.reshape(2 * b, h, w, c)
    out = motion_head(sd, "local_motion_mlp", xs, motions, b, h, w)
    # feature enhancement (:354-365)
    for blk in range(2):
        xs = swin_block(sd, f"feat_enhance_transformer.{blk}", x, 8, 0 if blk == 0 else 4)
        x = xs.reshape(2 * b, h, w, c)
    enh = xs.reshape(2, b, h, w, c).permute(1, 0, 4, 2, 3).reshape(b, 2 * c, h, w)
    fl0, fl1 = out[:, :2], out[:, 2:4]
    i0 = flow_warp(pyr0[3], fl0)
    i1 = flow_warp(pyr1[3], fl1)
    it, m1, m2 = blend(i0, i1, out[:, 4:5])
    w0_list.insert(0, i0); w1_list.insert(0, i1); it_list.insert(0, it)
    x = torch.cat([flow_warp(enh[:, :c], fl0), flow_warp(enh[:, c:], fl1), out], 1)
    skips: List[Tensor] = []
    for st, scale in enumerate((2, 1, 0)):
        p = f"upsample_pyramid.{st}"
        o = 0
        if st:
            x = F.prelu(x, sd[f"{p}.0.weight"])
            o = 1
        x = deconv_act(sd, f"{p}.{o}", x)
        x = conv_act(sd, f"{p}.{o + 1}", x)
        x = F.conv2d(x, sd[f"{p}.{o + 2}.weight"], sd[f"{p}.{o + 2}.bias"], padding=1)
        out = x[:, -MOTION_OUT:]
        fl0, fl1 = out[:, :2], out[:, 2:4]
        if scale:
            skips.append(x[:, :-MOTION_OUT])
        i0 = flow_warp(pyr0[scale], fl0)
        i1 = flow_warp(pyr1[scale], fl1)
        it, m1, m2 = blend(i0, i1, out[:, 4:5])
        w0_list.insert(0, i0); w1_list.insert(0, i1); it_list.insert(0, it)
    res = refiner(sd, x, im0, i0, im1, i1, it, skips)
    it_ref = it + res                       # the reference adds in place: im_t_list[0] is this tensor (:532)
    it_list[0] = it_ref
    return {"I_t": it_ref.clamp(0, 1), "im_t_list": it_list, "im0_warped_list": w0_list,
            "im1_warped_list": w1_list, "opt_flow_0": fl0, "opt_flow_1": fl1,
            "I_t_0": i0, "I_t_1": i1, "occ_mask1": m1, "occ_mask2": m2}


# --------------------------------------------------------------------------- #
# host boundary (demo_2x.py:54-87, benchmark/utils.py:57-80)
# --------------------------------------------------------------------------- #
def pad_amounts(ht: int, wd: int, divisor: int) -> Tuple[int, int, int, int]:
    ph = (((ht // divisor) + 1) * divisor - ht) % divisor
    pw = (((wd // divisor) + 1) * divisor - wd) % divisor
    return pw // 2, pw - pw // 2, ph // 2, ph - ph // 2      # left, right, top, bottom


def inference_2frame(sd: SD, img0, img1, isBGR: bool = True, divisor: int = 64, **kw):
    """uint8 HWC frames in, uint8 HWC interpolated frame out (demo_2x.py:54-87)."""
    import numpy as np
    if isBGR:
        img0 = img0[:, :, ::-1].copy()
        img1 = img1[:, :, ::-1].copy()
    t0 = (torch.tensor(img0.transpose(2, 0, 1)) / 255.).unsqueeze(0)
    t1 = (torch.tensor(img1.transpose(2, 0, 1)) / 255.).unsqueeze(0)
    l, r, t, bt = pad_amounts(t0.shape[-2], t0.shape[-1], divisor)
    t0 = F.pad(t0, (l, r, t, bt), mode="replicate")
    t1 = F.pad(t1, (l, r, t, bt), mode="replicate")
    pred = forward(sd, t0, t1, **kw)["I_t"][0]
    hh, ww = pred.shape[-2:]
    pred = pred[..., t:hh - bt, l:ww - r].numpy().transpose(1, 2, 0)
    pred = np.round(pred * 255).astype(np.uint8)
    if isBGR:
        pred = pred[:, :, ::-1].copy()
    return pred
