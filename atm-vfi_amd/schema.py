"""Variant dimensions and the state-dict contract of the ATM-VFI hot path.

The drop-in boundary is the Python ``Network`` surface; part of that
surface is the exact 236-entry ``state_dict`` (names, shapes, order) that published
checkpoints carry.  This module derives that schema from the two variant
descriptions instead of re-stating the reference's module tree:

* base: reference ``network/network_base.py:88-260``
* lite: reference ``network/network_lite.py:88-273`` (differs only in widths)

Every entry is ``ParamSpec(key, shape, kind, fan)``; ``kind`` selects the init rule
(reference ``network/attention.py:172-185`` for the modules that call
``_init_weights``; PyTorch defaults elsewhere).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch

MOTION_OUT = 5          # flow0(2) flow1(2) mask-logit(1): network_base.py:153
NUM_HEADS = 8           # network_base.py:119,173
PYRAMID_LEVELS = 4      # network_base.py:91


@dataclass(frozen=True)
class Variant:
    name: str
    hidden_dims: Tuple[int, int, int, int]
    mlp_ratio: int            # transformer MLP expansion (base 4: attention.py:217 default; lite 2)
    local_mlp_scale: float    # hidden of local_motion_mlp = int(2*fused*scale)
    last_extra: int           # last_feat_dim = hidden_dims[3] + last_extra
    global_mlp_hidden: int    # 768 for base (network_base.py:191); concat_dim for lite
    refine_hidden: int        # 64 base / 32 lite
    local_window: int = 8
    global_window: int = 12

    # ---- derived widths ----
    @property
    def local_dim(self) -> int:      # network_base.py:113
        d = self.hidden_dims
        return d[3] + d[2] + 2 * d[1]

    @property
    def last_feat_dim(self) -> int:  # network_base.py:162
        return self.hidden_dims[3] + self.last_extra

    @property
    def global_dim(self) -> int:     # network_base.py:168
        d = self.hidden_dims
        return self.last_feat_dim + d[3] + 2 * d[2]

    @property
    def fused_dim(self) -> int:      # network_base.py:152
        return 2 * self.local_dim

    @property
    def local_mlp_hidden(self) -> int:   # network_base.py:154
        return int(self.fused_dim * self.local_mlp_scale)

    @property
    def decoder_widths(self) -> Tuple[int, int, int]:   # network_base.py:198-200
        f = self.fused_dim
        return (f // 2, f // 4, f // 8)

    @property
    def refine_in(self) -> int:      # network_base.py:224
        return self.decoder_widths[2] + MOTION_OUT + 15


VARIANTS: Dict[str, Variant] = {
    "base": Variant("base", (24, 48, 96, 192), 4, 0.75, 96, 768, 64),
    "lite": Variant("lite", (16, 32, 64, 96), 2, 0.5, 32, 352, 32),
}


@dataclass(frozen=True)
class ParamSpec:
    key: str
    shape: Tuple[int, ...]
    kind: str      # see init_tensor()
    fan: int = 0   # fan_in (default-init kinds) or fan_out (fanout kinds)
    is_buffer: bool = False


def relative_coord_table(ws: int) -> torch.Tensor:
    """``[1,1,2,N,N]`` table with ``R[0,q,k] = kx-qx`` and ``R[1,q,k] = ky-qy``
    (reference attention.py:150-165; verified in SURVEY.md Appendix E.5)."""
    idx = torch.arange(ws * ws)
    x = (idx % ws).float()
    y = (idx // ws).float()
    rx = x[None, :] - x[:, None]
    ry = y[None, :] - y[:, None]
    return torch.stack([rx, ry])[None, None]


def _conv_act(out: List[ParamSpec], prefix: str, cin: int, cout: int, k: int = 3):
    """``conv()`` helper of the reference: Conv2d (default init) + PReLU."""
    fan_in = cin * k * k
    out.append(ParamSpec(f"{prefix}.0.weight", (cout, cin, k, k), "default_w", fan_in))
    out.append(ParamSpec(f"{prefix}.0.bias", (cout,), "default_b", fan_in))
    out.append(ParamSpec(f"{prefix}.1.weight", (cout,), "prelu"))


def _plain_conv(out, prefix, cin, cout, k):
    fan_in = cin * k * k
    out.append(ParamSpec(f"{prefix}.weight", (cout, cin, k, k), "default_w", fan_in))
    out.append(ParamSpec(f"{prefix}.bias", (cout,), "default_b", fan_in))


def _deconv_act(out, prefix, cin, cout):
    """``deconv()``: ConvTranspose2d(k2,s2) weight (Cin,Cout,2,2) + PReLU.
    PyTorch's fan_in for this layout is ``Cout*4``."""
    fan_in = cout * 4
    out.append(ParamSpec(f"{prefix}.0.weight", (cin, cout, 2, 2), "default_w", fan_in))
    out.append(ParamSpec(f"{prefix}.0.bias", (cout,), "default_b", fan_in))
    out.append(ParamSpec(f"{prefix}.1.weight", (cout,), "prelu"))


def _fusion(out, prefix, in_dims, fused):
    """CrossScaleFeatureFusion (network_base.py:34-85): every conv here is
    re-initialised by ``_init_weights`` (fan_out normal, zero bias)."""
    k = 0
    for i in range(len(in_dims) - 1):
        c = in_dims[-2 - i]
        for _ in range(2 ** i):
            out.append(ParamSpec(f"{prefix}.layers.{k}.weight", (c, c, 3, 3), "fanout_w", 9 * c))
            out.append(ParamSpec(f"{prefix}.layers.{k}.bias", (c,), "zero"))
            k += 1
    out.append(ParamSpec(f"{prefix}.proj.weight", (fused, fused, 1, 1), "fanout_w", fused))
    out.append(ParamSpec(f"{prefix}.proj.bias", (fused,), "zero"))
    out.append(ParamSpec(f"{prefix}.norm.weight", (fused,), "one"))
    out.append(ParamSpec(f"{prefix}.norm.bias", (fused,), "zero"))


def _mlp(out, prefix, dim, ratio):
    hid = int(dim * ratio)
    out.append(ParamSpec(f"{prefix}.fc1.weight", (hid, dim), "trunc02"))
    out.append(ParamSpec(f"{prefix}.fc1.bias", (hid,), "zero"))
    out.append(ParamSpec(f"{prefix}.dwconv.dwconv.weight", (hid, 1, 3, 3), "fanout_w", 9))
    out.append(ParamSpec(f"{prefix}.dwconv.dwconv.bias", (hid,), "zero"))
    out.append(ParamSpec(f"{prefix}.fc2.weight", (dim, hid), "trunc02"))
    out.append(ParamSpec(f"{prefix}.fc2.bias", (dim,), "zero"))


def _swin_block(out, prefix, dim, ratio):
    """RefineBottleneck (attention.py:393-431)."""
    out.append(ParamSpec(f"{prefix}.norm1.weight", (dim,), "one"))
    out.append(ParamSpec(f"{prefix}.norm1.bias", (dim,), "zero"))
    out.append(ParamSpec(f"{prefix}.attn.qkv.weight", (3 * dim, dim), "trunc02"))
    out.append(ParamSpec(f"{prefix}.attn.proj.weight", (dim, dim), "trunc02"))
    out.append(ParamSpec(f"{prefix}.attn.proj.bias", (dim,), "zero"))
    out.append(ParamSpec(f"{prefix}.norm2.weight", (dim,), "one"))
    out.append(ParamSpec(f"{prefix}.norm2.bias", (dim,), "zero"))
    _mlp(out, f"{prefix}.mlp", dim, ratio)


def _atm_block(out, prefix, dim, ratio, ws):
    """ATMFormer (attention.py:216-253) with AttentionToMotion (attention.py:126-148)."""
    n = ws * ws
    out.append(ParamSpec(f"{prefix}.norm1.weight", (dim,), "one"))
    out.append(ParamSpec(f"{prefix}.norm1.bias", (dim,), "zero"))
    out.append(ParamSpec(f"{prefix}.attn.relative_coord", (1, 1, 2, n, n), f"relcoord{ws}", is_buffer=True))
    out.append(ParamSpec(f"{prefix}.attn.q.weight", (dim, dim), "trunc02"))
    out.append(ParamSpec(f"{prefix}.attn.kv.weight", (2 * dim, dim), "trunc02"))
    out.append(ParamSpec(f"{prefix}.attn.proj.weight", (dim, dim), "trunc02"))
    out.append(ParamSpec(f"{prefix}.attn.proj.bias", (dim,), "zero"))
    out.append(ParamSpec(f"{prefix}.attn.mlp.0.weight", (NUM_HEADS // 2, NUM_HEADS), "trunc02"))
    out.append(ParamSpec(f"{prefix}.attn.mlp.0.bias", (NUM_HEADS // 2,), "zero"))
    out.append(ParamSpec(f"{prefix}.attn.mlp.2.weight", (1, NUM_HEADS // 2), "trunc02"))
    out.append(ParamSpec(f"{prefix}.attn.mlp.2.bias", (1,), "zero"))
    out.append(ParamSpec(f"{prefix}.norm2.weight", (dim,), "one"))
    out.append(ParamSpec(f"{prefix}.norm2.bias", (dim,), "zero"))
    _mlp(out, f"{prefix}.mlp", dim, ratio)


def param_schema(v: Variant) -> List[ParamSpec]:
    """The 236 state-dict entries in the reference's registration order."""
    d = v.hidden_dims
    s: List[ParamSpec] = []
    # encoder: network_base.py:99-110
    for i in range(PYRAMID_LEVELS):
        cin = 3 if i == 0 else d[i - 1]
        _conv_act(s, f"feat_extracts.{i}.0", cin, d[i])
        _conv_act(s, f"feat_extracts.{i}.1", d[i], d[i])
    _fusion(s, "cross_scale_feature_fusion", list(d[1:]), v.local_dim)
    for b in range(2):
        _swin_block(s, f"feat_enhance_transformer.{b}", v.local_dim, v.mlp_ratio)
    for b in range(2):
        _atm_block(s, f"local_motion_atmformer.{b}", v.local_dim, v.mlp_ratio, v.local_window)
    hid = v.local_mlp_hidden
    _conv_act(s, "local_motion_mlp.0", v.fused_dim + NUM_HEADS, hid)
    _conv_act(s, "local_motion_mlp.1", hid, hid)
    _plain_conv(s, "local_motion_mlp.2", hid, MOTION_OUT, 1)
    # global branch: network_base.py:161-196
    _conv_act(s, "last_feat_extract.0", d[3], v.last_feat_dim)
    _conv_act(s, "last_feat_extract.1", v.last_feat_dim, v.last_feat_dim)
    _fusion(s, "global_feature_fusion", [d[2], d[3], v.last_feat_dim], v.global_dim)
    for b in range(2):
        _atm_block(s, f"global_motion_atmformer.{b}", v.global_dim, v.mlp_ratio, v.global_window)
    gh = v.global_mlp_hidden
    _conv_act(s, "global_motion_mlp.0", 2 * v.global_dim + NUM_HEADS, gh)
    _conv_act(s, "global_motion_mlp.1", gh, gh)
    _plain_conv(s, "global_motion_mlp.2", gh, MOTION_OUT, 1)
    # decoder: network_base.py:198-221
    widths = [v.fused_dim + MOTION_OUT] + [w + MOTION_OUT for w in v.decoder_widths]
    for st in range(3):
        cin, cout = widths[st], widths[st + 1]
        p = f"upsample_pyramid.{st}"
        o = 0
        if st > 0:
            s.append(ParamSpec(f"{p}.0.weight", (cin,), "prelu"))
            o = 1
        _deconv_act(s, f"{p}.{o}", cin, cout)
        _conv_act(s, f"{p}.{o + 1}", cout, cout)
        _plain_conv(s, f"{p}.{o + 2}", cout, cout, 3)
    # residual refinement U-Net: network_base.py:223-260
    h = v.refine_hidden
    w1, w2, _ = v.decoder_widths
    _conv_act(s, "proj", v.refine_in, h)
    _conv_act(s, "down1.0", h, h)
    _conv_act(s, "down2.0", w2 + h, 2 * h)
    _conv_act(s, "down2.1", 2 * h, 2 * h)
    _conv_act(s, "down3.0", w1 + 2 * h, 4 * h)
    _conv_act(s, "down3.1", 4 * h, 4 * h)
    _conv_act(s, "down3.2", 4 * h, 4 * h)
    _deconv_act(s, "up1.0", 4 * h, 2 * h)
    _conv_act(s, "up1.1", 2 * h, 2 * h)
    _deconv_act(s, "up2.0", 4 * h, 2 * h)
    _conv_act(s, "up2.1", 2 * h, h)
    _deconv_act(s, "up3.0", 2 * h, h)
    _conv_act(s, "refine_head.0", 2 * h, h)
    _conv_act(s, "refine_head.1", h, 3)
    return s


def init_tensor(spec: ParamSpec, gen: torch.Generator) -> torch.Tensor:
    """Reference-distribution init.  Seeded, but not the
    reference's RNG stream: parity tests load one state dict into both sides."""
    k = spec.kind
    if k == "default_w" or k == "default_b":
        bound = 1.0 / math.sqrt(spec.fan)          # kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(fan_in))
        return (torch.rand(spec.shape, generator=gen) * 2 - 1) * bound
    if k == "prelu":
        return torch.full(spec.shape, 0.25)
    if k == "fanout_w":
        return torch.randn(spec.shape, generator=gen) * math.sqrt(2.0 / spec.fan)
    if k == "trunc02":
        t = torch.randn(spec.shape, generator=gen) * 0.02
        return t.clamp_(-2.0, 2.0)                 # trunc_normal_(std=.02) cuts at +-2 (100 sigma)
    if k == "zero":
        return torch.zeros(spec.shape)
    if k == "one":
        return torch.ones(spec.shape)
    if k.startswith("relcoord"):
        return relative_coord_table(int(k[len("relcoord"):]))
    raise ValueError(k)


def reference_init_state_dict(variant: str, seed: int = 0) -> Dict[str, torch.Tensor]:
    v = VARIANTS[variant]
    gen = torch.Generator().manual_seed(seed)
    return {sp.key: init_tensor(sp, gen) for sp in param_schema(v)}


# Gains that bring the random-weight network into a regime where every stage matters
# (flows of about a pixel with spatial variation, occlusion masks away from 0.5, a
# visible refinement residual).  Tuned once against the CPU oracle on smooth 128x192
# frame pairs; (flow gain, flow-bias std, mask gain) per motion head.
_STRESS_GAINS = {
    "lite": {"global": (2.0, 0.10, 12.0), "local": (7.0, 0.5, 12.0),
             "dec": ((14.0, 0.7, 25.0), (130.0, 0.8, 160.0), (33.0, 1.0, 11.0)), "residual": 20.0},
    "base": {"global": (0.75, 0.10, 12.0), "local": (5.0, 0.5, 9.0),
             "dec": ((14.0, 0.7, 20.0), (80.0, 0.8, 110.0), (45.0, 1.0, 35.0)), "residual": 20.0},
}


def synthetic_state_dict(variant: str, seed: int = 0, motion_gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Deterministic *stress* weights for parity tests and the benchmark.

    Reference init leaves every bias at zero, every PReLU at 0.25 and the flows at
    ~0.03 px, which would leave the bias/PReLU/LayerNorm-affine/warp
    paths almost untested.  Here biases, LN affine terms and PReLU slopes are
    randomised and the motion heads are scaled (``_STRESS_GAINS``) so flows reach
    about a pixel at every pyramid level without pushing content out of frame.

    ``motion_gain`` > 1 is the *large-motion* weight set: the four flow rows (weights and biases) of every motion head -- global
    (network_base.py:391-415), local (:367-389) and the three decoder stages (:511-521) -- are multiplied by it on top of the stress
    gains, so the flows reach tens of pixels at full resolution: the regime the global branch exists for (:457-485), where the tiled
    warps overflow their staged bounding boxes and taps leave the frame.  Everything else (and the RNG stream) is unchanged."""
    v = VARIANTS[variant]
    gen = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for sp in param_schema(v):
        t = init_tensor(sp, gen)
        if sp.kind == "zero":
            t = torch.randn(sp.shape, generator=gen) * 0.05
        elif sp.kind == "one":
            t = 1.0 + torch.randn(sp.shape, generator=gen) * 0.1
        elif sp.kind == "prelu":
            t = 0.1 + 0.3 * torch.rand(sp.shape, generator=gen)
        elif sp.kind == "default_b":
            t = torch.randn(sp.shape, generator=gen) * 0.05
        sd[sp.key] = t
    g = _STRESS_GAINS[variant]

    def scale_head(wkey: str, bkey: str, gains):
        fg, fb, mg = gains
        sd[wkey][-MOTION_OUT:-1] *= fg * motion_gain
        sd[wkey][-1] *= mg
        sd[bkey][-MOTION_OUT:-1] = torch.randn(4, generator=gen) * (fb * motion_gain)

    scale_head("global_motion_mlp.2.weight", "global_motion_mlp.2.bias", g["global"])
    scale_head("local_motion_mlp.2.weight", "local_motion_mlp.2.bias", g["local"])
    for st, idx in ((0, 2), (1, 3), (2, 3)):
        scale_head(f"upsample_pyramid.{st}.{idx}.weight", f"upsample_pyramid.{st}.{idx}.bias", g["dec"][st])
    sd["refine_head.1.0.weight"] *= g["residual"]
    # make the motion read-out of the attention heads non-trivial
    for br in ("local_motion_atmformer", "global_motion_atmformer"):
        for b in range(2):
            sd[f"{br}.{b}.attn.mlp.0.weight"] = torch.randn(4, 8, generator=gen) * 0.5
            sd[f"{br}.{b}.attn.mlp.2.weight"] = torch.randn(1, 4, generator=gen) * 0.5
            # sharper attention so the expected offset is not ~0
            sd[f"{br}.{b}.attn.q.weight"] *= 12.0
            sd[f"{br}.{b}.attn.kv.weight"] *= 12.0
    return sd


def schema_signature(variant: str) -> List[Tuple[str, Tuple[int, ...]]]:
    return [(sp.key, tuple(sp.shape)) for sp in param_schema(VARIANTS[variant])]
