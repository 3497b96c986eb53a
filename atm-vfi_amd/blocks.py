"""Stand-alone ``ATMFormer`` / ``RefineBottleneck`` modules: the import surface of the reference's ``network/attention.py``
(``from network.attention import ATMFormer``, ``network_base.py:8-9``) on the MI355X kernels.

Same constructor arguments, parameter / buffer names, ``_set_window_size_`` and ``forward`` signatures as ``attention.py:216-334``
(ATMFormer) and ``:393-495`` (RefineBottleneck); the arithmetic is ``BlockRunner._block`` of ``network.py`` -- LayerNorm-gather,
fused projection GEMM, ``atmvfi_window_attention``, projection-scatter GEMM, motion head, MLP with dw-conv -- i.e. exactly what
``Network.forward`` runs for its six transformer blocks.  No CPU path: inputs must be CUDA (HIP) tensors."""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn as nn

from . import schema as S
from .hip_ops import GEMM_LINEAR, HipOps
from .network import BlockRunner


class _Node(nn.Module):
    pass


class _Block(BlockRunner, nn.Module):
    CROSS = True

    def __init__(self, dim, window_size=7, shift_size=0, patch_size=1, num_heads=8, mlp_ratio=4., bidirectional=True, qkv_bias=False,
                 qk_scale=None, drop=0., attn_drop=0., drop_path=0., act_layer=None, norm_layer=None):
        super().__init__()
        if qkv_bias or qk_scale is not None or drop or attn_drop or drop_path:
            raise NotImplementedError("only the configuration the reference's networks use (no qkv bias, no dropout) is provided")
        if num_heads != S.NUM_HEADS or dim % num_heads:
            raise NotImplementedError(f"num_heads must be {S.NUM_HEADS} (the motion head MLP is Linear(8, 4) -> Linear(4, 1))")
        self._init_runner()
        self.dim, self.num_heads, self.mlp_ratio = dim, num_heads, mlp_ratio
        self.window_size = (window_size, window_size) if not isinstance(window_size, (tuple, list)) else tuple(window_size)
        self.shift_size = (shift_size, shift_size) if not isinstance(shift_size, (tuple, list)) else tuple(shift_size)
        specs = []
        if self.CROSS:
            S._atm_block(specs, "b", dim, mlp_ratio, self.window_size[0])
        else:
            S._swin_block(specs, "b", dim, mlp_ratio)
        gen = torch.Generator().manual_seed(torch.initial_seed() % (2 ** 31))
        for sp in specs:
            t = S.init_tensor(sp, gen)
            parts = sp.key.split(".")[1:]
            node: nn.Module = self
            for name in parts[:-1]:
                if name not in node._modules:
                    node.add_module(name, _Node())
                node = node._modules[name]
            if sp.is_buffer:
                node.register_buffer(parts[-1], t)
            else:
                node.register_parameter(parts[-1], nn.Parameter(t))
        self._packed: Dict[str, object] = {}
        self._packed_sig = None

    def _set_window_size_(self, window_size, shift_size=0):          # attention.py:255-263
        self.window_size = (window_size, window_size) if not isinstance(window_size, (tuple, list)) else tuple(window_size)
        self.shift_size = (shift_size, shift_size) if not isinstance(shift_size, (tuple, list)) else tuple(shift_size)
        attn = self._modules["attn"]
        if "relative_coord" in attn._buffers:
            attn._buffers["relative_coord"] = S.relative_coord_table(self.window_size[0]).to(attn._buffers["relative_coord"].device)

    def _prepare(self, device):
        if self._ops_obj is None or torch.device(self._ops_obj.device) != torch.device(device):
            if device.type != "cuda":
                raise RuntimeError("atm-vfi_amd blocks run on MI355X only: move the module and its input to 'cuda' (HIP)")
            self._ops_obj = HipOps(device)
            self._bufs.clear(); self._geo.clear(); self._packed_sig = None
        ops = self._ops_obj
        sd = dict(self.named_parameters())
        sig = tuple((p.data_ptr(), p._version) for p in sd.values())
        if sig != self._packed_sig:
            P = {"b." + k: v.detach() for k, v in sd.items()}
            if self.CROSS:
                qkv = torch.cat([sd["attn.q.weight"].detach(), sd["attn.kv.weight"].detach()], 0)
            else:
                qkv = sd["attn.qkv.weight"].detach()
            P["pk:b.attn.qkv.weight"] = ops.pack_weight(GEMM_LINEAR, qkv)
            for nm in ("attn.proj", "mlp.fc1", "mlp.fc2"):
                P[f"pk:b.{nm}.weight"] = ops.pack_weight(GEMM_LINEAR, sd[f"{nm}.weight"].detach())
            P["pk:b.mlp.dwconv.dwconv.weight"] = ops.pack_dw_weight(sd["mlp.dwconv.dwconv.weight"].detach())
            self._packed, self._packed_sig = P, sig
        return ops, self._packed

    def _run(self, x, frames, H, W):
        if x.dim() != 4 or tuple(x.shape[:3]) != (frames, H, W) or x.shape[3] != self.dim:
            raise ValueError(f"expected x of shape [{frames},{H},{W},{self.dim}], got {tuple(x.shape)}")
        if self.shift_size[0] != self.shift_size[1] or self.window_size[0] != self.window_size[1]:
            raise NotImplementedError("square windows and shifts only")
        with torch.no_grad(), torch.cuda.device(x.device):
            ops, P = self._prepare(x.device)
            xin = x.detach().contiguous().float().reshape(frames * H * W, self.dim)
            out = ops.empty(frames * H * W, self.dim)
            motion = ops.empty(2, (frames // 2) * H * W, 2) if self.CROSS else None          # [frame, B*hw, 2]
            self._block(ops, P, "b", xin, frames, H, W, self.window_size[0], self.shift_size[0], self.CROSS, out, motion, "blk")
        return out, motion


class ATMFormer(_Block):
    """attention.py:216-334.  forward(x [2B,H,W,C], H, W, B) -> (x [2B, H*W, C], motion [2B, H*W, 2])."""
    CROSS = True

    def forward(self, x, H, W, B):
        out, motion = self._run(x, 2 * B, H, W)
        return out.reshape(2 * B, H * W, self.dim), motion.reshape(2 * B, H * W, 2)


class RefineBottleneck(_Block):
    """attention.py:393-495.  forward(x [B,H,W,C]) -> x [B, H*W, C]."""
    CROSS = False

    def __init__(self, dim, window_size=8, **kw):
        super().__init__(dim, window_size=window_size, **kw)

    def forward(self, x):
        b, h, w, _ = x.shape
        out, _ = self._run(x, b, h, w)
        return out.reshape(b, h * w, self.dim)
