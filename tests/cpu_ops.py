"""TEST DOUBLE for ``atm-vfi_amd/hip_ops.HipOps`` -- CPU only, test infrastructure only.

Implements each op of the C ABI's vocabulary with plain torch on CPU tensors, following
the *op contracts* in ``include/atmvfi.h`` (views, row maps, group strides, epilogue
order).  It exists so that ``-m "not gpu"`` tests can exercise the HOST logic of
``Network.forward`` (buffer slicing, concat-free layouts, window maps, ordering quirks)
against the oracle without a GPU.  It is never importable from the product package and
is never selected automatically: the product raises without the HIP library.
"""
from __future__ import annotations

import importlib
from typing import Optional

import torch
import torch.nn.functional as F

hip_ops = importlib.import_module("atm-vfi_amd.hip_ops")
PackedWeight = hip_ops.PackedWeight
GEMM_CONV, GEMM_LINEAR, GEMM_DECONV = 0, 1, 2


class CpuOps:
    def __init__(self):
        self.device = torch.device("cpu")
        self.profile = None
        self.calls = []

    def empty(self, *shape):
        # poison so that any element the host forgets to produce shows up as NaN
        return torch.full(shape, float("nan"), dtype=torch.float32)

    def to_device_int(self, t):
        return t.to(torch.int32).contiguous()

    def pack_weight(self, mode, w):
        w = w.detach()
        if mode == GEMM_DECONV:
            cin, cout, kh, kw = w.shape
        elif mode == GEMM_LINEAR:
            cout, cin = w.shape[:2]
            kh = kw = 1
        else:
            cout, cin, kh, kw = w.shape
        return PackedWeight(mode, cout, cin, kh, kw, w, None)

    def pack_dw_weight(self, w):
        return w.detach()

    def pad_channels(self, v, mult=32):
        c = v.shape[0]
        out = torch.zeros((c + mult - 1) // mult * mult)
        out[:c] = v.detach()
        return out

    # ---- GEMMs ----
    def conv(self, x, w, out, stride=1, pad=1, dil=1, bias=None, prelu=None, in_prelu=None):
        self.calls.append("conv2d")
        xi = x.permute(0, 3, 1, 2)
        y = F.conv2d(xi, w.orig, bias, stride=stride, padding=pad, dilation=dil)
        if prelu is not None:
            y = F.prelu(y, prelu)
        assert y.shape[1:] == out.permute(0, 3, 1, 2).shape[1:], (y.shape, out.shape)
        out.copy_(y.permute(0, 2, 3, 1))

    def deconv(self, x, w, out, bias=None, prelu=None, in_prelu=None):
        self.calls.append("deconv2x2")
        xi = x.permute(0, 3, 1, 2)
        if in_prelu is not None:
            xi = F.prelu(xi, in_prelu[:xi.shape[1]])
        y = F.conv_transpose2d(xi, w.orig, bias, stride=2)
        if prelu is not None:
            y = F.prelu(y, prelu)
        out.copy_(y.permute(0, 2, 3, 1))

    def linear(self, x, w, out, bias=None, residual=None, out_row_map=None):
        self.calls.append("linear")
        xm = x.reshape(-1, x.shape[-1])
        y = F.linear(xm, w.orig.reshape(w.cout, w.cin), bias)
        if residual is not None:
            y = y + residual
        if out_row_map is None:
            if out.dim() == 3:
                g, r, c = out.shape
                out.copy_(y.reshape(g, r, c))
            else:
                out.copy_(y)
        else:
            keep = out_row_map >= 0
            idx = out_row_map[keep].long()
            if out.dim() == 3:
                g, r, c = out.shape
                for gi in range(g):
                    sel = (idx // r) == gi
                    out[gi][idx[sel] % r] = y[keep][sel]
            else:
                out[idx] = y[keep]

    # ---- transformer ----
    def layernorm(self, x, out, gamma, beta, src_row_map=None):
        self.calls.append("layernorm")
        xm = x.reshape(-1, x.shape[-1])
        c = xm.shape[-1]
        if src_row_map is None:
            out.copy_(F.layer_norm(xm, (c,), gamma, beta, 1e-5))
        else:
            idx = src_row_map.long()
            src = torch.zeros(idx.numel(), c)
            ok = idx >= 0
            src[ok] = xm[idx[ok]]
            out.copy_(F.layer_norm(src, (c,), gamma, beta, 1e-5))

    def dwconv_gelu(self, x, out, w9, bias):
        self.calls.append("dwconv3x3_gelu")
        xi = x.permute(0, 3, 1, 2)
        y = F.gelu(F.conv2d(xi, w9, bias, padding=1, groups=xi.shape[1]))
        out.copy_(y.permute(0, 2, 3, 1))

    def window_attention(self, qkv, out, motion, labels, bw, nw, ws, heads, hd, kv_shift):
        self.calls.append("window_attention")
        n = ws * ws
        c = heads * hd
        t = qkv.reshape(bw, n, 3, heads, hd)
        q = t[:, :, 0].permute(0, 2, 1, 3)
        src = (torch.arange(bw) + kv_shift) % bw
        k = t[src, :, 1].permute(0, 2, 1, 3)
        v = t[src, :, 2].permute(0, 2, 1, 3)
        attn = (q @ k.transpose(-2, -1)) * (hd ** -0.5)
        if labels is not None:
            mask = (labels[:, :, None] != labels[:, None, :]).float() * -100.0
            attn = (attn.reshape(bw // nw, nw, heads, n, n) + mask[None, :, None]).reshape(bw, heads, n, n)
        attn = attn.softmax(-1)
        out.copy_((attn @ v).transpose(1, 2).reshape(bw * n, c))
        if motion is not None:
            idx = torch.arange(n)
            cx, cy = (idx % ws).float(), (idx // ws).float()
            rel = torch.stack([cx[None, :] - cx[:, None], cy[None, :] - cy[:, None]])
            m = (attn[:, :, None] * rel[None, None]).sum(-1)          # [Bw,heads,2,N]
            motion.copy_(m.permute(0, 3, 1, 2).reshape(bw * n, heads, 2))

    def motion_head(self, motion, row_map, w0, b0, w1, b1, out):
        self.calls.append("motion_head")
        m = motion.permute(0, 2, 1)                                     # [rows,2,heads]
        y = F.linear(F.gelu(F.linear(m, w0, b0)), w1, b1)[..., 0]      # [rows,2]
        keep = row_map >= 0
        idx = row_map[keep].long()
        g, r, _ = out.shape
        for gi in range(g):
            sel = (idx // r) == gi
            out[gi][idx[sel] % r] = y[keep][sel]

    # ---- warps ----
    @staticmethod
    def _warp(src, flow):
        b, c, h, w = src.shape
        ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
        gx = 2 * (xs[None] + flow[:, 0]) / (w - 1) - 1
        gy = 2 * (ys[None] + flow[:, 1]) / (h - 1) - 1
        return F.grid_sample(src, torch.stack([gx, gy], -1), mode="bilinear", padding_mode="zeros", align_corners=True)

    def flow_warp(self, src, flow, dst):
        self.calls.append("flow_warp")
        dst.copy_(self._warp(src, flow))

    def flow_warp_nhwc(self, src, flow, dst):
        self.calls.append("flow_warp_nhwc")
        dst.copy_(self._warp(src.permute(0, 3, 1, 2), flow).permute(0, 2, 3, 1))

    def warp_blend(self, im0, im1, motion, i0w, i1w, it, flow0=None, flow1=None, mask1=None, mask2=None,
                   orig0=None, orig1=None, pack15=None):
        self.calls.append("warp_blend")
        m = motion.permute(0, 3, 1, 2)
        a = self._warp(im0, m[:, 0:2])
        c = self._warp(im1, m[:, 2:4])
        m1 = torch.sigmoid(m[:, 4:5])
        m2 = 1 - m1
        t = m1 * a + m2 * c
        i0w.copy_(a); i1w.copy_(c); it.copy_(t)
        if flow0 is not None:
            flow0.copy_(m[:, 0:2]); flow1.copy_(m[:, 2:4])
        if mask1 is not None:
            mask1.copy_(m1); mask2.copy_(m2)
        if pack15 is not None:
            pack15.copy_(torch.cat([orig0, a, orig1, c, t], 1).permute(0, 2, 3, 1))

    def resize(self, src, dst, value_scale=1.0):
        self.calls.append("resize_bilinear_ac")
        dst.copy_(F.interpolate(src, size=dst.shape[-2:], mode="bilinear", align_corners=True) * value_scale)

    def flow_warp_up2(self, src, flow, dst, flow_up):
        self.flow_warp(src, flow, dst)
        self.resize(flow, flow_up, 2.0)

    def image_pyramid(self, im0, im1, l1, l2, l3):
        self.calls.append("image_pyramid")
        b = im0.shape[0]
        prev = torch.cat([im0, im1], 0)
        for dst in (l1, l2, l3):
            dst.copy_(F.interpolate(prev, size=dst.shape[-2:], mode="bilinear", align_corners=True))
            prev = dst

    def pack_frames(self, im0, im1, dst):
        self.calls.append("pack_frames")
        x = torch.cat([im0, im1], 0).permute(0, 2, 3, 1)
        dst[..., :3] = x
        dst[..., 3] = 0

    def final_residual(self, it, r, it_sum, it_clamped):
        self.calls.append("final_residual")
        s = it + (2 * torch.sigmoid(r.permute(0, 3, 1, 2)) - 1)
        it_sum.copy_(s)
        it_clamped.copy_(s.clamp(0, 1))

    def l1_mean(self, a, b, out, workspace=None):
        self.calls.append("l1_mean")
        out.copy_((a - b).abs().mean(dim=[1, 2, 3]))
