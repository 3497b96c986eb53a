"""Host boundary either side of the hot path: the reference's
``InputPadder`` (benchmark/utils.py:57-80), ``inference_2frame`` (demo_2x.py:54-87) and
``load_model_checkpoint`` (demo_2x.py:24-51), restated so that the reference's scripts
run unchanged on top of this package."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


class InputPadder:
    """Centre replicate-padding to a multiple of ``divisor`` (benchmark/utils.py:57-80)."""

    def __init__(self, dims, divisor: int = 16):
        self.ht, self.wd = dims[-2:]
        ph = (((self.ht // divisor) + 1) * divisor - self.ht) % divisor
        pw = (((self.wd // divisor) + 1) * divisor - self.wd) % divisor
        self._pad = [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2]

    def pad(self, *inputs):
        out = [F.pad(x, self._pad, mode="replicate") for x in inputs]
        return out[0] if len(out) == 1 else out

    def unpad(self, *inputs):
        out = [self._unpad(x) for x in inputs]
        return out[0] if len(out) == 1 else out

    def _unpad(self, x):
        ht, wd = x.shape[-2:]
        l, r, t, b = self._pad
        return x[..., t:ht - b, l:wd - r]


def img2tensor(img):        # benchmark/utils.py:83-86
    if img.shape[-1] > 3:
        img = img[:, :, :3]
    return torch.tensor(img).permute(2, 0, 1).unsqueeze(0) / 255.0


def inference_2frame(img0, img1, model, isBGR: bool = True, divisor: int = 64):
    """uint8 [H,W,3] frames -> uint8 [H,W,3] interpolated frame (demo_2x.py:54-87)."""
    dev = next(model.parameters()).device
    if isBGR:
        img0 = img0[:, :, ::-1].copy()
        img1 = img1[:, :, ::-1].copy()
    t0 = (torch.tensor(img0.transpose(2, 0, 1)).to(dev) / 255.).unsqueeze(0)
    t1 = (torch.tensor(img1.transpose(2, 0, 1)).to(dev) / 255.).unsqueeze(0)
    padder = InputPadder(t0.shape, divisor=divisor)
    t0, t1 = padder.pad(t0, t1)
    pred = model.forward(t0, t1)["I_t"][0]
    pred = padder.unpad(pred).detach().cpu().numpy().transpose(1, 2, 0)
    pred = np.round(pred * 255).astype(np.uint8)
    if isBGR:
        pred = pred[:, :, ::-1].copy()
    return pred


def strip_lazy_buffers(state):
    """Saved checkpoints carry the reference's lazily registered ``attn_mask``/``HW`` buffers
    (attention.py:304-305); every loader of the reference drops them (demo_2x.py:38-46)."""
    return {k: v for k, v in state.items() if "attn_mask" not in k and "HW" not in k}


def load_model_checkpoint(model, checkpoint_path, strict: bool = True, map_location=None):
    """demo_2x.py:24-51 -- accepts the trainer's 5-key dict or a bare state dict."""
    ck = torch.load(checkpoint_path, map_location=map_location or "cpu")
    optim = None
    if isinstance(ck, dict) and "model_state_dict" in ck:
        param = ck["model_state_dict"]
        optim = ck.get("optimizer_state_dict")
    else:
        param = ck
    model.load_state_dict(strip_lazy_buffers(param), strict=strict)
    return optim
