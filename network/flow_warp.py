"""Drop-in for the reference's ``network/flow_warp.py`` (flow_warp.py:50-60): the same
``flow_warp(feature, flow, mask=False, padding_mode='zeros')`` signature backed by the HIP bilinear-gather kernels."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import atmvfi_amd as _pkg  # noqa: E402
from importlib import import_module

_hip = import_module("atm-vfi_amd.hip_ops")
_ops = {}


def flow_warp(feature, flow, mask=False, padding_mode="zeros"):
    """feature [B,C,H,W], flow [B,2,H,W] (CUDA/HIP fp32) -> backward-warped feature; with ``mask=True`` the pair
    (warped, in-range mask [B,H,W] bool) of bilinear_sample(return_mask=True) (flow_warp.py:26-47); ``padding_mode`` as
    ``F.grid_sample`` takes it ('zeros', 'border', 'reflection')."""
    assert flow.size(1) == 2                                    # flow_warp.py:55
    dev = feature.device
    if dev not in _ops:
        _ops[dev] = _hip.HipOps(dev)
    ops = _ops[dev]
    src = feature.contiguous().float()
    fl = flow.contiguous().float()
    out = torch.empty_like(src, memory_format=torch.contiguous_format)
    if not mask and padding_mode == "zeros":                    # the hot-path form
        ops.flow_warp(src, fl, out)
        return out
    b, _, h, w = src.shape
    m = torch.empty(b, h, w, dtype=torch.bool, device=dev) if mask else None
    ops.flow_warp_ex(src, fl, out, mask=m, padding_mode=padding_mode)
    return (out, m) if mask else out
