"""Drop-in for the reference's ``network/flow_warp.py`` (flow_warp.py:50-60): the same
``flow_warp(feature, flow)`` signature backed by the HIP bilinear-gather kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import atmvfi_amd as _pkg  # noqa: E402
from importlib import import_module

_hip = import_module("atm-vfi_amd.hip_ops")
_ops = {}


def flow_warp(feature, flow, mask=False, padding_mode="zeros"):
    """feature [B,C,H,W], flow [B,2,H,W] (CUDA/HIP fp32) -> backward-warped feature."""
    if mask or padding_mode != "zeros":
        raise NotImplementedError("only the hot-path form flow_warp(feature, flow) is provided")
    dev = feature.device
    if dev not in _ops:
        _ops[dev] = _hip.HipOps(dev)
    out = torch.empty_like(feature, memory_format=torch.contiguous_format)
    _ops[dev].flow_warp(feature.contiguous().float(), flow.contiguous().float(), out)
    return out
