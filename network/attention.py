"""Drop-in for the reference's ``network/attention.py`` import path (``from network.attention import ATMFormer`` and
``from network.attention import RefineBottleneck as SwinTransformer``, network_base.py:8-9): the two transformer blocks as
stand-alone modules backed by the HIP kernels (``atm-vfi_amd/blocks.py``)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from importlib import import_module

_blocks = import_module("atm-vfi_amd.blocks")
ATMFormer = _blocks.ATMFormer
RefineBottleneck = _blocks.RefineBottleneck
