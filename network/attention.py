"""Placeholder for the reference's ``network/attention.py`` import path.

The reference's ``ATMFormer`` / ``RefineBottleneck`` modules are not separate objects here:
their arithmetic is the ``_block`` method of ``atm-vfi_amd/network.py`` (LayerNorm-gather,
fused qkv GEMM, ``atmvfi_window_attention``, projection-scatter GEMM, MLP), driven by the
parameters stored under the reference's names inside ``Network``."""
