"""atm-vfi_amd: MI355X-native forward hot path of ATM-VFI.

``csrc/`` holds the hand-written HIP kernels and the C ABI (``include/atmvfi.h``),
``hip_ops.py`` the ctypes binding, ``network.py`` the reference-compatible ``Network``.
"""
from .schema import VARIANTS, param_schema, synthetic_state_dict, reference_init_state_dict  # noqa: F401
from .network import Network, NetworkBase, NetworkLite  # noqa: F401
