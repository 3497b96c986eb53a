"""Top-level shim (the reference's network_base.py:7 does ``from flow_warp import flow_warp``)."""
from network.flow_warp import flow_warp  # noqa: F401
