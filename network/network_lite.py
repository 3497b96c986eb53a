"""Drop-in for the reference's ``network/network_lite.py``: ``from network_lite import Network``
(demo_2x.py:11-12, benchmark/*.py) resolves to the MI355X-native implementation."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import atmvfi_amd as _pkg  # noqa: E402

Network = _pkg.NetworkLite
