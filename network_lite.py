"""Top-level shim so that ``sys.path.append('../'); from network_lite import Network``
(benchmark/test_*.py:12-16 of the reference) resolves; see network/network_lite.py."""
import atmvfi_amd as _pkg

Network = _pkg.NetworkLite
