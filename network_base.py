"""Top-level shim so that ``sys.path.append('../'); from network_base import Network``
(benchmark/test_*.py:12-16 of the reference) resolves; see network/network_base.py."""
import atmvfi_amd as _pkg

Network = _pkg.NetworkBase
