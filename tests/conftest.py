import importlib
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLD, "manifest.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def schema():
    return importlib.import_module("atm-vfi_amd.schema")


@pytest.fixture(scope="session")
def weights(schema):
    cache = {}

    def get(variant, motion_gain=1.0):
        """The stress weights (seed 1); ``motion_gain`` != 1: the large-motion set of the ``*_large`` fixtures."""
        key = (variant, float(motion_gain))
        if key not in cache:
            cache[key] = schema.synthetic_state_dict(variant, seed=1, motion_gain=float(motion_gain))
        return cache[key]
    return get


def pytest_sessionstart(session):
    # diagnostic: ATMVFI_ABORT_BT=1 chains a native-backtrace SIGABRT handler (tools/probes/abort_bt.c) behind pytest's faulthandler
    if os.environ.get("ATMVFI_ABORT_BT") == "1":
        import ctypes
        ctypes.CDLL(os.path.join(ROOT, "tools", "probes", "abort_bt.so"))
