"""Is -fno-slp-vectorize on attention.hip (csrc/Makefile) still needed?  Runs the window-attention GPU tests against
``libatmvfi_hip_slp.so`` (`make -C atm-vfi_amd/csrc slp`: the product sources with the SLP vectorizer left on in every file)
in this process, by pointing the binding's default library path at it before the tests import it.

    python tools/slp_check.py [library file name in atm-vfi_amd/] [pytest -k expression]

``libatmvfi_hip_oldattn_slp.so`` is the round-2 reproducer: attention.hip of commit 40b1371 (the commit that introduced the flag; its
motion expectation was a scalar fp32 chain per (query, key)) compiled WITH the SLP vectorizer and linked with today's other objects:
    git show 40b1371:atm-vfi_amd/csrc/attention.hip > /tmp/old/attention.hip   (+ its common.h, gemm_common.h, conv3_common.h, atmvfi.h)
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I/tmp/old/inc -c /tmp/old/attention.hip -o /tmp/old/attention_old_slp.o
    hipcc -shared -fPIC --offload-arch=gfx950 <today's *.o except attention.o> /tmp/old/attention_old_slp.o -o tools/lib/libatmvfi_hip_oldattn_slp.so
"""
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
H = importlib.import_module("atm-vfi_amd.hip_ops")
args = sys.argv[1:]
name = args.pop(0) if args and args[0].endswith(".so") else "libatmvfi_hip_slp.so"
lib = os.path.join(ROOT, "atm-vfi_amd" if name == "libatmvfi_hip.so" else os.path.join("tools", "lib"), name)
assert os.path.exists(lib), "build it first: make -C atm-vfi_amd/csrc slp"
H.LIB_PATH = lib
H.load_library.__defaults__ = (lib,)
print("library under test:", lib, flush=True)
k = args[0] if args else "window_attention or atm_block or atmformer"
sys.exit(pytest.main([os.path.join(ROOT, "tests", "test_gpu_ops.py"), "-q", "-m", "gpu", "-k", k, "-p", "no:cacheprovider"]))
