"""``from benchmark.utils import InputPadder`` (demo_2x.py:7 of the reference) resolves here.
Only the two helpers on the hot path's boundary are provided."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from importlib import import_module

_io = import_module("atm-vfi_amd.host_io")
InputPadder = _io.InputPadder
img2tensor = _io.img2tensor
