"""Importable alias for the ``atm-vfi_amd`` package directory (the hyphen in the directory
name keeps it out of reach of a plain ``import`` statement)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("atm-vfi_amd")
sys.modules[__name__] = _pkg
