#!/usr/bin/env python3
"""sha256 over everything that determines ``libatmvfi_hip.so``: the kernel sources, headers and generated includes under
``atm-vfi_amd/csrc`` (``*.hip``, ``*.h``, ``*.inc``), the Makefile (compiler flags) and the C-ABI header.

The Makefile bakes this value into the library (``atmvfi_source_digest()``); ``__graft_entry__.build()`` rebuilds when the library's
value differs from the sources', ``smoke()`` and a ``-m gpu`` test assert they agree, and ``bench.py`` quotes PMC traffic figures only
from passes stamped with the same value.  Prints the digest; ``digest()`` is the importable form."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_files():
    d = os.path.join(ROOT, "atm-vfi_amd", "csrc")
    files = [os.path.join(d, fn) for fn in sorted(os.listdir(d)) if fn.endswith((".hip", ".h", ".inc")) or fn == "Makefile"]
    files.append(os.path.join(ROOT, "include", "atmvfi.h"))
    return files


def digest() -> str:
    hsh = hashlib.sha256()
    for path in source_files():
        hsh.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            hsh.update(f.read())
    return hsh.hexdigest()


if __name__ == "__main__":
    print(digest())
