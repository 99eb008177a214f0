#!/usr/bin/env python3
"""sha256 (first 16 hex digits) over the sources libqbhip.so is built from: what a measured profile is stamped with and what
bench.py compares against before it quotes that profile's traffic (there is no .git on the GPU box)."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_sources_sha16():
    h = hashlib.sha256()
    # the translation units and their private header; the public header is mostly prose (a layout change there shows up in qbh_api.cpp)
    files = sorted(glob.glob(os.path.join(ROOT, "quantum_basis_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "quantum_basis_amd", "csrc", "*.cpp")) +
                   glob.glob(os.path.join(ROOT, "quantum_basis_amd", "csrc", "*.hpp")))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_sources_sha16())
