#!/usr/bin/env python3
"""Hash of the sources the device library is built from (pbrt-rust_amd/csrc/*.{h,hip,cpp}, its Makefile, include/mi355pt.h).
tools/summarize_profile.py / tools/merge_pmc.py store it with every PMC record they write; bench.py compares it with the tree it runs from and
reports `traffic_code_match` -- a committed traffic figure that no longer belongs to the code is not turned into a roofline fraction."""
import glob
import hashlib
import os


def code_hash(root=None):
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "pbrt-rust_amd", "csrc", "*.h")) + glob.glob(os.path.join(root, "pbrt-rust_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(root, "pbrt-rust_amd", "csrc", "*.cpp")) + [os.path.join(root, "pbrt-rust_amd", "csrc", "Makefile"), os.path.join(root, "include", "mi355pt.h")])
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.relpath(f, root).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(code_hash())
