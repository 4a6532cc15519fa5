#!/usr/bin/env python3
"""Extract the Sobol' generator-matrix DATA tables from the reference into a binary fixture.

Source of the numbers (data, not code): /root/reference/src/core/sobolmatrices.rs
  SOBOL_MATRICES_32       :5-8881      1024 dims x 52 u32 (Joe-Kuo direction numbers as shipped with pbrt-v3)
  VD_C_SOBOL_MATRICES     :26636-26845 M1..M25  (50,48,...,2 u64)
  VD_C_SOBOL_MATRICES_INV :26846-27537 MI1..MI26 (2,4,...,52 u64)

Output layout (little endian), pbrt-rust_amd/data/sobol_tables.bin:
  char[8]  magic "PTSOBOL1"
  u32      sobol32[1024*52]
  u64      vdc[25][52]      zero padded rows
  u64      vdc_inv[26][52]  zero padded rows
Run in the build container only (the reference does not travel to the GPU box).
"""
import re, struct, sys, pathlib
import numpy as np

SRC = pathlib.Path("/root/reference/src/core/sobolmatrices.rs")
OUT = pathlib.Path(__file__).resolve().parent.parent / "pbrt-rust_amd" / "data" / "sobol_tables.bin"

def ints(body):
    vals = []
    for tok in re.findall(r"0x[0-9a-fA-F_]+|\b\d[\d_]*\b", body):
        tok = tok.replace("_", "")
        if tok.endswith("u64"): tok = tok[:-3]
        vals.append(int(tok, 16) if tok.startswith("0x") else int(tok))
    return vals

def main():
    text = SRC.read_text()
    m = re.search(r"SOBOL_MATRICES_32[^=]*=\s*\[(.*?)\];", text, re.S)
    s32 = ints(m.group(1))
    assert len(s32) == 1024 * 52, len(s32)
    def grab(prefix, n):
        rows = []
        for i in range(1, n + 1):
            mm = re.search(r"const %s%d: \[u64; (\d+)\] = \[(.*?)\];" % (prefix, i), text, re.S)
            body = re.sub(r"_u64", "", mm.group(2))
            v = ints(body)
            assert len(v) == int(mm.group(1)), (prefix, i, len(v))
            rows.append(v)
        return rows
    vdc = grab("M", 25)
    inv = grab("MI", 26)
    a = np.zeros((25, 52), dtype=np.uint64); b = np.zeros((26, 52), dtype=np.uint64)
    for i, r in enumerate(vdc): a[i, :len(r)] = np.array(r, dtype=np.uint64)
    for i, r in enumerate(inv): b[i, :len(r)] = np.array(r, dtype=np.uint64)
    OUT.parent.mkdir(parents=True, exist_ok=True)
    with open(OUT, "wb") as f:
        f.write(b"PTSOBOL1")
        f.write(np.array(s32, dtype="<u4").tobytes())
        f.write(a.astype("<u8").tobytes())
        f.write(b.astype("<u8").tobytes())
    print("wrote", OUT, OUT.stat().st_size, "bytes")

if __name__ == "__main__":
    main()
