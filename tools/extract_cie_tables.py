#!/usr/bin/env python3
"""Regenerates include/pt_cie_tables.h -- the CIE 1931 standard-observer DATA tables (CIE_X / CIE_Y / CIE_Z / CIE_LAMBDA, 471 samples,
360..830 nm) and CIE_Y_INTEGRAL -- from the reference's core/cie.rs. Data only (like the Sobol' matrices and the Perlin permutation)."""
import re, sys
src = open(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src/core/cie.rs").read()
def table(name):
    a = src.index("pub const %s: [Float; N_CIE_SAMPLES] = [" % name); b = src.index("];", a)
    body = re.sub(r"//.*", "", src[a:b][src[a:b].index("= [") + 3:])
    vals = re.findall(r"[-+]?(?:\d+\.?\d*(?:[eE][-+]?\d+)?|\.\d+)", body)
    assert len(vals) == 471, (name, len(vals))
    return vals
yint = re.search(r"CIE_Y_INTEGRAL: Float = ([0-9.eE+-]+);", src).group(1)
out = ["/* CIE 1931 2-degree standard observer, 360..830 nm in 1 nm steps: DATA from core/cie.rs (CIE_X, CIE_Y, CIE_Z, CIE_LAMBDA,",
       " * CIE_Y_INTEGRAL), extracted by tools/extract_cie_tables.py. Used by the .pbrt front end to convert \"xyz\" / \"blackbody\" /",
       " * \"spectrum\" parameters to RGB exactly as core/spectrum.rs does. */", "#pragma once", "#define PT_N_CIE_SAMPLES 471", "#define PT_CIE_Y_INTEGRAL %sf" % yint]
for name in ("CIE_X", "CIE_Y", "CIE_Z", "CIE_LAMBDA"):
    vals = table(name)
    out.append("#define PT_%s_VALUES \\" % name)
    rows = ["    " + ", ".join((v if ("." in v or "e" in v.lower()) else v + ".0") + "f" for v in vals[i:i + 8]) + "," for i in range(0, 471, 8)]
    rows[-1] = rows[-1].rstrip(",")
    out.append(" \\\n".join(rows))
print("\n".join(out))
