#!/usr/bin/env python3
"""Merge the per-config PMC records a round's profile runs wrote (gpurun_out/<tag>/<config>/pmc_traffic.json, tools/profile_config.sh) and the
SQ issue counters (gpurun_out/<tag>_sq*/..., tools/sq_profile.sh output text) into the committed profiles/pmc_traffic.json.
usage: tools/merge_pmc.py <gpurun_out tag dir> [<config>=<sq summary text> ...]"""
import json, os, re, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
dst = os.path.join(R, "profiles", "pmc_traffic.json")
data = json.load(open(dst)) if os.path.exists(dst) else {}
import glob
for cfg in ("C2", "C3", "C4", "C5"):
    for f in sorted(glob.glob(os.path.join(tag, "**", "pmc_traffic.json"), recursive=True)):
        rec = json.load(open(f))
        if cfg in rec and rec[cfg]:
            data[cfg] = rec[cfg]
for spec in sys.argv[2:]:
    cfg, path = spec.split("=", 1)
    # lines of tools/sq_profile.sh: "<kernel> n=.. NAME=value ..."; VALU busy = 4 cycles x wave instructions / (SIMDs x active cycles of one XCD)
    vals = {}
    for line in open(path):
        m = re.match(r"(k_[a-z_0-9]+(?:<[^>]*>)?)\s+n=\s*\d+\s+(.*)", line.strip())
        if m:
            vals.setdefault(m.group(1), {}).update({k: float(v) for k, v in (kv.split("=") for kv in m.group(2).split())})
    for k, v in vals.items():
        if k in data.get(cfg, {}) and "SQ_INSTS_VALU" in v and "GRBM_GUI_ACTIVE" in v:
            data[cfg][k]["sq"] = {n: v[n] for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY") if n in v}
            data[cfg][k]["valu_busy_frac"] = round(4.0 * v["SQ_INSTS_VALU"] / (1024.0 * v["GRBM_GUI_ACTIVE"] / 8.0), 4)
            data[cfg][k]["valu_busy_def"] = "4 cycles x SQ_INSTS_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), tools/sq_profile.sh"
json.dump(data, open(dst, "w"), indent=1)
print("wrote", dst, {c: len(v) for c, v in data.items()})
