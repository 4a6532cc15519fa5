#!/bin/bash
# SQ issue/stall counters of one headline-size pass (run through gpurun): tools/sq_profile.sh <tag> [spp]
TAG=${1:-sq}
PPASS=${2:-32}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq1 -- python3 $REPO/bench.py --config ${CONFIG:-C2} --steps 1 --warmup 0 --spp $PPASS --cpu-seconds 0 > $OUT/sq1_bench.json 2> $OUT/sq1.err
timeout -k 5 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/sq2 -- python3 $REPO/bench.py --config ${CONFIG:-C2} --steps 1 --warmup 0 --spp $PPASS --cpu-seconds 0 > $OUT/sq2_bench.json 2> $OUT/sq2.err
timeout -k 5 200 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_INSTS_VALU_FMA_F64 --kernel-trace --output-format csv -d $OUT/sq3 -- python3 $REPO/bench.py --config ${CONFIG:-C2} --steps 1 --warmup 0 --spp $PPASS --cpu-seconds 0 > $OUT/sq3_bench.json 2> $OUT/sq3.err
python3 - $OUT <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
out = sys.argv[1]
def short(name):
    m = re.search(r"(k_[a-z_0-9]+(<[^>]*>)?)", name)
    return m.group(1) if m else name.split("(")[0][:30]
for tag in ("sq1", "sq2", "sq3"):
    acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int); first = None
    for f in glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            k = short(r["Kernel_Name"]); c = r["Counter_Name"]; first = first or c
            acc[k][c] += float(r["Counter_Value"])
            if c == first: n[k] += 1
    print("==", tag)
    for k in sorted(acc, key=lambda k: -sum(acc[k].values()))[:6]:
        print(f"{k:28s} n={n[k]:3d} " + " ".join(f"{c}={v:.4g}" for c, v in sorted(acc[k].items())))
    tail = open(os.path.join(out, tag + ".err")).read()[-600:] if not acc else ""
    if tail: print(tail)
PY
find $OUT -name '*kernel_trace.csv' -size +4M -delete
