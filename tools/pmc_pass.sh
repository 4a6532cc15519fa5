#!/bin/bash
# Every rocprofv3 run is wrapped in `timeout`: a counter set the hardware cannot schedule aborts and then hangs in finalize.
# Generic PMC passes over one headline-size pass: tools/pmc_pass.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...]
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for CTRS in "$@"; do
  i=$((i+1))
  timeout -k 5 ${PMC_TIMEOUT:-150} rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/p$i -- python3 $REPO/bench.py --steps 1 --warmup 0 --spp ${PMC_SPP:-32} --cpu-seconds 0 ${PMC_ARGS:-} > $OUT/p${i}_bench.json 2> $OUT/p$i.err
done
python3 - $OUT $i <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
out, n = sys.argv[1], int(sys.argv[2])
def short(name):
    m = re.search(r"(k_[a-z_0-9]+(<[^>]*>)?)", name)
    return m.group(1) if m else name.split("(")[0][:30]
for i in range(1, n + 1):
    acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int); first = None
    for f in glob.glob(os.path.join(out, f"p{i}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            k = short(r["Kernel_Name"]); c = r["Counter_Name"]; first = first or c
            acc[k][c] += float(r["Counter_Value"])
            if c == first: cnt[k] += 1
    print("== pass", i)
    for k in sorted(acc, key=lambda k: -sum(acc[k].values()))[:5]:
        print(f"{k:24s} n={cnt[k]:3d} " + " ".join(f"{c}={v:.4g}" for c, v in sorted(acc[k].items())))
    if not acc: print(open(os.path.join(out, f"p{i}.err")).read()[-800:])
PY
find $OUT -name '*kernel_trace.csv' -size +2M -delete; find $OUT -name '*counter_collection.csv' -size +8M -delete
