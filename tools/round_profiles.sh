#!/bin/bash
# The round's record (run through gpurun): GPU tests, the headline profile recipe on C2 (tools/profile_gpu.sh: bench, rocprofv3 kernel
# stats, PMC traffic passes), and one bench.py line per BASELINE config C3 / C4 / C5 at its named spp.
# usage: tools/round_profiles.sh <tag>
TAG=${1:-r2_final}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
echo "== GPU tests"; timeout -k 10 600 python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1; tail -n 2 $OUT/gpu_tests.log
echo "== C2 profile recipe"; tools/profile_gpu.sh $TAG 128 128 > $OUT/profile.log 2>&1; head -12 $OUT/summary.txt
for C in C3 C4 C5; do
  echo "== $C at its named spp"
  timeout -k 10 400 python bench.py --config $C --steps 1 --warmup 1 --cpu-seconds 8 > $OUT/bench_$C.json 2> $OUT/bench_$C.err || { echo "$C failed"; tail -5 $OUT/bench_$C.err; }
  python3 - $OUT/bench_$C.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print(d["config"]["name"], d["value"], "Msamples/s", d["ms_per_step"], "ms/step; cpu", d["cpu_baseline"]["value"] if d["cpu_baseline"] else None, "; roofline", d["roofline"]["kernel"], d["roofline"]["frac"])
except Exception as e:
    print("no result:", e)
PY
done
find $OUT -name '*kernel_trace.csv' -size +8M -delete
du -sh $OUT
