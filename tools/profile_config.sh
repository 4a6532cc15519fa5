#!/bin/bash
# One BASELINE config's profile record (run through gpurun): the bench line, rocprofv3 kernel stats and the PMC traffic passes over ONE
# pass of the size the library picks for the config. usage: tools/profile_config.sh <tag> <C2|C3|C4|C5>
TAG=$1; CFG=${2:-C2}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG/$CFG
mkdir -p $OUT
cd $REPO
timeout -k 10 400 python3 bench.py --config $CFG --steps 1 --warmup 1 --cpu-seconds 0 --other-configs off > $OUT/probe.json 2> $OUT/probe.err
PPASS=$(python3 -c "import json;print(json.loads(open('$OUT/probe.json').read().strip().split('\n')[-1])['config']['spp_per_pass'])")
echo "$CFG: spp per pass = $PPASS"
CONFIG=$CFG tools/profile_gpu.sh $TAG/$CFG $PPASS $PPASS > $OUT/profile.log 2>&1
head -14 $OUT/summary.txt
