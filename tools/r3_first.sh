#!/bin/bash
# round 3, first GPU call: suite + default bench line + strong-scaling studies on one device
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r3_first
mkdir -p $OUT
cd $REPO
echo "== GPU tests"; timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/gpu_tests.log 2>&1; tail -n 3 $OUT/gpu_tests.log
echo "== default bench"; timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench.json; tail -3 $OUT/bench.err
echo "== gpus 2 on a 1-gpu box"; python bench.py --gpus 2 --cpu-seconds 0 2>&1 | tail -2
for spec in "1:0" "8:0" "8:0,0" "8:0,0,0" "1:0,0" "4:0" "4:0,0"; do
  sw=${spec%%:*}; devs=${spec##*:}
  echo "== sim-world $sw devices $devs"
  timeout -k 10 300 python bench.py --sim-world $sw --devices $devs --steps 3 --warmup 1 --cpu-seconds 0 --other-configs off > $OUT/sim_${sw}_${devs//,/_}.json 2> $OUT/sim.err || tail -3 $OUT/sim.err
  python3 -c "
import json,sys
d=json.loads(open('$OUT/sim_${sw}_${devs//,/_}.json').read().strip().split('\n')[-1])
print(d['ms_per_step'], d['value'], d.get('multi_gpu'))"
done
