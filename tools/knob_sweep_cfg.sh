#!/bin/bash
# Sweep the traversal scheduling knobs on one config: tools/knob_sweep_cfg.sh <tag> <config> <spp> "ENV=VAL ENV=VAL" ...
TAG=$1; CFG=$2; SPP=$3; shift 3
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; cd $REPO
for envs in "$@"; do
  name=$(echo "$envs" | tr ' =,' '___')
  env $envs timeout -k 10 120 python bench.py --config $CFG --spp $SPP --steps 1 --warmup 1 --cpu-seconds 0 > $OUT/$name.json 2> $OUT/$name.err
  rc=$?
  python3 - "$envs" $OUT/$name.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
    k = d["kernels_ms_per_step"]
    print(f"{sys.argv[1]:60s} {d['value']:9.2f} Msamples/s  " + " ".join(f"{n}={k[n]['ms']:.1f}" for n in ("extend_camera", "trace", "extend", "extend_mis", "shadow", "extend_probe") if n in k))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping"; exit 1; fi
done
