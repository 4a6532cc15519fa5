#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "24,24,24,32 8,8,8,16" "24,24,24,32 4,4,4,16" "16,16,16,32 8,8,8,16" "16,16,16,32 4,4,4,16" "24,24,24,32 1,1,1,16" "20,20,20,32 6,6,6,24" "8,8,8,32 4,4,4,16"; do
  set -- $cfg
  echo "== refill_min=$1 leaf_quorum=$2"
  PT_TRACE_REFILL_MIN=$1 PT_TRACE_LEAF_QUORUM=$2 python bench.py --spp 32 --steps 1 --warmup 1 --cpu-seconds 0 --spp-per-pass 32 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print(d['value'], {n:(k[n]['ms'],k[n].get('Mrays_s')) for n in ('extend_camera','extend','extend_mis','shadow')})"
done
