#!/bin/bash
# sweep the persistent-traversal knobs: "refill_min(ext,mis,shadow,camera) leaf_quorum(ext,mis,shadow,camera)" per line
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "24,24,24,32 8,8,8,20" "32,32,32,32 8,8,8,20" "16,16,16,32 8,8,8,20" "24,24,24,32 12,12,12,20" "24,24,24,32 4,4,4,20" "24,24,24,40 8,8,8,28" "24,24,24,24 8,8,8,12" "32,32,32,32 12,12,12,20" "40,40,40,48 16,16,16,24"; do
  set -- $cfg
  PT_TRACE_REFILL_MIN=$1 PT_TRACE_LEAF_QUORUM=$2 python bench.py --spp 64 --steps 1 --warmup 1 --cpu-seconds 0 --spp-per-pass 32 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print('refill=$1 quorum=$2', d['value'], {n:k[n]['ms'] for n in ('extend_camera','extend','extend_mis','shadow')})"
done
