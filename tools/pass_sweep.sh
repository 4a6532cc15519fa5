#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export PT_TRACE_REFILL_MIN=${PT_TRACE_REFILL_MIN:-32,32,32,64} PT_TRACE_LEAF_QUORUM=${PT_TRACE_LEAF_QUORUM:-20,20,20,8}
for spb in 32; do
  echo "== spp_per_pass=$spb"
  python bench.py --spp 32 --steps 1 --warmup 1 --cpu-seconds 0 --spp-per-pass $spb 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], json.dumps(d['kernels_ms_per_step']))"
done
