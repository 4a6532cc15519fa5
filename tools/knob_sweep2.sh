#!/bin/bash
# sweep leaf_quorum / refill_min for a variant library: tools/knob_sweep2.sh <variant.so> "<refill> <quorum>" ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
V=$1; shift
if [ "$V" != default ]; then export PT_LIB_PATH=$PWD/pbrt-rust_amd/csrc/variants/$V; fi
for cfg in "$@"; do
  set -- $cfg
  PT_TRACE_REFILL_MIN=$1 PT_TRACE_LEAF_QUORUM=$2 python bench.py --spp 64 --steps 1 --warmup 1 --cpu-seconds 0 --spp-per-pass 32 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print('$V refill=$1 quorum=$2', d['value'], {n:k[n]['ms'] for n in ('extend_camera','extend','extend_mis','shadow')})"
done
