#!/bin/bash
# usage: tools/waves_ab.sh "<variant|default> <waves_per_cu>" ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "$@"; do
  set -- $cfg
  if [ "$1" = default ]; then unset PT_LIB_PATH; else export PT_LIB_PATH=$PWD/pbrt-rust_amd/csrc/variants/$1; fi
  PT_TRACE_WAVES_PER_CU=$2 python bench.py --spp 64 --steps 1 --warmup 1 --cpu-seconds 0 --spp-per-pass 32 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print('$1 waves=$2', d['value'], {n:k[n]['ms'] for n in ('extend_camera','extend','extend_mis','shadow')})"
done
