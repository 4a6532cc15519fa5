#!/bin/bash
# usage: tools/variant_waves.sh "<variant.so|default>:<waves_per_cu>" ...
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for vw in "$@"; do
  v=${vw%%:*}; w=${vw##*:}
  if [ "$v" = default ]; then unset PT_LIB_PATH; else export PT_LIB_PATH="$PWD/pbrt-rust_amd/csrc/variants/$v"; fi
  PT_TRACE_WAVES_PER_CU=$w python bench.py --steps 2 --warmup 1 --cpu-seconds 0.5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print('$vw', 'value', d['value'], {n: round(x['ms'],1) for n,x in k.items() if n in ('extend_camera','extend','extend_mis','shadow','shade_matte')}, 'frac', d['roofline']['frac'])"
done
