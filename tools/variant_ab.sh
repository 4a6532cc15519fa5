#!/bin/bash
# usage: tools/variant_ab.sh <bench args...> -- <variant.so|default> ...   : one bench line per library variant (value + per-kind ms)
ARGS=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do ARGS+=("$1"); shift; done; shift
for v in "$@"; do
  if [ "$v" = default ]; then unset PT_LIB_PATH; else export PT_LIB_PATH=pbrt-rust_amd/csrc/variants/$v; fi
  python bench.py "${ARGS[@]}" --cpu-seconds 0 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print('%-14s %9.2f Msamples/s  algo_over_peak %.3f  ' % ('$v', d['value'], d['roofline']['algorithmic_over_hbm_peak']) + ' '.join('%s=%.1f' % (n, x['ms']) for n, x in k.items() if x['ms'] >= 1.0))"
done
