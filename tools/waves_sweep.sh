#!/bin/bash
# sweep the number of resident persistent trace waves per CU (headline config, 2 steps each)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for w in "$@"; do
  PT_TRACE_WAVES_PER_CU=$w python bench.py --steps 2 --warmup 1 --cpu-seconds 0.5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print('waves/CU', $w, 'value', d['value'], {n: round(x['ms'],1) for n,x in k.items() if n in ('extend_camera','extend','extend_mis','shadow')}, 'frac', d['roofline']['frac'])"
done
