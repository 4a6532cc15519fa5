#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3_d; mkdir -p $O
timeout -k 10 800 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; tail -n 3 $O/gpu_tests.log
timeout -k 10 600 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err
python3 - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print(d['value'], d['ms_per_step'], d['config']['spp_per_pass'], {k:v['ms'] for k,v in d['kernels_ms_per_step'].items()})
print({c:(v.get('value'),v.get('ms_per_step')) for c,v in d.get('other_configs',{}).items()})
PY
timeout -k 10 600 python bench.py --config C3 --steps 1 --warmup 1 --cpu-seconds 0 > $O/bench_C3.json 2>> $O/bench.err
python3 - $O/bench_C3.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print(d['value'], d['ms_per_step'], d['config']['spp_per_pass'], {k:v['ms'] for k,v in d['kernels_ms_per_step'].items()})
PY
