#!/bin/bash
# round 3 record: GPU tests, the driver's bench command, sim-world lines, the other configs at their named spp, rocprofv3 stats of the bench command
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r3_final; mkdir -p $O
echo "== GPU tests"; timeout -k 10 900 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; tail -n 2 $O/gpu_tests.log
echo "== bench (driver form)"; timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python3 - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
r=d['roofline']
print(d['value'], d['ms_per_step'], d['config']['spp_per_pass'], 'frac', r['frac'], 'algo/peak', r['algorithmic_over_hbm_peak'], 'gather', r['frac_of_gather_ceiling'], 'avg launch ms', r['avg_launch_ms'], 'cpu', d['cpu_baseline']['value'])
print({k:v['ms'] for k,v in d['kernels_ms_per_step'].items()})
print({c:(v.get('value'),v.get('ms_per_step'),v.get('hbm_frac')) for c,v in d.get('other_configs',{}).items()})
PY
for sw in 2 4 8; do
  timeout -k 10 300 python3 bench.py --sim-world $sw --steps 3 --warmup 1 --cpu-seconds 0 --other-configs off > $O/sim_$sw.json 2>> $O/bench.err
  python3 -c "
import json;d=json.loads(open('$O/sim_$sw.json').read().strip().split('\n')[-1]);print('sim-world $sw', d['ms_per_step'], d['value'])"
done
for C in C3 C4 C5; do
  timeout -k 10 500 python3 bench.py --config $C --steps 1 --warmup 1 --cpu-seconds 8 > $O/bench_$C.json 2>> $O/bench.err
  python3 -c "
import json;d=json.loads(open('$O/bench_$C.json').read().strip().split('\n')[-1]);r=d['roofline'];print('$C', d['value'], d['ms_per_step'], d['config']['spp_per_pass'], r['kernel'], 'frac', r['frac'], 'algo/peak', r['algorithmic_over_hbm_peak'], 'cpu', d['cpu_baseline']['value'] if d['cpu_baseline'] else None)"
done
cd /tmp && export TMPDIR=/tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/stats -- python3 $REPO/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --other-configs off > $REPO/$O/stats_bench.json 2> $REPO/$O/stats.err
cd $REPO; find $O -name '*kernel_trace.csv' -size +4M -delete
python3 - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r3_final/stats/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]: print(r['Name'][:40], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3)
PY
