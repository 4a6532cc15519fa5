#!/bin/bash
# round 6 profile record (run through gpurun, one part per call): tools/r6_profiles.sh <part>
#   pmc <C2|C3|C4|C5>   rocprofv3 kernel stats + PMC traffic passes over one pass of the config (tools/profile_config.sh) + SQ issue counters (tools/sq_profile.sh)
#   util                PT_TRACE_UTIL build: lane occupancy / wave cycles by part, instance entries and stack spills for C2 and C4 (pbrt-rust_amd/csrc/variants/qutil: tools/build_variants.sh)
#   parity              tools/full_frame_parity.py over the whole 1080p frame of all four configs
#   final               GPU tests (both walks), driver-form bench, the other configs' whole jobs, rocprofv3 --stats of the bench command
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
part=$1; shift
case $part in
pmc)
  C=${1:-C2}
  tools/profile_config.sh r6/$C-prof $C
  CONFIG=$C tools/sq_profile.sh r6/$C-sq 64 > gpurun_out/r6/sq_$C.txt 2>&1; tail -n 30 gpurun_out/r6/sq_$C.txt | cut -c1-400 ;;
util)
  O=gpurun_out/r6; mkdir -p $O
  PT_LIB_PATH=pbrt-rust_amd/csrc/variants/qutil python bench.py --config C2 --spp 128 --steps 1 --warmup 0 --cpu-seconds 0 --other-configs off --projection off 2>&1 >/dev/null | grep trace-util | tee $O/C2_trace_util.txt
  PT_LIB_PATH=pbrt-rust_amd/csrc/variants/qutil python bench.py --config C4 --spp 32 --steps 1 --warmup 0 --cpu-seconds 0 --other-configs off --projection off 2>&1 >/dev/null | grep trace-util | tee $O/C4_trace_util.txt ;;
parity)
  O=gpurun_out/r6; mkdir -p $O
  timeout -k 10 1100 python tools/full_frame_parity.py C2:64 C5:32 C3:16 C4:8 > $O/full_frame_parity.jsonl 2> $O/parity_err.log
  python3 -c "
import json
for l in open('$O/full_frame_parity.jsonl'):
    d=json.loads(l); print(d['config'], d['spp'], d['samples'], 'differing', d['counters_differing'], 'weights', d['weights_identical'], 'rel', d['max_rel_diff_film'], 'linf', d['linf_normalised'], 'oracle s', d['oracle_render_s'])" ;;
final)
  O=gpurun_out/r6/final; mkdir -p $O
  echo "== GPU tests"; timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=12 > $O/gpu_tests.log 2>&1; tail -n 16 $O/gpu_tests.log
  echo "== bench (driver form)"; timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
  python3 - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
r=d['roofline']
print(d['value'], d['ms_per_step'], d['config']['spp_per_pass'], 'bound', r['bound'], 'frac', r['frac'], 'valu', r['valu_busy_frac'], 'code match', r['traffic_code_match'], 'algo/peak', r['algorithmic_over_hbm_peak'], 'avg launch ms', r['avg_launch_ms'])
print('cpu', {k: d['cpu_baseline'][k] for k in ('value','threads','single_thread_msamples_s','parallel_efficiency','cgroup_cpu_quota','physical_cores')}, 'readback', d['film_readback_ms'], d['value_with_readback'])
print({k:v['ms'] for k,v in d['kernels_ms_per_step'].items()})
print('projection', d.get('scaling_projection'))
print({c:(v.get('value'),v.get('ms_per_step'),v.get('hbm_frac'),v.get('scaling_projection',{}).get('efficiency')) for c,v in d.get('other_configs',{}).items()})
PY
  for C in C3 C4 C5; do
    timeout -k 10 500 python3 bench.py --config $C --steps 1 --warmup 1 --cpu-seconds 8 --projection off > $O/bench_$C.json 2>> $O/bench.err
    python3 -c "
import json;d=json.loads(open('$O/bench_$C.json').read().strip().split('\n')[-1]);r=d['roofline'];print('$C', d['value'], d['ms_per_step'], d['config']['spp_per_pass'], r['kernel'], 'frac', r['frac'], 'algo/peak', r['algorithmic_over_hbm_peak'], 'cpu', d['cpu_baseline']['value'] if d['cpu_baseline'] else None)"
  done
  cd /tmp && export TMPDIR=/tmp
  timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/stats -- python3 $REPO/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --other-configs off --projection off > $REPO/$O/stats_bench.json 2> $REPO/$O/stats.err
  cd $REPO; find $O -name '*kernel_trace.csv' -size +4M -delete
  python3 - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r6/final/stats/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]: print(r['Name'][:44], r['Calls'], round(float(r['TotalDurationNs'])/1e6,3), round(float(r['AverageNs'])/1e3,2))
PY
  # the record that gets committed: profiles/r6/final/ = these files + a manifest naming the code they were taken on (tools/check_profiles.py holds the tree against it)
  F=$O/record; rm -rf $F; mkdir -p $F
  cp $O/gpu_tests.log $O/bench.json $O/bench_C3.json $O/bench_C4.json $O/bench_C5.json $O/stats_bench.json $F/ 2>/dev/null
  find $O/stats -name '*kernel_stats.csv' | head -n 1 | xargs -I{} cp {} $F/kernel_stats_spp256.csv
  python3 tools/check_profiles.py --write-manifest $F
  ;;
esac
