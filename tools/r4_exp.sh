#!/bin/bash
# round-4 traversal experiments on the GPU box: tools/r4_exp.sh <tag> <what...>
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
tag=$1; shift
OUT=gpurun_out/$tag; mkdir -p $OUT
one() {   # one <label> <bench args...>   (env of the caller applies)
  local label=$1; shift
  python bench.py "$@" --cpu-seconds 0 --other-configs off 2>$OUT/err_$label.log | tail -1 > $OUT/bench_$label.json || { tail -5 $OUT/err_$label.log; return 1; }
  python3 - $OUT/bench_$label.json "$label" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); k=d['kernels_ms_per_step']
print('%-22s %9.2f Msamples/s  %8.2f ms/step  ' % (sys.argv[2], d['value'], d['ms_per_step']) + ' '.join('%s=%.1f' % (n, x['ms']) for n, x in k.items() if x['ms'] >= 1.0), flush=True)
PY
}
for what in "$@"; do
case $what in
exact_tests) PT_TRACE_EXACT=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_configs.py -m gpu -x -q 2>&1 | tail -5 ;;
parity) timeout -k 10 900 python tools/full_frame_parity.py C2:16 C5:8 C3:8 C4:4 > $OUT/full_frame_parity.jsonl 2>$OUT/parity_err.log; python3 -c "
import json
for l in open('$OUT/full_frame_parity.jsonl'):
    d=json.loads(l); print(d['config'], d['spp'], 'differing', d['counters_differing'], 'weights', d['weights_identical'], 'rel', d['max_rel_diff_film'], 'linf', d['linf_normalised'])
" ;;
c2sweep)
  PT_TRACE_EXACT=1 one c2_exact --config C2 --steps 2 --warmup 1
  for q in 8 16 24 32 48; do PT_TRACE_LEAF_QUORUM=$q one c2_spec_q$q --config C2 --steps 2 --warmup 1; done ;;
c4sweep)
  PT_TRACE_EXACT=1 one c4_exact --config C4 --spp 64 --steps 1 --warmup 1
  for q in 8 16 32; do PT_TRACE_LEAF_QUORUM=$q one c4_spec_q$q --config C4 --spp 64 --steps 1 --warmup 1; done ;;
c35)
  PT_TRACE_EXACT=1 one c3_exact --config C3 --spp 256 --steps 1 --warmup 1
  for q in 8 24; do PT_TRACE_LEAF_QUORUM=$q one c3_spec_q$q --config C3 --spp 256 --steps 1 --warmup 1; done
  PT_TRACE_EXACT=1 one c5_exact --config C5 --spp 216 --steps 1 --warmup 1
  for q in 8 24; do PT_TRACE_LEAF_QUORUM=$q one c5_spec_q$q --config C5 --spp 216 --steps 1 --warmup 1; done ;;
base)
  one c2 --config C2 --steps 3 --warmup 1
  one c4 --config C4 --spp 64 --steps 1 --warmup 1 ;;
base35)
  one c3 --config C3 --spp 256 --steps 1 --warmup 1
  one c5 --config C5 --spp 216 --steps 2 --warmup 1 ;;
both)
  one c2_quad --config C2 --steps 3 --warmup 1
  PT_TRACE_EXACT=1 one c2_exact --config C2 --steps 2 --warmup 1
  one c4_quad --config C4 --spp 64 --steps 1 --warmup 1
  PT_TRACE_EXACT=1 one c4_exact --config C4 --spp 64 --steps 1 --warmup 1 ;;
exact_quick) PT_TRACE_EXACT=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -5 ;;
knobs)
  for q in 4 12 16 24; do PT_TRACE_LEAF_QUORUM=$q one c2_lq$q --config C2 --steps 2 --warmup 1; done
  for r in 8 24 32; do PT_TRACE_REFILL_MIN=$r one c2_rf$r --config C2 --steps 2 --warmup 1; done
  for w in 20 24; do PT_TRACE_WAVES_PER_CU=$w one c2_w$w --config C2 --steps 2 --warmup 1; done ;;
stage)
  for r in 4 8 16; do PT_TRACE_REFILL_MIN=$r one c2_stage_rf$r --config C2 --steps 2 --warmup 1; done
  PT_TRACE_STAGE=0 one c2_nostage --config C2 --steps 2 --warmup 1
  for r in 4 8; do PT_TRACE_REFILL_MIN=$r one c4_stage_rf$r --config C4 --spp 64 --steps 1 --warmup 1; done
  PT_TRACE_STAGE=0 one c4_nostage --config C4 --spp 64 --steps 1 --warmup 1 ;;
streams)
  for v in 1 0; do PT_SHADE_STREAMS=$v one c2_streams$v --config C2 --steps 3 --warmup 1; PT_SHADE_STREAMS=$v one c3_streams$v --config C3 --spp 256 --steps 1 --warmup 1; PT_SHADE_STREAMS=$v one c5_streams$v --config C5 --spp 216 --steps 2 --warmup 1; done ;;
knobs2)
  for r in 16,16,16,32 16,16,16,64 16,16,16,24 12,12,12,48 20,20,20,48; do PT_TRACE_REFILL_MIN=$r one c2_rf_$r --config C2 --steps 2 --warmup 1; done
  for q in 8,8,8,16 8,8,8,4 2,2,2,8; do PT_TRACE_LEAF_QUORUM=$q one c2_lq_$q --config C2 --steps 2 --warmup 1; done
  for v in qlds14 qchunk512 qchunk128; do PT_LIB_PATH=pbrt-rust_amd/csrc/variants/$v one c2_$v --config C2 --steps 2 --warmup 1; done ;;
knobs4)
  for q in 8 12 24; do PT_TRACE_INST_QUORUM=$q one c4_iq$q --config C4 --spp 64 --steps 1 --warmup 1; done
  for q in 4 12 16; do PT_TRACE_LEAF_QUORUM=$q one c4_lq$q --config C4 --spp 64 --steps 1 --warmup 1; done
  for r in 4 12 16; do PT_TRACE_REFILL_MIN=$r one c4_rf$r --config C4 --spp 64 --steps 1 --warmup 1; done ;;
metal)
  one c3_metal3 --config C3 --spp 256 --steps 1 --warmup 1
  PT_SHADE_SPECIALISE=0 one c3_general --config C3 --spp 256 --steps 1 --warmup 1
  PT_LIB_PATH=pbrt-rust_amd/csrc/variants/metal2 one c3_metal2 --config C3 --spp 256 --steps 1 --warmup 1
  timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_configs.py tests/test_wavefront_schedule.py -m gpu -x -q -k 'zoo or c3 or specular' 2>&1 | tail -3 ;;
p2)
  one c3_default --config C3 --spp 256 --steps 1 --warmup 1
  PT_LIB_PATH=pbrt-rust_amd/csrc/variants/p2w3 one c3_p2w3 --config C3 --spp 256 --steps 1 --warmup 1 ;;
sss)
  one c5_spec --config C5 --spp 216 --steps 2 --warmup 1
  PT_SHADE_SPECIALISE=0 one c5_general --config C5 --spp 216 --steps 2 --warmup 1
  timeout -k 10 800 python -m pytest tests/test_gpu_parity.py tests/test_configs.py tests/test_golden.py -m gpu -x -q -k 'subsurface or c5 or probe or golden' 2>&1 | tail -3 ;;
ab_old)
  for i in 1 2; do one c2_new$i --config C2 --steps 3 --warmup 1; PT_LIB_PATH=pbrt-rust_amd/csrc/variants/oldblock one c2_old$i --config C2 --steps 3 --warmup 1; done
  one c4_new --config C4 --spp 64 --steps 1 --warmup 1; PT_LIB_PATH=pbrt-rust_amd/csrc/variants/oldblock one c4_old --config C4 --spp 64 --steps 1 --warmup 1
  one c3_new --config C3 --spp 256 --steps 1 --warmup 1; PT_LIB_PATH=pbrt-rust_amd/csrc/variants/oldblock one c3_old --config C3 --spp 256 --steps 1 --warmup 1 ;;
v5w)
  PT_LIB_PATH=pbrt-rust_amd/csrc/variants/q5w one c2_q5w --config C2 --steps 3 --warmup 1 ;;
variants)
  for v in q4w q6w; do PT_LIB_PATH=pbrt-rust_amd/csrc/variants/$v one c2_$v --config C2 --steps 2 --warmup 1; done
  PT_LIB_PATH=pbrt-rust_amd/csrc/variants/q4w PT_TRACE_WAVES_PER_CU=16 one c2_q4w_w16 --config C2 --steps 2 --warmup 1
  PT_LIB_PATH=pbrt-rust_amd/csrc/variants/q6w PT_TRACE_WAVES_PER_CU=24 one c2_q6w_w24 --config C2 --steps 2 --warmup 1 ;;
util)
  PT_LIB_PATH=pbrt-rust_amd/csrc/variants/qutil python bench.py --config C2 --spp 128 --steps 1 --warmup 0 --cpu-seconds 0 --other-configs off 2>&1 >/dev/null | grep trace-util | tee $OUT/c2_util.txt
  PT_LIB_PATH=pbrt-rust_amd/csrc/variants/qutil python bench.py --config C4 --spp 32 --steps 1 --warmup 0 --cpu-seconds 0 --other-configs off 2>&1 >/dev/null | grep trace-util | tee $OUT/c4_util.txt ;;
quick_tests) timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -5 ;;
esac
done
