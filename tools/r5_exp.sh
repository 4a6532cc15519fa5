#!/bin/bash
# round-5 A/B runs on the GPU box: tools/r5_exp.sh <tag> <what...>      (variants: tools/build_variant.sh <name> <-D flags>, listed in tools/variants.txt)
#   ab:<variant>[,<variant>...]   C2 (3 steps) with the in-tree library and with each pbrt-rust_amd/csrc/variants/<variant>, twice, interleaved
#   ab4:<variant>[,...]           the same on C4 at 64 spp;  ab3 / ab5: C3 at 256 spp / C5 at 216 spp
#   quickv:<variant>              parity + golden + config tests with a variant library;  quick / full: the in-tree library
set -o pipefail
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
lib() { if [ "$1" = tree ]; then echo ""; else echo "pbrt-rust_amd/csrc/variants/$1"; fi; }
for what in "$@"; do
case $what in
ab:*)  vs="tree,${what#ab:}"; for i in 1 2; do for v in ${vs//,/ }; do PT_LIB_PATH=$(lib $v) one c2_${v}_$i --config C2 --steps 3 --warmup 1 || exit 1; done; done ;;
ab4:*) vs="tree,${what#ab4:}"; for v in ${vs//,/ }; do PT_LIB_PATH=$(lib $v) one c4_${v} --config C4 --spp 64 --steps 1 --warmup 1 || exit 1; done ;;
ab3:*) vs="tree,${what#ab3:}"; for v in ${vs//,/ }; do PT_LIB_PATH=$(lib $v) one c3_${v} --config C3 --spp 256 --steps 1 --warmup 1 || exit 1; done ;;
knobs4)
  for q in 4 12 16; do PT_TRACE_INST_QUORUM=$q one c4_iq$q --config C4 --spp 64 --steps 1 --warmup 1; done
  for q in 4 12 16; do PT_TRACE_LEAF_QUORUM=$q one c4_lq$q --config C4 --spp 64 --steps 1 --warmup 1; done
  for q in 4 12; do PT_TRACE_REFILL_MIN=$q one c4_rf$q --config C4 --spp 64 --steps 1 --warmup 1; done ;;
shardab:*)  vs="tree,${what#shardab:}"; for v in ${vs//,/ }; do PT_LIB_PATH=$(lib $v) one sh_${v} --config C2 --sim-world 8 --steps 5 --warmup 2 --projection off || exit 1; done ;;
ranks) for r in 0 1 2 3 4 5 6 7; do one sh_rank$r --config C2 --sim-world 8 --sim-rank $r --steps 3 --warmup 1 --projection off || exit 1; done ;;
shard)   # rank 0's shard of an 8-rank C2 job: one render, and two / three half-shards in flight on the same device (pt_multi_render, replicas 0,0[,0])
  one sh_1 --config C2 --sim-world 8 --steps 5 --warmup 2 --projection off || exit 1
  one sh_2 --config C2 --sim-world 8 --steps 5 --warmup 2 --projection off --gpus 2 --devices 0,0 --in-process || exit 1
  one sh_3 --config C2 --sim-world 8 --steps 5 --warmup 2 --projection off --gpus 3 --devices 0,0,0 --in-process || exit 1 ;;
big) one c2_plain --config C2 --steps 3 --warmup 1 || exit 1; PT_TEST_POOL_PAD_RECORDS=34000000 one c2_bigpool --config C2 --steps 3 --warmup 1 || exit 1
     one c4_plain --config C4 --spp 64 --steps 1 --warmup 1 || exit 1; PT_TEST_POOL_PAD_RECORDS=34000000 one c4_bigpool --config C4 --spp 64 --steps 1 --warmup 1 || exit 1 ;;
gateq) for q in 8 10 12 16 20; do PT_TRACE_INST_QUORUM=$q PT_LIB_PATH=pbrt-rust_amd/csrc/variants/gate one c4_gate_iq$q --config C4 --spp 64 --steps 1 --warmup 1; done; one c4_tree --config C4 --spp 64 --steps 1 --warmup 1 ;;
c5proj) for v in tree base; do PT_LIB_PATH=$(lib $v) one c5p_$v --config C5 --steps 1 --warmup 1 --projection on || exit 1; python3 -c "
import json; d=json.loads(open('$OUT/bench_c5p_$v.json').read().strip().split('\n')[-1]); print('$v', d['config']['spp_per_pass'], d.get('scaling_projection'))"; done ;;
nofin)  # A/B of the film kernel that ends the paths against the per-iteration miss pass (PT_FILM_FINAL=0), same library
  for i in 1 2; do one c2_fin_$i --config C2 --steps 3 --warmup 1 || exit 1; PT_FILM_FINAL=0 one c2_miss_$i --config C2 --steps 3 --warmup 1 || exit 1; done
  one c3_fin --config C3 --spp 256 --steps 1 --warmup 1 || exit 1; PT_FILM_FINAL=0 one c3_miss --config C3 --spp 256 --steps 1 --warmup 1 || exit 1
  one c4_fin --config C4 --spp 64 --steps 1 --warmup 1 || exit 1; PT_FILM_FINAL=0 one c4_miss --config C4 --spp 64 --steps 1 --warmup 1 || exit 1 ;;
nofin5) one c5_fin --config C5 --spp 216 --steps 2 --warmup 1 || exit 1; PT_FILM_FINAL=0 one c5_miss --config C5 --spp 216 --steps 2 --warmup 1 || exit 1 ;;
aos) PT_LIB_PATH=pbrt-rust_amd/csrc/variants/aos timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_configs.py -m gpu -x -q 2>&1 | tail -4
  for i in 1 2; do one c2_tree_$i --config C2 --steps 3 --warmup 1 || exit 1; PT_LIB_PATH=pbrt-rust_amd/csrc/variants/aos one c2_aos_$i --config C2 --steps 3 --warmup 1 || exit 1; done
  one c3_tree --config C3 --spp 256 --steps 1 --warmup 1; PT_LIB_PATH=pbrt-rust_amd/csrc/variants/aos one c3_aos --config C3 --spp 256 --steps 1 --warmup 1
  one c4_tree --config C4 --spp 64 --steps 1 --warmup 1; PT_LIB_PATH=pbrt-rust_amd/csrc/variants/aos one c4_aos --config C4 --spp 64 --steps 1 --warmup 1
  one c5_tree --config C5 --spp 216 --steps 2 --warmup 1; PT_LIB_PATH=pbrt-rust_amd/csrc/variants/aos one c5_aos --config C5 --spp 216 --steps 2 --warmup 1 ;;
c3m) one c3_plain --config C3 --spp 256 --steps 1 --warmup 1 || exit 1; one c3_mixed --config C3M --spp 256 --steps 1 --warmup 1 || exit 1 ;;
ab5:*) vs="tree,${what#ab5:}"; for v in ${vs//,/ }; do PT_LIB_PATH=$(lib $v) one c5_${v} --config C5 --spp 216 --steps 2 --warmup 1 || exit 1; done ;;
quickv:*) PT_LIB_PATH=$(lib ${what#quickv:}) timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_configs.py -m gpu -x -q 2>&1 | tail -5 ;;
quick) timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -5 || exit 1 ;;
inst) timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_configs.py -m gpu -x -q -k "instanc or c4 or foliage or garden" 2>&1 | tail -5 || exit 1 ;;
full) timeout -k 10 1100 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 || exit 1 ;;
parity) timeout -k 10 900 python tools/full_frame_parity.py C2:16 C5:8 C3:8 C4:4 > $OUT/full_frame_parity.jsonl 2>$OUT/parity_err.log; python3 -c "
import json
for l in open('$OUT/full_frame_parity.jsonl'):
    d=json.loads(l); print(d['config'], d['spp'], 'differing', d['counters_differing'], 'weights', d['weights_identical'], 'rel', d['max_rel_diff_film'], 'linf', d['linf_normalised'])
" ;;
*) echo "unknown step $what"; exit 2 ;;
esac
done
