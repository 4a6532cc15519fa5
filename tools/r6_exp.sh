#!/bin/bash
# round 6 quick measurements (one gpurun call): tools/r6_exp.sh <tag> <step>...
#   tests        the -m gpu suite (both walks) -> gpurun_out/r6/<tag>/gpu_tests.log
#   c2 | c3 | c4 | c5   one bench line of the config (C2: 3 steps of the headline; C3/C5: one pass; C4: 64 spp), compact summary printed
#   v2:<variant> C2 three steps with the variant only
#   ab:<variant> C2 three steps: the in-tree library, then pbrt-rust_amd/csrc/variants/<variant> (tools/build_variant.sh), same box
#   ab4:<variant> / ab5:<variant> / ab3:<variant>  the same for C4 (64 spp) / C5 / C3 (one pass each)
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
tag=$1; shift
O=gpurun_out/r6/$tag; mkdir -p $O
B="--cpu-seconds 0 --other-configs off --projection off"
run() {  # <label> <lib or ""> <bench args...>
  local label=$1 lib=$2; shift 2
  if [ -n "$lib" ]; then PT_LIB_PATH=$lib timeout -k 10 400 python3 bench.py $B "$@" > $O/$label.json 2>> $O/err.log
  else timeout -k 10 400 python3 bench.py $B "$@" > $O/$label.json 2>> $O/err.log; fi
  python3 tools/show_bench.py $O/$label.json $label || tail -n 5 $O/err.log
}
for step in "$@"; do
  case $step in
  tests) timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=8 > $O/gpu_tests.log 2>&1; tail -n 14 $O/gpu_tests.log ;;
  c2) run c2 "" --steps 3 --warmup 1 ;;
  c3) run c3 "" --config C3 --steps 1 --warmup 1 --spp -1 ;;
  c4) run c4 "" --config C4 --steps 1 --warmup 1 --spp 64 ;;
  c5) run c5 "" --config C5 --steps 1 --warmup 1 --spp -1 ;;
  v2:*) v=${step#v2:}; run c2_$v pbrt-rust_amd/csrc/variants/$v --steps 3 --warmup 1 ;;   # one variant, no bracketing tree runs (several variants in one call: start and end with c2)
  ab:*) v=${step#ab:}; run c2_tree "" --steps 3 --warmup 1; run c2_$v pbrt-rust_amd/csrc/variants/$v --steps 3 --warmup 1; run c2_tree2 "" --steps 3 --warmup 1 ;;
  ab3:*) v=${step#ab3:}; run c3_tree "" --config C3 --steps 1 --warmup 1 --spp -1; run c3_$v pbrt-rust_amd/csrc/variants/$v --config C3 --steps 1 --warmup 1 --spp -1 ;;
  ab4:*) v=${step#ab4:}; run c4_tree "" --config C4 --steps 1 --warmup 1 --spp 64; run c4_$v pbrt-rust_amd/csrc/variants/$v --config C4 --steps 1 --warmup 1 --spp 64 ;;
  ab5:*) v=${step#ab5:}; run c5_tree "" --config C5 --steps 1 --warmup 1 --spp -1; run c5_$v pbrt-rust_amd/csrc/variants/$v --config C5 --steps 1 --warmup 1 --spp -1 ;;
  esac
done
