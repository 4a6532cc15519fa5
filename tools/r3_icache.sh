#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
tools/r3_ab.sh default nimath
for v in default nimath; do
  if [ "$v" = default ]; then unset PT_LIB_PATH; else export PT_LIB_PATH=$PWD/pbrt-rust_amd/csrc/variants/$v; fi
  echo "== icache counters: $v"
  PMC_SPP=64 tools/pmc_pass.sh r3_icache_$v "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU" 2>&1 | grep -E "pass|k_shade|k_trace<2"
done
