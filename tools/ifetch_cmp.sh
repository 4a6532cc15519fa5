#!/bin/bash
# compare instruction-fetch counters of two builds of the library (one headline-size pass each)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
PMC_SPP=32 tools/pmc_pass.sh if_new "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES" | grep -E "pass|k_shade"
export PT_LIB_PATH=$REPO/pbrt-rust_amd/csrc/variants/old.so
PMC_SPP=32 tools/pmc_pass.sh if_old "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES" | grep -E "pass|k_shade"
