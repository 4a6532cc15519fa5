#!/bin/bash
# Sobol' nibble tables: the GPU suite on the new library, then A/B against the previous build (variants/pre_nib)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3_nib; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -n 3 $O/gpu_tests.log
tools/r3_ab.sh pre_nib default 2>&1 | tee $O/ab.txt
