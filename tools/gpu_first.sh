#!/bin/bash
# first GPU contact: parity tests + smoke
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log
python __graft_entry__.py smoke 2>&1 | tail -5
