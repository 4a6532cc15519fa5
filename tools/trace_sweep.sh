#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "64 64" "64 20" "64 8" "16 20" "32 20" "16 64" "1 20"; do
  set -- $cfg
  echo "== refill_min=$1 leaf_quorum=$2"
  PT_TRACE_REFILL_MIN=$1 PT_TRACE_LEAF_QUORUM=$2 python tools/trace_bench.py 2>&1 | grep case
done
