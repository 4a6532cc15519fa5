#!/bin/bash
# Run a list of GPU steps one after another under `timeout -k 10`, logging each to gpurun_out/<tag>/<name>.log.
# A step that times out or is killed (exit 124 / 137 / 143) ends the sequence: nothing further is started on that box.
# usage: tools/gpu_steps.sh <tag> <name>:<seconds>:<command> ...
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
for spec in "$@"; do
  name=${spec%%:*}; rest=${spec#*:}; secs=${rest%%:*}; cmd=${rest#*:}
  echo "== $name (limit ${secs}s): $cmd"
  start=$(date +%s)
  timeout -k 10 $secs bash -c "$cmd" > $OUT/$name.log 2> $OUT/$name.err
  rc=$?
  echo "   rc=$rc in $(( $(date +%s) - start ))s"; tail -c 1500 $OUT/$name.log | tail -n 6
  if [ $rc -ne 0 ]; then tail -n 15 $OUT/$name.err; fi
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 143 ]; then echo "step $name was killed: stopping"; exit $rc; fi
done
exit 0
