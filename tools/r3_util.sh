#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r3_util
export PT_LIB_PATH=pbrt-rust_amd/csrc/variants/util
for cfg in "C4 --spp 32" "C2 --spp 128" "C3 --spp 128"; do
  set -- $cfg
  echo "== $cfg"
  timeout -k 10 300 python bench.py --config $1 $2 $3 --steps 1 --warmup 0 --cpu-seconds 0 --other-configs off > gpurun_out/r3_util/$1.json 2> gpurun_out/r3_util/$1.err
  grep "trace-util" gpurun_out/r3_util/$1.err | tail -12
done
