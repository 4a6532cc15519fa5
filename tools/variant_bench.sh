#!/bin/bash
# usage: tools/variant_bench.sh <variant.so | default> ...   -> one compact line per variant (kernel-variant experiments)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
for v in "$@"; do
  if [ "$v" = default ]; then unset PT_LIB_PATH; else export PT_LIB_PATH="$PWD/pbrt-rust_amd/csrc/variants/$v"; fi
  python bench.py --steps 2 --warmup 1 --cpu-seconds 0.5 2>/dev/null | tail -1 > gpurun_out/variant_$v.json
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
d = json.load(open(f"gpurun_out/variant_{v}.json"))
k = d["kernels_ms_per_step"]
print(v, "value", d["value"], "ms", d["ms_per_step"], {n: round(x["ms"], 1) for n, x in k.items()})
PY
done
