#!/bin/bash
# A/B of library variants over the four configs: tools/r3_ab.sh <variant...>   (variants live in pbrt-rust_amd/csrc/variants/, "default" = the in-tree library)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for spec in "--config C2 --steps 2" "--config C3 --spp 256 --steps 1" "--config C5 --spp 256 --steps 1" "--config C4 --spp 32 --steps 1"; do
  echo "== $spec"
  tools/variant_ab.sh $spec --warmup 1 --other-configs off -- "$@"
done
