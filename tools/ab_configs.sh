#!/bin/bash
# A/B of the default library against pbrt-rust_amd/csrc/variants/<variant> on several configs, alternating within one box:
# tools/ab_configs.sh <variant.so> "<env for the variant>" C2:256 C3:128 ...
V=$1; VENV=$2; shift 2
for cs in "$@"; do
  cfg=${cs%%:*}; spp=${cs#*:}
  echo "== $cfg at $spp spp"
  tools/variant_ab.sh --config $cfg --spp $spp --steps 1 --warmup 1 -- default
  env $VENV tools/variant_ab.sh --config $cfg --spp $spp --steps 1 --warmup 1 -- $V
  tools/variant_ab.sh --config $cfg --spp $spp --steps 1 --warmup 1 -- default
done
