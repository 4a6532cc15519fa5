#!/bin/bash
# builds every variant listed in tools/variants.txt into pbrt-rust_amd/csrc/variants/<name> (tools/build_variant.sh does one)
REPO="$(cd "$(dirname "$0")/.." && pwd)"
grep -v "^#" $REPO/tools/variants.txt | sed "s/#.*//" | while read name flags; do [ -n "$name" ] && $REPO/tools/build_variant.sh $name $flags; done
