#!/bin/bash
# usage: tools/build_variant.sh <name> <-Dflags...>   -> pbrt-rust_amd/csrc/variants/<name> (a libmi355pt.so built with extra flags)
cd "$(dirname "$0")/../pbrt-rust_amd/csrc" || exit 1
name=$1; shift
mkdir -p variants/obj_$name
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-result -Wno-unused-value "$@" -c -o variants/obj_$name/capi.o capi.hip &&
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o variants/$name variants/obj_$name/capi.o gpu_bvh.o host_bvh.o tables_blob.o && rm -rf variants/obj_$name && echo "built $name"
