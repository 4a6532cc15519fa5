#!/bin/bash
# usage: tools/build_variant.sh <name> <-Dflags...>   -> pbrt-rust_amd/csrc/variants/<name> (a libmi355pt.so built with extra flags)
# Builds in a scratch copy of csrc/ (make EXTRA_HIPFLAGS=...), so the in-tree objects stay those of the default build.
REPO="$(cd "$(dirname "$0")/.." && pwd)"
name=$1; shift
scratch=/tmp/pt_variant_$name
rm -rf $scratch && mkdir -p $scratch/pbrt-rust_amd && cp -r $REPO/pbrt-rust_amd/csrc $scratch/pbrt-rust_amd/csrc && cp -r $REPO/pbrt-rust_amd/data $scratch/pbrt-rust_amd/data && cp -r $REPO/include $scratch/include
rm -f $scratch/pbrt-rust_amd/csrc/*.o $scratch/pbrt-rust_amd/csrc/libmi355pt.so; rm -rf $scratch/pbrt-rust_amd/csrc/variants
make -C $scratch/pbrt-rust_amd/csrc -j8 EXTRA_HIPFLAGS="$*" > $scratch/build.log 2>&1 || { tail -20 $scratch/build.log; exit 1; }
mkdir -p $REPO/pbrt-rust_amd/csrc/variants && cp $scratch/pbrt-rust_amd/csrc/libmi355pt.so $REPO/pbrt-rust_amd/csrc/variants/$name && echo "built $name"
