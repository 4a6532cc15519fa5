#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer run of the CPU test-suite: the oracle, the .pbrt front end and the host SAH builder
# are rebuilt with -fsanitize=address,undefined (make SAN=1) and loaded into an uninstrumented python with the sanitizer runtimes
# preloaded. CPU only (GPU AddressSanitizer is not available on this pool). usage: tools/san_cpu_tests.sh [pytest args]
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd $REPO
export PT_SAN=1
make -C oracle SAN=1 || exit 1
make -C pbrt-rust_amd/frontend SAN=1 || exit 1
ASAN_LIB=$(g++ -print-file-name=libasan.so); UBSAN_LIB=$(g++ -print-file-name=libubsan.so)
export LD_PRELOAD=$ASAN_LIB:$UBSAN_LIB
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1:detect_odr_violation=0
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
python -m pytest tests -m "not gpu" -x -q -p no:cacheprovider "$@"
