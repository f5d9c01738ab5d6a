#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer and ThreadSanitizer runs of the multi-threaded host code (plan.cpp,
# reorder.cpp, amg_setup.cpp, plan_api.cpp) under the CPU tests that drive it.  CPU only: never on the GPU box
# (GPU sanitizers are not available on this pool).  usage: tools/run_sanitizers.sh [asan|tsan]   (default: both)
set -euo pipefail
cd "$(dirname "$0")/.."
make -C fem-shell_amd/csrc -s san
TESTS="tests/test_plan_cpu.py tests/test_amg_host.py"
# tests that need the full library (symbol inventory of libfemshell.so, the Python reorder mirror against the device build)
SKIP='not exports_every and not no_device_fails'
run() {
    local kind=$1 rt=$2; shift 2
    echo "== $kind =="
    env "$@" FEMSHELL_HOST_LIBRARY="$PWD/fem-shell_amd/libfemshell_host_${kind}.so" FEMSHELL_HOST_THREADS=8 \
        LD_PRELOAD="$(gcc -print-file-name=$rt)" python -m pytest $TESTS -x -q -m "not gpu" -k "$SKIP" -p no:cacheprovider
}
what=${1:-both}
if [ "$what" = asan ] || [ "$what" = both ]; then
    run asan libasan.so ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
fi
if [ "$what" = tsan ] || [ "$what" = both ]; then
    run tsan libtsan.so TSAN_OPTIONS="halt_on_error=1 report_signal_unsafe=0 ignore_noninstrumented_modules=1"
fi
echo "sanitizer runs clean"
