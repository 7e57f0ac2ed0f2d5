#!/bin/bash
# Runs the no-GPU tests of the host side (plan / arena layout / argument validation / ABI table) against the AddressSanitizer + UBSan
# build of the library (make -C diffusion-deconvolution-dia-msms-data_amd asan).  CPU only: no GPU sanitizer runs on this pool.
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
LIB=$REPO/diffusion-deconvolution-dia-msms-data_amd/build/asan/libdq_hip.so
[ -f "$LIB" ] || { echo "build it first: make -C diffusion-deconvolution-dia-msms-data_amd asan"; exit 2; }
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
cd "$REPO"
# detect_leaks=0: the interpreter itself is not leak-clean; every other ASan / UBSan report aborts the run
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
  DQ_HIP_LIB=$LIB python -m pytest tests/test_abi.py "tests/test_generic_config.py" "tests/test_host_logic.py" -q -m "not gpu" -p no:cacheprovider "$@"
