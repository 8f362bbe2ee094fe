#!/bin/bash
# CPU suite (-m "not gpu") against the host-sanitised build of the library (ASan + UBSan on bc_host.cpp,
# bc_ingest.cpp, bc_sample.cpp and the host halves of the .hip files).  usage: bash tests/run_asan.sh
set -e
cd "$(dirname "$0")/.."
make -C barcoder_amd/csrc asan
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
BC_NATIVE_SO=$PWD/build/libbarcoder_hip_asan.so LD_PRELOAD=$RT \
  ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest tests -q -m "not gpu" -p no:cacheprovider "$@"
