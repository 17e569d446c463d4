#!/bin/bash
# ASan + UBSan over the host-only code paths (no GPU): scenes, cameras, BVH builder/verifier, oracle.
# pt_api.cpp needs the HIP runtime headers but none of the calls made here touch a device.
set -e
cd "$(dirname "$0")/../.."
OUT=${1:-/tmp/pt_san_driver}
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1"
HIPINC="-D__HIP_PLATFORM_AMD__ -I/opt/rocm/include"
/opt/rocm/lib/llvm/bin/clang++ -std=c++17 $SAN $HIPINC -Wno-unused-value -x c++ \
    tests/tools/san_driver.cpp pathtrace_amd/csrc/pt_bvh.cpp pathtrace_amd/csrc/pt_scenes.cpp pathtrace_amd/csrc/pt_api.cpp pathtrace_amd/csrc/pt_multi.cpp \
    -x c++ oracle/oracle_capi.cpp -ffp-contract=off \
    -DPT_SAN_NO_KERNELS -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -lpthread -ldl -o "$OUT" tests/tools/san_stubs.cpp
ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 "$OUT"
