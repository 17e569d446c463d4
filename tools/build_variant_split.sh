#!/bin/bash
# Variant of the library whose k_paths_regen_split translation unit is compiled with extra options (the other units as in the Makefile):
#   tools/build_variant_split.sh <name> [hipcc flags]   ->  pathtrace_amd/libpt_<name>.so
set -eu
name=${1:?usage: tools/build_variant_split.sh <name> [flags]}; shift
case "$name" in */*|.*|"") echo "bad name: $name" >&2; exit 2;; esac
root="$(cd "$(dirname "$0")/.." && pwd)"
cd "$root/pathtrace_amd/csrc"
make -j6 >/dev/null
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-parameter -DPT_TU=2"
mkdir -p /tmp/ptvs_$name
/opt/rocm/bin/hipcc $FLAGS -DPT_MATH_EXACT=1 "$@" -c pt_kernels.hip -o /tmp/ptvs_$name/s1.o &
/opt/rocm/bin/hipcc $FLAGS -DPT_MATH_EXACT=0 "$@" -c pt_kernels.hip -o /tmp/ptvs_$name/s0.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpt_$name.so pt_kernels_exact.o pt_kernels_fast.o /tmp/ptvs_$name/s1.o /tmp/ptvs_$name/s0.o \
    pt_kernels_bvh_exact.o pt_kernels_bvh_fast.o pt_api.o pt_bvh.o pt_scenes.o pt_multi.o pt_sched.o -ldl
echo "built pathtrace_amd/libpt_$name.so"
