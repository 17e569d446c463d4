#!/bin/bash
# Build a variant of the library for same-box A/B runs (tools/ab.sh):
#   tools/build_variant.sh <name> [extra hipcc flags, e.g. -DPT_BOUNCE_WAVES_LDS=7]   ->  pathtrace_amd/libpt_<name>.so
# Only the kernels (pt_kernels.hip, both arithmetic modes) are recompiled; the host objects of the regular build are reused.
set -e
cd "$(dirname "$0")/../pathtrace_amd/csrc"
name=$1; shift
make -j4 >/dev/null
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-parameter"
mkdir -p /tmp/ptvar_$name
/opt/rocm/bin/hipcc $FLAGS -DPT_MATH_EXACT=1 "$@" -c pt_kernels.hip -o /tmp/ptvar_$name/k1.o &
/opt/rocm/bin/hipcc $FLAGS -DPT_MATH_EXACT=0 "$@" -c pt_kernels.hip -o /tmp/ptvar_$name/k0.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpt_$name.so /tmp/ptvar_$name/k1.o /tmp/ptvar_$name/k0.o pt_api.o pt_bvh.o pt_scenes.o pt_multi.o pt_sched.o -ldl
echo "built pathtrace_amd/libpt_$name.so"
