#!/bin/bash
# Build a variant of the WHOLE library (host objects too) for same-box A/B runs (tools/ab.sh):
#   tools/build_variant_full.sh <name> [extra hipcc flags, e.g. -DPT_BVH_LEAF_TARGET=2]   ->  pathtrace_amd/libpt_<name>.so
set -eu
name=${1:?usage: tools/build_variant_full.sh <name> [flags]}; shift
case "$name" in */*|.*|"") echo "bad name: $name" >&2; exit 2;; esac
root="$(cd "$(dirname "$0")/.." && pwd)"
tmp=/tmp/ptvfull_$name
rm -rf "$tmp"; mkdir -p "$tmp/pathtrace_amd" "$tmp/include"
cp -r "$root/pathtrace_amd/csrc" "$tmp/pathtrace_amd/"; cp "$root"/include/*.h "$tmp/include/"
rm -f "$tmp"/pathtrace_amd/csrc/*.o
make -C "$tmp/pathtrace_amd/csrc" -j6 FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-parameter $*" >/dev/null
cp "$tmp/pathtrace_amd/libpathtrace_amd.so" "$root/pathtrace_amd/libpt_$name.so"
echo "built pathtrace_amd/libpt_$name.so"
