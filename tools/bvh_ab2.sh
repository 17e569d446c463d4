#!/bin/bash
# usage: tools/bvh_ab2.sh lib...   C4 (spheres) and the 100k-triangle terrain with accel = 1
for so in "$@"; do echo "== $so"; PATHTRACE_AMD_LIB=$PWD/$so python tools/configs_gpu.py c4b | tail -1; PATHTRACE_AMD_LIB=$PWD/$so python tools/mesh_bench.py 224 1024 16 | tail -1; done
