#!/bin/bash
# usage: tools/bvh_ab.sh lib1.so lib2.so ...   (C4 with accel=1, 64 spp; 2 interleaved rounds)
for r in 1 2; do for so in "$@"; do echo -n "$so: "; PATHTRACE_AMD_LIB=$PWD/$so python tools/configs_gpu.py c4b | tail -1; done; done
