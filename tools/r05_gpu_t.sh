#!/bin/bash
# round 5 lease: the next pair's normal requested a pair ahead in the GENERIC-material kernels too (PT_PAIR_PREFETCH_GENERIC=1): small jobs on World::new()
set -u
for round in 1 2 3; do for lib in libpathtrace_amd.so libpt_pfg.so; do
  echo "== $lib (round $round)"; PATHTRACE_AMD_LIB=$PWD/pathtrace_amd/$lib python tools/small_batches.py 1 -1 1 2>&1 | grep -v amdgpu.ids
done; done > gpurun_out/r05_ab_pf_generic.txt
cat gpurun_out/r05_ab_pf_generic.txt
