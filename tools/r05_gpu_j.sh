#!/bin/bash
# round 5 lease: pair test with o - v0 before the determinant's test (PT_PAIR_S_EARLY: v0 requested with the normal), with (se) and without (se0)
# the prefetch of the next normal; base = new default (prefetch in the split kernel, rem2)
set -u
L=pathtrace_amd
tools/ab.sh "base=$L/libpathtrace_amd.so:--workload c1" "se=$L/libpt_se.so:--workload c1" "se0=$L/libpt_se0.so:--workload c1" "base_c2=$L/libpathtrace_amd.so" 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_ab_s_early.txt
cat gpurun_out/r05_ab_s_early.txt
