#!/bin/bash
# round 5 lease: options for the k_paths_regen_split translation unit alone (all on top of max-ilp): if-conversion threshold 1, no loop unrolling,
# AMDGPU register-pressure trackers, schedule metric bias 30, -O2, no hoisting of common instructions, no LICM promotion.  C1.
set -u
L=pathtrace_amd
specs=""
for v in s_base s_phi1 s_nounroll s_trk s_bias s_o2 s_hoist s_nolicm; do specs="$specs $v=$L/libpt_$v.so:--workload_c1"; done
tools/ab.sh $(for v in s_base s_phi1 s_nounroll s_trk s_bias s_o2 s_hoist s_nolicm; do echo "$v=$L/libpt_$v.so:--workload=c1"; done) 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_ab_split_flags.txt
cat gpurun_out/r05_ab_split_flags.txt
