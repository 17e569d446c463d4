#!/bin/bash
# round 5 lease: compiler options on top of -fno-slp-vectorize + PT_SPLIT_REPLACE (nb): scheduling strategies, if-conversion thresholds,
# early if-conversion; the Mirror-only batch code (mirnb).  C2 and C1, 3 interleaved rounds each.
set -u
L=pathtrace_amd
tools/ab.sh "nb=$L/libpt_nb.so" "maxilp=$L/libpt_maxilp.so" "iterilp=$L/libpt_iterilp.so" "minreg=$L/libpt_minreg.so" "phi1=$L/libpt_phi1.so" "phi8=$L/libpt_phi8.so" "eifcvt=$L/libpt_eifcvt.so" 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_ab_flags_c2.txt
cat gpurun_out/r05_ab_flags_c2.txt
tools/ab.sh "nb=$L/libpt_nb.so:--workload c1" "maxilp=$L/libpt_maxilp.so:--workload c1" "iterilp=$L/libpt_iterilp.so:--workload c1" "minreg=$L/libpt_minreg.so:--workload c1" "phi1=$L/libpt_phi1.so:--workload c1" "phi8=$L/libpt_phi8.so:--workload c1" "eifcvt=$L/libpt_eifcvt.so:--workload c1" "mirnb=$L/libpt_mirnb.so:--workload c1" 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_ab_flags_c1.txt
cat gpurun_out/r05_ab_flags_c1.txt
