#!/bin/bash
# round 5 lease: (a) 60 seeds of the random pipelined load over three contexts (every film against the same job run alone and in order),
# (b) scheduling options once more, on the FINAL kernels: max-ilp, if-conversion threshold 1, both -- C1 and C2
set -u
for s in $(seq 1 60); do
  timeout -k 10 120 python tools/dbg/seq_debug.py $s > gpurun_out/seq_$s.txt 2>&1
  echo "seed $s: $(grep -c ' BAD ' gpurun_out/seq_$s.txt) bad of $(grep -c -E '^[0-9]+ (ok |BAD)' gpurun_out/seq_$s.txt)"; rm -f gpurun_out/seq_$s.txt
done > gpurun_out/r05_stress_random_sequences.txt
tail -3 gpurun_out/r05_stress_random_sequences.txt; grep -c ": 0 bad of 28" gpurun_out/r05_stress_random_sequences.txt
L=pathtrace_amd
tools/ab.sh "base=$L/libpathtrace_amd.so:--workload c1" "maxilp=$L/libpt_maxilp.so:--workload c1" "phi1=$L/libpt_phi1.so:--workload c1" "mp=$L/libpt_mp.so:--workload c1" "base_c2=$L/libpathtrace_amd.so" "maxilp_c2=$L/libpt_maxilp.so" "phi1_c2=$L/libpt_phi1.so" "mp_c2=$L/libpt_mp.so" 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_ab_flags2.txt
cat gpurun_out/r05_ab_flags2.txt
