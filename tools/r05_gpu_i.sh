#!/bin/bash
# round 5 lease: LDS latencies off the scans' critical path -- next pair's normal (pf) / normal + v0 (pf2) requested one pair ahead; the last
# (n mod 4) spheres two at a time (rem2); the first run record in scalar registers (run0); combo = pf + rem2 + run0.  C1 and C2.
set -u
L=pathtrace_amd
tools/ab.sh "base=$L/libpathtrace_amd.so:--workload c1" "pf=$L/libpt_pf.so:--workload c1" "pf2=$L/libpt_pf2.so:--workload c1" "run0=$L/libpt_run0.so:--workload c1" "combo=$L/libpt_combo.so:--workload c1" 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_ab_latency_c1.txt
cat gpurun_out/r05_ab_latency_c1.txt
tools/ab.sh "base=$L/libpathtrace_amd.so" "rem2=$L/libpt_rem2.so" "run0=$L/libpt_run0.so" "combo=$L/libpt_combo.so" 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_ab_latency_c2.txt
cat gpurun_out/r05_ab_latency_c2.txt
