#!/bin/bash
set -u
L=pathtrace_amd
tools/ab.sh "base=$L/libpathtrace_amd.so:--workload c1" "pf3=$L/libpt_pf3.so:--workload c1" 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_ab_pf3.txt
cat gpurun_out/r05_ab_pf3.txt
