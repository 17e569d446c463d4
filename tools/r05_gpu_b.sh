#!/bin/bash
# round 5 lease: new GPU tests (scheduler / statistics / bench record), then A/Bs: cost of the finished-sample count (C2, C1),
# the Mirror batch's stay-in-lane survivors (C1), 5 waves for the generic queue-form kernels (small jobs on World::new())
set -u
python -m pytest tests/test_gpu_sched.py tests/test_host_mirror.py tests/test_gpu_quadrature.py tests/test_gpu_functions.py -m gpu -x -q > gpurun_out/r05_gputests_b.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/r05_gputests_b.log
tools/ab.sh "count=pathtrace_amd/libpathtrace_amd.so" "nocount=pathtrace_amd/libpt_nocount.so" 2>&1 | grep -v amdgpu.ids | grep -v "does not verify" > gpurun_out/r05_ab_count.txt
cat gpurun_out/r05_ab_count.txt
tools/ab.sh "stay32=pathtrace_amd/libpathtrace_amd.so:--workload c1" "stay0=pathtrace_amd/libpt_stay0.so:--workload c1" "stay16=pathtrace_amd/libpt_stay16.so:--workload c1" "stay48=pathtrace_amd/libpt_stay48.so:--workload c1" "stay64=pathtrace_amd/libpt_stay64.so:--workload c1" 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_ab_c1_stay.txt
cat gpurun_out/r05_ab_c1_stay.txt
for round in 1 2; do
  for lib in libpathtrace_amd.so libpt_gen5.so; do
    echo "== $lib (round $round)"; PATHTRACE_AMD_LIB=$PWD/pathtrace_amd/$lib python tools/small_batches.py 1 -1 1 2>&1 | grep -v amdgpu.ids
  done
done > gpurun_out/r05_ab_generic_waves.txt
cat gpurun_out/r05_ab_generic_waves.txt
