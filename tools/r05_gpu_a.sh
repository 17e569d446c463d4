#!/bin/bash
# round 5 lease: new GPU tests of the scheduler / statistics, then an A/B of the per-lane finished-sample count
set -u
python -m pytest tests/test_gpu_sched.py tests/test_host_mirror.py tests/test_gpu_quadrature.py -m gpu -x -q > gpurun_out/r05_gputests_b.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/r05_gputests_b.log
tools/ab.sh "count=pathtrace_amd/libpathtrace_amd.so" "nocount=pathtrace_amd/libpt_nocount.so" "count_c1=pathtrace_amd/libpathtrace_amd.so:--workload c1" "nocount_c1=pathtrace_amd/libpt_nocount.so:--workload c1" > gpurun_out/r05_ab_count.txt 2>&1
cat gpurun_out/r05_ab_count.txt
