#!/bin/bash
# round 5 lease: the library built as three translation units (main: no SLP; split: no SLP + max-ilp; bvh: SLP) against the one-unit build (one):
# the whole GPU suite with the new build, then A/B on C1, C2 and C4 (BVH)
set -u
python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputests_o.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r05_gputests_o.log
L=pathtrace_amd
tools/ab.sh "one=$L/libpt_one.so:--workload c1" "tus=$L/libpathtrace_amd.so:--workload c1" "one_c2=$L/libpt_one.so" "tus_c2=$L/libpathtrace_amd.so" "one_bvh=$L/libpt_one.so:--workload c4 --accel 1 --steps 4 --warmup 1" "tus_bvh=$L/libpathtrace_amd.so:--workload c4 --accel 1 --steps 4 --warmup 1" 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_ab_tus.txt
cat gpurun_out/r05_ab_tus.txt
