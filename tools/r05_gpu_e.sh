#!/bin/bash
# round 5 lease: C1 -- (a) a lane that pushes a Mirror vertex takes a pre-scanned path from the plain stack (PT_SPLIT_REPLACE),
# (b) the pair test's barycentric part without branches (PT_PAIR_FLAT), (c) both; parity subset with (c), then A/B on C1
set -u
PATHTRACE_AMD_LIB=$PWD/pathtrace_amd/libpt_flatrep.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_functions.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r05_flatrep_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r05_flatrep_tests.log
tools/ab.sh "base=pathtrace_amd/libpathtrace_amd.so:--workload c1" "replace=pathtrace_amd/libpt_replace.so:--workload c1" "flat=pathtrace_amd/libpt_flat.so:--workload c1" "flatrep=pathtrace_amd/libpt_flatrep.so:--workload c1" 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_ab_c1_replace_flat.txt
cat gpurun_out/r05_ab_c1_replace_flat.txt
