#!/bin/bash
# round 5 lease: scan records through the scalar cache with one-step-ahead requests (PT_SCAN_SMEM=1) against the LDS form:
# parity subset with the variant library, then A/B on C1 and C2
set -u
PATHTRACE_AMD_LIB=$PWD/pathtrace_amd/libpt_smem.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_functions.py -m gpu -x -q > gpurun_out/r05_smem_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r05_smem_tests.log
tools/ab.sh "lds_c1=pathtrace_amd/libpathtrace_amd.so:--workload c1" "smem_c1=pathtrace_amd/libpt_smem.so:--workload c1" "lds_c2=pathtrace_amd/libpathtrace_amd.so" "smem_c2=pathtrace_amd/libpt_smem.so" 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_ab_smem.txt
cat gpurun_out/r05_ab_smem.txt
