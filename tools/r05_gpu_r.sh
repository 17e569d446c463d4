#!/bin/bash
set -u
L=pathtrace_amd
PATHTRACE_AMD_LIB=$PWD/$L/libpt_med3.so python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_closed_forms.py -m gpu -x -q > gpurun_out/r05_med3_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r05_med3_tests.log
tools/ab.sh "base=$L/libpathtrace_amd.so:--workload c1" "med3=$L/libpt_med3.so:--workload c1" 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_ab_med3.txt
cat gpurun_out/r05_ab_med3.txt
