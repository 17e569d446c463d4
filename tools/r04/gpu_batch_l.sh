#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04l
O=gpurun_out/r04l
PATHTRACE_AMD_LIB=$PWD/pathtrace_amd/libpt_sort1.so timeout -k 10 900 python -m pytest tests/test_gpu_bvh.py -m gpu -x -q -k "bit_identical or hand_off" > $O/tests.txt 2>&1
rc=$?; tail -4 $O/tests.txt
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 900 tools/ab.sh "sort1=pathtrace_amd/libpt_sort1.so:--workload c4 --accel 1 --steps 3 --warmup 1" "nosort=pathtrace_amd/libpt_nosort.so:--workload c4 --accel 1 --steps 3 --warmup 1" > $O/ab_c4.txt 2>&1 || { tail $O/ab_c4.txt; exit 4; }
grep round $O/ab_c4.txt
