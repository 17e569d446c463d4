#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04c
O=gpurun_out/r04c
PATHTRACE_AMD_LIB=$PWD/pathtrace_amd/libpt_drain.so timeout -k 10 300 python tools/r04/drain_timing.py > $O/drain_timing.txt 2>&1 || { cat $O/drain_timing.txt; exit 5; }
cat $O/drain_timing.txt
timeout -k 10 900 python -m pytest tests/test_gpu_quadrature.py -m gpu -q > $O/quadrature.txt 2>&1
rc=$?; tail -40 $O/quadrature.txt
if [ $rc -gt 1 ]; then exit $rc; fi
