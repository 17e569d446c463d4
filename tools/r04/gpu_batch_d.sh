#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04d
O=gpurun_out/r04d
timeout -k 10 900 tools/r04/ab_share.sh base=pathtrace_amd/libpathtrace_amd.so lazy=pathtrace_amd/libpt_lazy.so prio=pathtrace_amd/libpt_prio.so st0=pathtrace_amd/libpt_st0.so lazyst0=pathtrace_amd/libpt_lazyst0.so all3=pathtrace_amd/libpt_all3.so > $O/ab_share.txt 2>&1 || { tail $O/ab_share.txt; exit 4; }
cat $O/ab_share.txt
timeout -k 10 600 python -m pytest tests/test_gpu_quadrature.py -m gpu -q -k ggx > $O/quadrature.txt 2>&1
tail -5 $O/quadrature.txt
