#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04j
O=gpurun_out/r04j
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_host_mirror.py tests/test_gpu_bvh.py -m gpu -x -q \
    -k "not full_size and not 4096 and not 1024_spp and not stream_oracle" > $O/tests.txt 2>&1
rc=$?; tail -4 $O/tests.txt
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 600 tools/ab.sh "axis=pathtrace_amd/libpathtrace_amd.so:--workload c1" "noaxis=pathtrace_amd/libpt_noaxis.so:--workload c1" > $O/ab_c1.txt 2>&1 || { tail $O/ab_c1.txt; exit 4; }
cat $O/ab_c1.txt
