#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04ab
O=gpurun_out/r04ab
timeout -k 10 200 python tools/r04/multi_host_times.py 24 0 2>&1 | grep -v amdgpu | tee $O/host_times.txt || exit 4
timeout -k 10 200 python tools/r04/multi_host_times.py 24 1 2>&1 | grep -v amdgpu | tee -a $O/host_times.txt || exit 4
