#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04aq
timeout -k 10 600 python -m pytest tests/test_gpu_closed_forms.py -m gpu -q > gpurun_out/r04aq/tests.txt 2>&1
rc=$?; tail -60 gpurun_out/r04aq/tests.txt | cut -c1-220; exit $rc
