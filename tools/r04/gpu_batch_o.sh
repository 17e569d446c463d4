#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04o
O=gpurun_out/r04o
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_functions.py tests/test_gpu_fuzz.py -m gpu -x -q \
    -k "config2 or level0_forms or regenerating or ragged or tiny_images or statistics or full_size_exact_mode_is_bit_identical_to_the_f32_oracle and C2" > $O/tests.txt 2>&1
rc=$?; tail -4 $O/tests.txt
if [ $rc -ne 0 ]; then exit 9; fi
timeout -k 10 600 tools/r04/ab_share.sh mail=pathtrace_amd/libpathtrace_amd.so nomail=pathtrace_amd/libpt_nomail.so > $O/ab_share.txt 2>&1 || { tail $O/ab_share.txt; exit 4; }
cat $O/ab_share.txt
