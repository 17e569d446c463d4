#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04q
O=gpurun_out/r04q
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_functions.py tests/test_gpu_fuzz.py -m gpu -x -q \
    -k "config1 or level0_forms or regenerating or ragged or tiny_images or statistics or full_size_exact_mode_is_bit_identical_to_the_f32_oracle and C1" > $O/tests.txt 2>&1
rc=$?; tail -4 $O/tests.txt
if [ $rc -ne 0 ]; then exit 9; fi
PT_FUZZ_SEEDS=1 PT_FUZZ_DIRECT_SEEDS=60 PT_FUZZ_REGEN_SEEDS=150 timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q > $O/fuzz.txt 2>&1; tail -1 $O/fuzz.txt
timeout -k 10 600 tools/ab.sh "parklds=pathtrace_amd/libpathtrace_amd.so:--workload c1" "parkglobal=pathtrace_amd/libpt_parkglobal.so:--workload c1" > $O/ab_c1.txt 2>&1 || { tail $O/ab_c1.txt; exit 4; }
grep round $O/ab_c1.txt
