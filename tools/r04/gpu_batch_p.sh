#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04p
O=gpurun_out/r04p
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_functions.py tests/test_gpu_fuzz.py -m gpu -x -q \
    -k "config2 or level0_forms or regenerating or ragged or tiny_images or statistics or n_device or full_size_exact_mode_is_bit_identical_to_the_f32_oracle and C2" > $O/tests.txt 2>&1
rc=$?; tail -4 $O/tests.txt
if [ $rc -ne 0 ]; then exit 9; fi
timeout -k 10 600 tools/r04/ab_share.sh blk64=pathtrace_amd/libpathtrace_amd.so blk256=pathtrace_amd/libpt_blk256.so > $O/ab_share.txt 2>&1 || { tail $O/ab_share.txt; exit 4; }
cat $O/ab_share.txt
for so in libpathtrace_amd libpt_blk256; do
  PATHTRACE_AMD_LIB=$PWD/pathtrace_amd/$so.so timeout -k 10 300 python bench.py --steps 20 --no-cpu-baseline > $O/bench_$so.json 2> $O/bench_$so.err || exit 5
  python -c "import json; d=json.loads(open('$O/bench_$so.json').read().strip().splitlines()[-1]); print('$so', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done
