#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04e
O=gpurun_out/r04e
timeout -k 10 900 python -m pytest tests/test_gpu_functions.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q \
    -k "not full_size and not 4096 and not 1024_spp and not stream_oracle" > $O/tests.txt 2>&1
rc=$?; tail -3 $O/tests.txt
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 600 tools/r04/ab_share.sh lanes=pathtrace_amd/libpathtrace_amd.so nolanes=pathtrace_amd/libpt_nolanes.so > $O/ab_share.txt 2>&1 || { tail $O/ab_share.txt; exit 4; }
cat $O/ab_share.txt
for wl in c2 c1 c3; do
  for so in libpathtrace_amd libpt_nolanes; do
    PATHTRACE_AMD_LIB=$PWD/pathtrace_amd/$so.so timeout -k 10 300 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_${wl}_$so.json 2> $O/bench_${wl}_$so.err || exit 5
    python -c "import json,sys; d=json.loads(open('$O/bench_${wl}_$so.json').read().strip().splitlines()[-1]); print('$wl $so', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
  done
done
timeout -k 10 200 python tools/tile_scaling.py > $O/tile_scaling.txt 2>&1 || exit 6
cat $O/tile_scaling.txt
