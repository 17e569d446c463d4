#!/bin/bash
# a fourth buffer set (PT_SETS): launch k + 3 no longer waits for resolve k.  Same box, three rounds.
set -o pipefail
mkdir -p gpurun_out/r04an
O=gpurun_out/r04an
timeout -k 10 900 python -m pytest tests/test_gpu_functions.py tests/test_gpu_properties.py -m gpu -x -q > $O/tests.txt 2>&1
rc=$?; tail -3 $O/tests.txt
if [ $rc -ne 0 ]; then tail -40 $O/tests.txt; exit 9; fi
for round in 1 2 3; do
for v in default sets3; do
  lib=$PWD/pathtrace_amd/libpt_$v.so; [ $v = default ] && lib=$PWD/pathtrace_amd/libpathtrace_amd.so
  for wl in c2 c1; do
    PATHTRACE_AMD_LIB=$lib timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline > $O/b.json 2> $O/b.err || exit 4
    python -c "import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print('$v $wl', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt
  done
  PATHTRACE_AMD_LIB=$lib timeout -k 10 200 python tools/tile_scaling.py 2>&1 | grep "^N=" | cut -c1-40 | tr '\n' ' ' | tee -a $O/ab.txt; echo | tee -a $O/ab.txt
done
done
