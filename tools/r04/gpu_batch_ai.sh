#!/bin/bash
# core size of an overlapping launch (kLaneGrid24 / 24 of the device) with spare workgroups: 9 .. 13, C2 and C1, two interleaved rounds
set -o pipefail
mkdir -p gpurun_out/r04ai
O=gpurun_out/r04ai
for round in 1 2; do
for v in default lg9 lg10 lg12 lg13; do
  lib=$PWD/pathtrace_amd/libpt_$v.so; [ $v = default ] && lib=$PWD/pathtrace_amd/libpathtrace_amd.so
  for wl in c2 c1; do
    PATHTRACE_AMD_LIB=$lib timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline > $O/b.json 2> $O/b.err || exit 4
    python -c "import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print('$v $wl', d['value'], d['ms_per_step'])" | tee -a $O/scan.txt
  done
  PATHTRACE_AMD_LIB=$lib timeout -k 10 200 python tools/tile_scaling.py 2>&1 | grep "^N=" | cut -c1-40 | tr '\n' ' ' | tee -a $O/scan.txt; echo | tee -a $O/scan.txt
done
done
