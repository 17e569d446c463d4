#!/bin/bash
# the resolve beside resident path-kernel waves: raised wave priority, 4 / 8 samples' loads in flight.  Same box, three rounds.
set -o pipefail
mkdir -p gpurun_out/r04ao
O=gpurun_out/r04ao
for round in 1 2 3; do
for v in default rprio ru4 ru8; do
  lib=$PWD/pathtrace_amd/libpt_$v.so; [ $v = default ] && lib=$PWD/pathtrace_amd/libpathtrace_amd.so
  for wl in c2 c1; do
    PATHTRACE_AMD_LIB=$lib timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline > $O/b.json 2> $O/b.err || exit 4
    python -c "import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print('$v $wl', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt
  done
  PATHTRACE_AMD_LIB=$lib timeout -k 10 200 python tools/tile_scaling.py 2>&1 | grep "^N=" | cut -c1-40 | tr '\n' ' ' | tee -a $O/ab.txt; echo | tee -a $O/ab.txt
done
done
