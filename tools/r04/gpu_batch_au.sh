#!/bin/bash
# k_paths_regen_split compiled for 4 / 5 (default) / 6 waves per SIMD under the pipelined bench (C1), same box, three rounds
set -o pipefail
mkdir -p gpurun_out/r04au
O=gpurun_out/r04au
for round in 1 2 3; do
for v in default ws4 ws6; do
  lib=$PWD/pathtrace_amd/libpt_$v.so; [ $v = default ] && lib=$PWD/pathtrace_amd/libpathtrace_amd.so
  PATHTRACE_AMD_LIB=$lib timeout -k 10 300 python bench.py --workload c1 --no-cpu-baseline > $O/b.json 2> $O/b.err || exit 4
  python -c "import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print('$v c1', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])" | tee -a $O/ab.txt
done
done
