#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04w
O=gpurun_out/r04w
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/fm -o st -- python3 bench.py --gpus 1 --force-multi --steps 8 --warmup 2 --no-cpu-baseline > $O/fm.log 2>&1 || exit 6
python tools/r04/trace_list.py $O/fm 30 40
