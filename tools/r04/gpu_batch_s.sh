#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04s
O=gpurun_out/r04s
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t8 -o st -- python3 tools/r04/share_trace.py 8 16 0 > $O/t8.log 2>&1 || exit 6
python tools/r04/trace_list.py $O/t8 20 26
