#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04f
O=gpurun_out/r04f
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t8 -o st -- python3 tools/r04/share_trace.py 8 12 0 > $O/t8.log 2>&1 || exit 6
python tools/r04/trace_list.py $O/t8 10 30
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t1 -o st -- python3 tools/r04/share_trace.py 1 8 0 > $O/t1.log 2>&1 || exit 7
python tools/r04/trace_list.py $O/t1 8 16
