#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04g
O=gpurun_out/r04g
export TMPDIR=/tmp
timeout -k 10 600 tools/r04/ab_share.sh lanelow=pathtrace_amd/libpathtrace_amd.so lanehi=pathtrace_amd/libpt_lanehi.so lanedef=pathtrace_amd/libpt_lanedef.so nolanes=pathtrace_amd/libpt_nolanes.so > $O/ab_share.txt 2>&1 || { tail $O/ab_share.txt; exit 4; }
cat $O/ab_share.txt
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t8 -o st -- python3 tools/r04/share_trace.py 8 12 0 > $O/t8.log 2>&1 || exit 6
python tools/r04/trace_list.py $O/t8 18 24
