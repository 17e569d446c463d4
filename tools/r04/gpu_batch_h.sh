#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04h
O=gpurun_out/r04h
export TMPDIR=/tmp
PATHTRACE_AMD_LIB=$PWD/pathtrace_amd/libpt_drain.so timeout -k 10 100 python tools/r04/lane_ramp.py 8 8 > $O/ramp8.txt 2>&1 || exit 3
cat $O/ramp8.txt
PATHTRACE_AMD_LIB=$PWD/pathtrace_amd/libpt_drain.so timeout -k 10 100 python tools/r04/lane_ramp.py 1 6 > $O/ramp1.txt 2>&1 || exit 3
cat $O/ramp1.txt
timeout -k 10 600 tools/r04/ab_share.sh lanes=pathtrace_amd/libpathtrace_amd.so nolanes=pathtrace_amd/libpt_nolanes.so > $O/ab_share.txt 2>&1 || { tail $O/ab_share.txt; exit 4; }
cat $O/ab_share.txt
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t8 -o st -- python3 tools/r04/share_trace.py 8 12 0 > $O/t8.log 2>&1 || exit 6
python tools/r04/trace_list.py $O/t8 18 16
