#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04b
O=gpurun_out/r04b
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_functions.py -m gpu -x -q -k "multi or packed or n_device" > $O/tests.txt 2>&1
rc=$?; tail -3 $O/tests.txt
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/share_trace -o st -- python3 tools/r04/share_trace.py 8 20 0 > $O/share_trace.log 2>&1 || exit 6
python tools/r04/trace_gaps.py $O/share_trace > $O/trace_gaps.txt 2>&1; cat $O/trace_gaps.txt
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/share_trace1 -o st -- python3 tools/r04/share_trace.py 1 10 0 > $O/share_trace1.log 2>&1 || exit 7
python tools/r04/trace_gaps.py $O/share_trace1 > $O/trace_gaps1.txt 2>&1; cat $O/trace_gaps1.txt
