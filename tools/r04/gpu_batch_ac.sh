#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04ac
O=gpurun_out/r04ac
export TMPDIR=/tmp
for mode in plain multi; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/$mode -o st -- python3 tools/r04/frames_trace.py $mode 1024 128 30 > $O/$mode.log 2>&1 || exit 6
  echo "== $mode 1024x128"; python tools/r04/frames_summary.py $O/$mode 30 | tee $O/summary_$mode.txt
done
