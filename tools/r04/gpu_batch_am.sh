#!/bin/bash
# do launches of a sequence run in pairs after the ramp?  trace of 24 whole-image frames; cores of launches 2 and 3 after idle varied (PT_RAMP)
set -o pipefail
mkdir -p gpurun_out/r04am
O=gpurun_out/r04am
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/plain -o st -- python3 tools/r04/frames_trace.py plain 1024 1024 24 > $O/plain.log 2>&1 || exit 6
python tools/r04/frames_summary.py $O/plain 24 > $O/summary_plain.txt; cat $O/summary_plain.txt
line() { python -c "import json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt; }
for round in 1 2; do
for r in "0,0" "16,6" "6,16" "11,5" "5,11" "14,8"; do
  PT_RAMP=$r timeout -k 10 300 python bench.py --no-cpu-baseline > $O/a.json 2> $O/a.err || exit 4; line $O/a.json "ramp=$r c2"
done
done
