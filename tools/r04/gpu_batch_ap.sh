#!/bin/bash
# the --in-order kernel-trace pass of the profiling recipe at 20 steps (tools/profile_workload.sh), and the extended exchange test
set -o pipefail
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_functions.py -m gpu -x -q -k "streamed or multi_gpu_entry" 2>&1 | tail -3 || exit 9
for spec in "c2:" "c1:--workload c1" "c4_bvh:--workload c4 --accel 1"; do
  tag=${spec%%:*}; args=${spec#*:}
  OUT=gpurun_out/prof_$tag; rm -rf $OUT/trace_in_order; mkdir -p $OUT
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_in_order -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --in-order $args > $OUT/bench_trace_in_order.json || exit 5
  find $OUT -name "*agent_info.csv" -delete
  python -c "import json; d=json.loads(open('$OUT/bench_trace_in_order.json').read().strip().splitlines()[-1]); print('$tag', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
