#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04m
O=gpurun_out/r04m
timeout -k 10 400 python bench.py --workload ref --steps 5 --warmup 2 > $O/bench_ref.json 2> $O/bench_ref.err || { tail -5 $O/bench_ref.err; exit 3; }
python -c "
import json
d=json.loads(open('$O/bench_ref.json').read().strip().splitlines()[-1]); print('ref', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['kernel'][:40], d.get('host_buffers'), d['cpu_baseline']['value'], d['cpu_baseline']['sample'][:120])"
timeout -k 10 300 python -m pytest tests/test_host_mirror.py tests/test_gpu_parity.py tests/test_gpu_functions.py -m gpu -x -q -k "reference_job or very_long or statistics_add_up" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
