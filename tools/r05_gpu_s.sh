#!/bin/bash
# round 5 lease: the C5 job (3840x2160, 1024 spp -- the configuration built for 8 GPUs) whole on ONE GPU, verified like every bench line; small jobs on the
# reference's scene (queue-form kernels) at the final commit
set -u
python bench.py --no-cpu-baseline --workload c5 --steps 3 --warmup 1 > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err; echo "c5 rc=$?"
python3 -c "
import json; d=json.loads(open('gpurun_out/bench_c5.json').read().strip().splitlines()[-1]); c=d['config']
print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], c['frame_equals_single_gpu'], c['timed_region_counters_equal_steps_x_per_step'])"
python tools/small_batches.py 1 -1 1 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_small_batches.txt; cat gpurun_out/r05_small_batches.txt
