#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04r
O=gpurun_out/r04r
timeout -k 10 600 python -m pytest tests/test_gpu_functions.py -m gpu -x -q -k "multi or n_device or packed" > $O/tests.txt 2>&1
rc=$?; tail -3 $O/tests.txt
if [ $rc -ne 0 ]; then exit 9; fi
for n in 2 8; do
  timeout -k 10 300 python bench.py --gpus $n --shared-device --steps 10 --warmup 2 > $O/bench_shared_$n.json 2> $O/bench_shared_$n.err || { tail -5 $O/bench_shared_$n.err; exit 3; }
  python -c "
import json
d=json.loads(open('$O/bench_shared_$n.json').read().strip().splitlines()[-1]); print('shared $n:', d['value'], d['ms_per_step'], d['config']['rccl'])"
done
timeout -k 10 200 python bench.py --gpus 1 --force-multi --steps 20 --no-cpu-baseline > $O/bench_force_multi.json 2> $O/bench_force_multi.err || exit 8
python -c "
import json
d=json.loads(open('$O/bench_force_multi.json').read().strip().splitlines()[-1]); print('force-multi:', d['value'], d['ms_per_step'], d['config']['rccl'], d['config']['tiles'][:120])"
