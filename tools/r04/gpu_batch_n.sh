#!/bin/bash
# rehearsals of bench.py's N > 1 branches on the one-GPU box
set -o pipefail
mkdir -p gpurun_out/r04n
O=gpurun_out/r04n
for n in 2 8; do
  timeout -k 10 300 python bench.py --gpus $n --shared-device --steps 10 --warmup 2 > $O/bench_shared_$n.json 2> $O/bench_shared_$n.err || { tail -5 $O/bench_shared_$n.err; exit 3; }
  python -c "
import json
d=json.loads(open('$O/bench_shared_$n.json').read().strip().splitlines()[-1]); print('shared $n:', d['value'], d['ms_per_step'], d['n_gpus'], d['roofline']['avg_launch_ms'], d['roofline'].get('launch_times_from','')[:30], d['config']['rccl'], d['config']['launch'][:60])"
done
timeout -k 10 300 bash tools/rehearse_multi.sh 2 > $O/rehearse_gloo2.txt 2>&1 || { tail -5 $O/rehearse_gloo2.txt; exit 4; }
tail -1 $O/rehearse_gloo2.txt | cut -c1-400
