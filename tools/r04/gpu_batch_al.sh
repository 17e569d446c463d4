#!/bin/bash
# does the stream the caller renders on matter?  bench.py on torch's default (legacy null) stream vs the context's own stream
set -o pipefail
mkdir -p gpurun_out/r04al
O=gpurun_out/r04al
line() { python -c "import json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt; }
for round in 1 2 3; do
  timeout -k 10 300 python bench.py --no-cpu-baseline > $O/a.json 2> $O/a.err || exit 4; line $O/a.json "null-stream"
  PT_BENCH_OWN_STREAM=1 timeout -k 10 300 python bench.py --no-cpu-baseline > $O/b.json 2> $O/b.err || exit 4; line $O/b.json "own-stream"
  PT_BENCH_OWN_STREAM=1 timeout -k 10 300 python bench.py --steps 40 --no-cpu-baseline > $O/c.json 2> $O/c.err || exit 4; line $O/c.json "own-stream-40-steps"
  timeout -k 10 300 python bench.py --steps 40 --no-cpu-baseline > $O/d.json 2> $O/d.err || exit 4; line $O/d.json "null-stream-40-steps"
done
