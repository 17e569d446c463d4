#!/bin/bash
# full GPU suite + bench lines at the current commit
set -o pipefail
mkdir -p gpurun_out/r04i
O=gpurun_out/r04i
timeout -k 10 1100 python -m pytest tests/ -m gpu -x -q > $O/tests.txt 2>&1
rc=$?; tail -5 $O/tests.txt
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 20 > $O/bench_c2.json 2> $O/bench_c2.err || exit 3
timeout -k 10 200 python bench.py --gpus 1 --force-multi --steps 20 --no-cpu-baseline > $O/bench_force_multi.json 2> $O/bench_force_multi.err || exit 8
timeout -k 10 200 python bench.py --gpus 1 --force-dist --steps 20 --no-cpu-baseline > $O/bench_force_dist.json 2> $O/bench_force_dist.err || exit 9
python - <<'PY'
import json
for n in ("bench_c2", "bench_force_multi", "bench_force_dist"):
    d = json.loads(open(f"gpurun_out/r04i/{n}.json").read().strip().splitlines()[-1])
    print(n, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["config"].get("rccl"))
PY
timeout -k 10 200 python tools/tile_scaling.py > $O/tile_scaling.txt 2>&1 || exit 6
cat $O/tile_scaling.txt
