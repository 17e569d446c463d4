#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04v
O=gpurun_out/r04v
timeout -k 10 600 python -m pytest tests/test_gpu_functions.py -m gpu -x -q -k "exchange_memory or tuning_knobs or statistics or level0" > $O/tests.txt 2>&1; rc=$?; tail -3 $O/tests.txt; if [ $rc -ne 0 ]; then exit 9; fi
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 3; }
timeout -k 10 300 python bench.py --in-order --no-cpu-baseline > $O/bench_in_order.json 2> $O/bench_in_order.err || exit 3
timeout -k 10 200 python bench.py --gpus 1 --force-multi --steps 20 --no-cpu-baseline > $O/bench_force_multi.json 2> $O/bench_force_multi.err || exit 8
timeout -k 10 200 python bench.py --gpus 1 --force-dist --steps 20 --no-cpu-baseline > $O/bench_force_dist.json 2> $O/bench_force_dist.err || exit 9
for n in 2 8; do timeout -k 10 300 python bench.py --gpus $n --shared-device --steps 10 --warmup 2 > $O/bench_shared_$n.json 2> $O/bench_shared_$n.err || exit 4; done
for wl in c1 c3 ref; do timeout -k 10 400 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err || exit 5; done
python - <<'PY'
import json
for n in ("bench_default", "bench_in_order", "bench_force_multi", "bench_force_dist", "bench_shared_2", "bench_shared_8", "bench_c1", "bench_c3", "bench_ref"):
    d = json.loads(open(f"gpurun_out/r04v/{n}.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print(n, d["value"], d["ms_per_step"], "launch", r["avg_launch_ms"], "frac", r["frac"], "frac_step", r.get("frac_of_timed_step"), (d.get("cpu_baseline") or {}).get("value"))
PY
