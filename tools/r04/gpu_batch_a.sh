#!/bin/bash
# round 4, first GPU batch: new tests + the suites they touch, bench, tile scaling, fixed-cost trace, host enqueue cost
set -o pipefail
mkdir -p gpurun_out/r04a
O=gpurun_out/r04a
timeout -k 10 900 python -m pytest tests/test_gpu_functions.py tests/test_gpu_parity.py -m gpu -x -q \
    -k "not full_size and not 4096 and not 1024_spp and not stream_oracle" > $O/tests.txt 2>&1
rc=$?
tail -3 $O/tests.txt
if [ $rc -gt 1 ]; then echo "tests ended with $rc: stopping"; exit $rc; fi
timeout -k 10 300 python bench.py --steps 20 > $O/bench_c2.json 2> $O/bench_c2.err || exit 3
tail -c 600 $O/bench_c2.json
timeout -k 10 200 python tools/tile_scaling.py > $O/tile_scaling.txt 2>&1 || exit 4
cat $O/tile_scaling.txt
timeout -k 10 200 python tools/r04/share_trace.py 8 20 0 > $O/share8.txt 2>&1 && timeout -k 10 200 python tools/r04/share_trace.py 8 20 1 >> $O/share8.txt 2>&1 || exit 5
cat $O/share8.txt
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/share_trace -o st -- python3 tools/r04/share_trace.py 8 20 0 > $O/share_trace.log 2>&1 || exit 6
python tools/r04/trace_gaps.py $O/share_trace > $O/trace_gaps.txt 2>&1; cat $O/trace_gaps.txt
timeout -k 10 300 python tools/r04/multi_enqueue.py > $O/multi_enqueue.txt 2>&1 || exit 7
cat $O/multi_enqueue.txt
timeout -k 10 200 python bench.py --gpus 1 --force-multi --steps 20 --no-cpu-baseline > $O/bench_force_multi.json 2> $O/bench_force_multi.err || exit 8
timeout -k 10 200 python bench.py --gpus 1 --force-dist --steps 20 --no-cpu-baseline > $O/bench_force_dist.json 2> $O/bench_force_dist.err || exit 9
python - <<'PY'
import json
for n in ("bench_c2", "bench_force_multi", "bench_force_dist"):
    d = json.loads(open(f"gpurun_out/r04a/{n}.json").read().strip().splitlines()[-1])
    print(n, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["config"].get("rccl"))
PY
