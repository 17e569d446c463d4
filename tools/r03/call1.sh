#!/bin/bash
# round 3, GPU call 1: the new full-size tests, the bench in its three launch forms, baselines for the perf work
O=gpurun_out/r03; mkdir -p $O
step() { echo "== $*" >&2; timeout -k 10 900 "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed ($rc): stopping" >&2; exit $rc; fi; return 0; }
step python -m pytest tests/test_gpu_fullsize.py -q -s -p no:cacheprovider > $O/fullsize.txt 2>&1
tail -5 $O/fullsize.txt
step python bench.py > $O/bench_default.json 2> $O/bench_default.err
step python bench.py --gpus 1 --force-multi --no-cpu-baseline > $O/bench_force_multi.json 2> $O/bench_force_multi.err
step python bench.py --workload c1 --no-cpu-baseline > $O/bench_c1.json 2>&1
step python bench.py --workload c5 --steps 2 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.err
step tools/ab.sh base=pathtrace_amd/libpathtrace_amd.so ph7=pathtrace_amd/libpt_ph7.so > $O/ab_ph7.txt 2>&1
cat $O/ab_ph7.txt
python - <<'PY'
import json
for n in ("bench_default","bench_force_multi","bench_c1","bench_c5"):
    try:
        d=json.loads(open(f"gpurun_out/r03/{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("all_cores"))
    except Exception as e:
        print(n, "ERR", e)
PY
