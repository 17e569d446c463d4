#!/bin/bash
# final validation of round 4 at the final commit: `gpu_final.sh tests` = the whole GPU suite; `gpu_final.sh` = the long fuzz runs, the bench lines, tile scaling
set -o pipefail
mkdir -p gpurun_out/r04final
O=gpurun_out/r04final
if [ "${1:-all}" = tests ]; then
  timeout -k 10 1150 python -m pytest tests/ -m gpu -q > $O/gpu_tests.txt 2>&1
  rc=$?; tail -4 $O/gpu_tests.txt
  exit $rc
fi
PT_FUZZ_SEEDS=3000 timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "matches_f32_oracle and not regenerating" > $O/fuzz_3000.txt 2>&1; tail -1 $O/fuzz_3000.txt
PT_FUZZ_SEEDS=300 timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "regenerating" > $O/fuzz_300_regen.txt 2>&1; tail -1 $O/fuzz_300_regen.txt
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 3
timeout -k 10 400 python bench.py --workload c1 --no-cpu-baseline > $O/bench_c1.json 2> $O/bench_c1.err || exit 4
for wl in c3 ref; do
  timeout -k 10 400 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err || exit 4
done
for form in in-order force-dist force-multi; do
  extra="--$form"; [ $form != in-order ] && extra="--gpus 1 --$form"
  timeout -k 10 300 python bench.py $extra --no-cpu-baseline > $O/bench_${form//-/_}.json 2> $O/bench_${form//-/_}.err || exit 4
done
for n in 2 8; do
  timeout -k 10 300 python bench.py --gpus $n --shared-device --no-cpu-baseline > $O/bench_shared_$n.json 2> $O/bench_shared_$n.err || exit 4
done
timeout -k 10 300 python bench.py --workload c4 --accel 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c4_bvh.json 2> $O/bench_c4_bvh.err || exit 5
python - <<'PY'
import json
for n in ("bench_default", "bench_in_order", "bench_force_dist", "bench_force_multi", "bench_shared_2", "bench_shared_8", "bench_c1", "bench_c3", "bench_ref", "bench_c4_bvh"):
    d = json.loads(open(f"gpurun_out/r04final/{n}.json").read().strip().splitlines()[-1])
    print(n, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], (d.get("cpu_baseline") or {}).get("value"))
PY
timeout -k 10 200 python tools/tile_scaling.py > $O/tile_scaling.txt 2>&1 || exit 6
grep -v amdgpu $O/tile_scaling.txt
