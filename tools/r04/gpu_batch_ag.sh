#!/bin/bash
# spare workgroups: every overlapping launch brings a full device's worth; those beyond the core end at once when successors wait
set -o pipefail
mkdir -p gpurun_out/r04ag
O=gpurun_out/r04ag
timeout -k 10 900 python -m pytest tests/test_gpu_functions.py tests/test_gpu_parity.py tests/test_gpu_properties.py -m gpu -x -q > $O/tests.txt 2>&1
rc=$?; tail -3 $O/tests.txt
if [ $rc -ne 0 ]; then tail -40 $O/tests.txt; exit 9; fi
timeout -k 10 200 python tools/tile_scaling.py > $O/tile_scaling.txt 2>&1 || exit 6
grep -v amdgpu $O/tile_scaling.txt
for wl in c2 c1 c3 ref; do
  timeout -k 10 400 python bench.py --workload $wl --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err || exit 4
  python -c "import json; d=json.loads(open('$O/bench_$wl.json').read().strip().splitlines()[-1]); print('$wl', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
timeout -k 10 300 python bench.py --gpus 1 --force-multi --no-cpu-baseline > $O/fm.json 2> $O/fm.err || exit 4
python -c "import json; d=json.loads(open('$O/fm.json').read().strip().splitlines()[-1]); print('force-multi', d['value'], d['ms_per_step'])"
python - <<'PY'
# isolated renders (synchronised after each) must still take the whole device
import time, torch, pathtrace_amd as pt
ctx = pt.Context(0); ctx.upload(pt.builtin_scene(2)); cam = pt.camera_new(width=1024, height=1024)
for spp in (64, 8):
    prm = pt.default_params(spp=spp)
    ctx.render(cam, prm); t0 = time.perf_counter()
    for _ in range(5): ctx.render(cam, prm)
    print(f"isolated C2 renders, {spp} spp: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms each (wall, incl. allocation of outputs)")
ctx.close()
PY
