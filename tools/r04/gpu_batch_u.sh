#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04u
O=gpurun_out/r04u
timeout -k 10 600 python -m pytest tests/test_gpu_functions.py tests/test_gpu_parity.py tests/test_gpu_properties.py -m gpu -x -q -k "tuning_knobs or statistics or level0_forms or config1 or config2 or ragged or schedule_invariance or spp_linearity or n_device" > $O/tests.txt 2>&1
rc=$?; tail -3 $O/tests.txt
if [ $rc -ne 0 ]; then exit 9; fi
timeout -k 10 200 python tools/tile_scaling.py > $O/tile_scaling.txt 2>&1 || exit 6
grep -v amdgpu $O/tile_scaling.txt
echo "== four lanes"; PATHTRACE_AMD_LIB=$PWD/pathtrace_amd/libpt_lanes4.so timeout -k 10 200 python tools/tile_scaling.py 2>&1 | grep "^N="
for wl in c2 c1 c3 ref; do
  timeout -k 10 400 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err || exit 4
  python -c "import json; d=json.loads(open('$O/bench_$wl.json').read().strip().splitlines()[-1]); print('$wl', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
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
