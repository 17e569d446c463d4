#!/bin/bash
# spare workgroups that stayed leave when successors arrive (PT_SPARE_RECHECK): bench C2 / C1, tile scaling, isolated renders; same box A/B
set -o pipefail
mkdir -p gpurun_out/r04aj
O=gpurun_out/r04aj
timeout -k 10 900 python -m pytest tests/test_gpu_functions.py tests/test_gpu_properties.py -m gpu -x -q > $O/tests.txt 2>&1
rc=$?; tail -3 $O/tests.txt
if [ $rc -ne 0 ]; then tail -40 $O/tests.txt; exit 9; fi
for round in 1 2 3; do
for v in default norecheck; do
  lib=$PWD/pathtrace_amd/libpt_$v.so; [ $v = default ] && lib=$PWD/pathtrace_amd/libpathtrace_amd.so
  for wl in c2 c1; do
    PATHTRACE_AMD_LIB=$lib timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline > $O/b.json 2> $O/b.err || exit 4
    python -c "import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print('$v $wl', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])" | tee -a $O/ab.txt
  done
  PATHTRACE_AMD_LIB=$lib timeout -k 10 200 python tools/tile_scaling.py 2>&1 | grep "^N=" | cut -c1-40 | tr '\n' ' ' | tee -a $O/ab.txt; echo | tee -a $O/ab.txt
done
done
for v in default norecheck; do
lib=$PWD/pathtrace_amd/libpt_$v.so; [ $v = default ] && lib=$PWD/pathtrace_amd/libpathtrace_amd.so
PATHTRACE_AMD_LIB=$lib python - <<'PY' | tee -a $O/ab.txt
import time, torch, pathtrace_amd as pt
ctx = pt.Context(0); ctx.upload(pt.builtin_scene(2)); cam = pt.camera_new(width=1024, height=1024)
prm = pt.default_params(spp=64)
lin = torch.empty((1024, 1024, 3), dtype=torch.float32, device="cuda"); rgba = torch.empty((1024, 1024, 4), dtype=torch.uint8, device="cuda")
ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr()); ctx.sync()
t0 = time.perf_counter()
for _ in range(10): ctx.render_into(cam, prm, lin.data_ptr(), rgba.data_ptr()); ctx.sync()
print(f"isolated C2 renders (sync after each): {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms each")
ctx.close()
PY
done
