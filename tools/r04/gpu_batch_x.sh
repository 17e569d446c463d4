#!/bin/bash
# reserved compute units (PtTuning.reserved_cus): does the gather's kernel find room beside three overlapping launches?
set -o pipefail
mkdir -p gpurun_out/r04x
O=gpurun_out/r04x
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_functions.py -m gpu -x -q -k "tuning or multi or n_device or rccl or packed or pipelined" > $O/tests.txt 2>&1
rc=$?; tail -3 $O/tests.txt
if [ $rc -ne 0 ]; then exit 9; fi
line() { python -c "import json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['value'], d['ms_per_step'])"; }
for r in -1 1 2 8 16; do
  timeout -k 10 300 python bench.py --gpus 1 --force-multi --reserved-cus $r --steps 20 --warmup 3 --no-cpu-baseline > $O/fm_$r.json 2> $O/fm_$r.err || exit 4
  line $O/fm_$r.json "force-multi reserved=$r"
done
for r in -1 8; do
  timeout -k 10 300 python bench.py --gpus 8 --shared-device --reserved-cus $r --steps 20 --warmup 3 --no-cpu-baseline > $O/sh8_$r.json 2> $O/sh8_$r.err || exit 4
  line $O/sh8_$r.json "shared-device 8 reserved=$r"
done
for r in 0 1 8; do
  echo "== tile_scaling reserved=$r"; timeout -k 10 200 python tools/tile_scaling.py 16 0 $r 2>&1 | grep -v amdgpu || exit 6
done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/fm8 -o st -- python3 bench.py --gpus 1 --force-multi --steps 8 --warmup 2 --no-cpu-baseline > $O/fm8.log 2>&1 || exit 6
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r04x/fm8/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
sel = [r for r in rows if any(k in r['Kernel_Name'] for k in ('k_paths', 'k_resolve', 'rccl', 'unpack'))]
t0 = int(sel[0]['Start_Timestamp'])
for r in sel[:60]:
    s = int(r['Start_Timestamp']) - t0; e = int(r['End_Timestamp']) - t0
    print(f"{r['Kernel_Name'].split('(')[0][-30:]:32s} q{r['Queue_Id']:>2s} start {s/1e3:9.1f} end {e/1e3:9.1f} dur {(e-s)/1e3:8.1f}")
PY
