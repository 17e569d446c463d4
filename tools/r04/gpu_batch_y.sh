#!/bin/bash
# the gather on an exchange stream of its own behind a ring of send buffers (pt_multi.cpp)
set -o pipefail
mkdir -p gpurun_out/r04y
O=gpurun_out/r04y
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_functions.py -m gpu -x -q > $O/tests.txt 2>&1
rc=$?; tail -3 $O/tests.txt
if [ $rc -ne 0 ]; then exit 9; fi
line() { python -c "import json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['value'], d['ms_per_step'])"; }
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/plain.json 2> $O/plain.err || exit 4
line $O/plain.json "plain"
timeout -k 10 300 python bench.py --gpus 1 --force-multi --steps 20 --warmup 3 --no-cpu-baseline > $O/fm.json 2> $O/fm.err || exit 4
line $O/fm.json "force-multi"
timeout -k 10 300 python bench.py --gpus 1 --force-dist --steps 20 --warmup 3 --no-cpu-baseline > $O/fd.json 2> $O/fd.err || exit 4
line $O/fd.json "force-dist"
for n in 2 8; do
  timeout -k 10 300 python bench.py --gpus $n --shared-device --steps 20 --warmup 3 --no-cpu-baseline > $O/sh$n.json 2> $O/sh$n.err || exit 4
  line $O/sh$n.json "shared-device $n"
done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/fmt -o st -- python3 bench.py --gpus 1 --force-multi --steps 8 --warmup 2 --no-cpu-baseline > $O/fmt.log 2>&1 || exit 6
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r04y/fmt/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
sel = [r for r in rows if any(k in r['Kernel_Name'] for k in ('k_paths', 'k_resolve', 'rccl', 'unpack'))]
t0 = int(sel[0]['Start_Timestamp'])
for r in sel[:48]:
    s = int(r['Start_Timestamp']) - t0; e = int(r['End_Timestamp']) - t0
    print(f"{r['Kernel_Name'].split('(')[0][-30:]:32s} q{r['Queue_Id']:>2s} start {s/1e3:9.1f} end {e/1e3:9.1f} dur {(e-s)/1e3:8.1f}")
PY
