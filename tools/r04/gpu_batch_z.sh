#!/bin/bash
# does a high-priority exchange stream get the gather's kernel placed beside three overlapping launches?
set -o pipefail
mkdir -p gpurun_out/r04z
O=gpurun_out/r04z
export TMPDIR=/tmp
line() { python -c "import json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['value'], d['ms_per_step'])"; }
for pr in normal high; do
  PT_XS_PRIORITY=$pr timeout -k 10 300 python bench.py --gpus 1 --force-multi --steps 20 --warmup 3 --no-cpu-baseline > $O/fm_$pr.json 2> $O/fm_$pr.err || exit 4
  line $O/fm_$pr.json "force-multi xs=$pr"
done
export PT_XS_PRIORITY=high
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/fmt -o st -- python3 bench.py --gpus 1 --force-multi --steps 8 --warmup 2 --no-cpu-baseline > $O/fmt.log 2>&1 || exit 6
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r04z/fmt/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
sel = [r for r in rows if any(k in r['Kernel_Name'] for k in ('k_paths', 'k_resolve', 'rccl', 'unpack'))]
t0 = int(sel[0]['Start_Timestamp'])
for r in sel[:44]:
    s = int(r['Start_Timestamp']) - t0; e = int(r['End_Timestamp']) - t0
    print(f"{r['Kernel_Name'].split('(')[0][-30:]:32s} q{r['Queue_Id']:>2s} start {s/1e3:9.1f} end {e/1e3:9.1f} dur {(e-s)/1e3:8.1f}")
PY
