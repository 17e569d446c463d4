#!/bin/bash
# the exchange by copies (pt_multi_set_exchange) against the ncclGather form, one device
set -o pipefail
mkdir -p gpurun_out/r04ak
O=gpurun_out/r04ak
timeout -k 10 900 python -m pytest tests/test_gpu_functions.py -m gpu -x -q > $O/tests.txt 2>&1
rc=$?; tail -3 $O/tests.txt
if [ $rc -ne 0 ]; then tail -40 $O/tests.txt; exit 9; fi
line() { python -c "import json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['value'], d['ms_per_step'], (d['config'].get('rccl') or {}).get('exchange'))" | tee -a $O/ab.txt; }
for round in 1 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline > $O/plain.json 2> $O/plain.err || exit 4; line $O/plain.json plain
  timeout -k 10 300 python bench.py --gpus 1 --force-multi --no-cpu-baseline > $O/fm_rccl.json 2> $O/fm_rccl.err || exit 4; line $O/fm_rccl.json force-multi-rccl
  timeout -k 10 300 python bench.py --gpus 1 --force-multi --exchange copy --no-cpu-baseline > $O/fm_copy.json 2> $O/fm_copy.err || exit 4; line $O/fm_copy.json force-multi-copy
  for ex in rccl copy; do timeout -k 10 200 python tools/r04/share_multi.py 48 $ex 2>&1 | grep "^1024" | tee -a $O/ab.txt || exit 4; done
done
