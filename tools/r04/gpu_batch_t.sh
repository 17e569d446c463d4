#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04t
O=gpurun_out/r04t
timeout -k 10 600 python -m pytest tests/test_gpu_functions.py tests/test_gpu_parity.py -m gpu -x -q -k "tuning_knobs or statistics or level0_forms or config1 or config2 or ragged" > $O/tests.txt 2>&1
rc=$?; tail -3 $O/tests.txt
if [ $rc -ne 0 ]; then exit 9; fi
for n in 8 4 2 1; do
  echo "== three lanes"; timeout -k 5 300 python tools/r04/share_grid.py $n 0,1536,1024,896,768,640,512 2>&1 | tail -7
done
echo "== two lanes, N = 8"; PATHTRACE_AMD_LIB=$PWD/pathtrace_amd/libpt_lanes2.so timeout -k 5 300 python tools/r04/share_grid.py 8 0,768,640 2>&1 | tail -3
