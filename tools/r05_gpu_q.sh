#!/bin/bash
# round 5 lease: (a) the bench-record test with its new roofline assertions; (b) the LDS-tiled C4 scan with the SLP vectoriser back in the main unit (mainslp)
set -u
python -m pytest tests/test_gpu_sched.py -m gpu -x -q -k "bench_record" > gpurun_out/r05_benchtest.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r05_benchtest.log
L=pathtrace_amd
for round in 1 2; do for v in libpathtrace_amd.so libpt_mainslp.so; do
  r=$(PATHTRACE_AMD_LIB=$PWD/$L/$v python bench.py --no-cpu-baseline --workload c4 --accel 0 --steps 2 --warmup 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['avg_launch_ms'])")
  echo "round $round $v: ms_per_step msamples avg_launch_ms = $r"; done; done 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_ab_tiled_slp.txt
cat gpurun_out/r05_ab_tiled_slp.txt
