#!/bin/bash
# round 5 lease: (a) does the BVH kernel want the SLP vectoriser back? C4 accel 1, default (no SLP) against -fslp-vectorize;
# (b) extended fuzz campaign over the regenerating forms with the new stack logic (PT_SPLIT_REPLACE): 300 + 90 seeds
set -u
L=pathtrace_amd
for round in 1 2 3; do for v in libpathtrace_amd.so libpt_slp.so; do
  r=$(PATHTRACE_AMD_LIB=$PWD/$L/$v python bench.py --no-cpu-baseline --workload c4 --accel 1 --steps 4 --warmup 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['avg_launch_ms'])")
  echo "round $round $v: ms_per_step msamples avg_launch_ms = $r"; done; done 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_ab_bvh_slp.txt
cat gpurun_out/r05_ab_bvh_slp.txt
PT_FUZZ_REGEN_SEEDS=300 PT_FUZZ_DIRECT_SEEDS=90 timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "regenerating" > gpurun_out/r05_fuzz_regen.log 2>&1; echo "fuzz rc=$?"; tail -3 gpurun_out/r05_fuzz_regen.log
