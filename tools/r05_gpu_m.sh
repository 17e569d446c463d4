#!/bin/bash
# round 5 lease: long campaigns at the final kernels -- 3000 random scenes bit-exact against the f32 oracle (queue form; the scan and vertex code all
# forms share), 60 seeds of the random pipelined load over three contexts (every film against the same job run alone and in order)
set -u
PT_FUZZ_SEEDS=3000 timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "matches_f32_oracle and not regenerating" > gpurun_out/r05_fuzz_3000.txt 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/r05_fuzz_3000.txt
for s in $(seq 1 60); do timeout -k 10 120 python tools/dbg/seq_debug.py $s 2>&1 | grep -v amdgpu.ids | tail -1; done > gpurun_out/r05_stress_random_sequences.txt
grep -c "0 bad" gpurun_out/r05_stress_random_sequences.txt; grep -v "0 bad" gpurun_out/r05_stress_random_sequences.txt | head
