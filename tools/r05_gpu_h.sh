#!/bin/bash
# round 5 lease: the whole GPU suite with the new default library (no SLP vectoriser, PT_SPLIT_REPLACE, triangle / pair records in stage
# order), then A/B: C1 old records (nb) against new (default); C2 at 7 waves per SIMD (w7)
set -u
python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputests_h.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r05_gputests_h.log
L=pathtrace_amd
tools/ab.sh "nb=$L/libpt_nb.so:--workload c1" "new=$L/libpathtrace_amd.so:--workload c1" "nb_c2=$L/libpt_nb.so" "new_c2=$L/libpathtrace_amd.so" "w7_c2=$L/libpt_w7.so" 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_ab_records.txt
cat gpurun_out/r05_ab_records.txt
