#!/bin/bash
# round 5 lease: the whole GPU suite, then A/Bs on one box: workgroup-level statistics totals and the finished-sample count (C2, C1)
set -u
python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputests_c.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r05_gputests_c.log
tools/ab.sh "default=pathtrace_amd/libpathtrace_amd.so" "nowg=pathtrace_amd/libpt_nowg.so" "nocount=pathtrace_amd/libpt_nocount.so" 2>&1 | grep -v amdgpu.ids | grep -v "does not verify" > gpurun_out/r05_ab_wg_totals.txt
cat gpurun_out/r05_ab_wg_totals.txt
tools/ab.sh "default=pathtrace_amd/libpathtrace_amd.so:--workload c1" "nowg=pathtrace_amd/libpt_nowg.so:--workload c1" "stay32=pathtrace_amd/libpt_stay32.so:--workload c1" 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_ab_wg_totals_c1.txt
cat gpurun_out/r05_ab_wg_totals_c1.txt
# stack traffic of the stay-in-lane variant against the default (one counter per pass)
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for lib in libpathtrace_amd.so libpt_stay32.so; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    PATHTRACE_AMD_LIB=$PWD/pathtrace_amd/$lib rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/prof_stay/$lib/$ctr -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --in-order --workload c1 > /dev/null 2>&1
    python3 - <<PY
import csv, glob
tot = n = 0
for f in glob.glob("gpurun_out/prof_stay/$lib/$ctr/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "regen_split" in r["Kernel_Name"] and r["Counter_Name"] == "$ctr":
            tot += float(r["Counter_Value"]); n += 1
print("$lib $ctr per launch (KB, raw):", tot / max(n, 1), "over", n, "dispatches")
PY
  done
done > gpurun_out/r05_stay_traffic.txt 2>&1
cat gpurun_out/r05_stay_traffic.txt
