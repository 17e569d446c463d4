#!/bin/bash
# rocprofv3 evidence for one bench.py workload on the GPU box: kernel trace + stats, then one --pmc pass per run
# (never combined with trace domains other than --kernel-trace).  Output: gpurun_out/prof_<tag>/ and the summary
# gpurun_out/prof_<tag>/roofline_<tag>.json (tools/roofline_from_profile.py), to be copied into profiles/.
#   tools/profile_workload.sh <tag> [bench.py args, e.g. --workload c1 | --workload c4 --accel 1]
set -eu
TAG=${1:?usage: tools/profile_workload.sh <tag> [bench.py args]}; shift
case "$TAG" in */*|.*|"") echo "bad tag: $TAG" >&2; exit 2;; esac
OUT="gpurun_out/prof_$TAG"
rm -rf "${OUT:?}"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
ARGS="bench.py --steps 5 --warmup 1 --no-cpu-baseline $*"
python3 $ARGS > $OUT/bench_plain.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/bench_trace.json
# the same command with --in-order under the kernel trace: every launch on its own, so that the AVERAGE of the dominant kernel in
# the --stats summary is a launch time too (in the default command's summary the timed steps' dispatches overlap and last 2-3 periods)
# (20 steps like the driver's run: the first launches of a process run at clocks that are still ramping up)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_in_order -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --in-order $* > $OUT/bench_trace_in_order.json
# counter passes: every launch in order and full size (--in-order), like the launches bench.py takes its launch time from -- a counter
# pass serialises the kernels anyway, and an overlapping launch run alone would work with its core workgroups only (pt_api.cpp, lanes)
ARGS="$ARGS --in-order"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/bench_pmc_fetch.json
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/bench_pmc_write.json
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/bench_pmc_sq.json
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- python3 $ARGS > $OUT/bench_pmc_sq2.json
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_FLOPS_FP32_TRANS SQ_INSTS_VALU_IOPS --output-format csv -d $OUT/pmc_sq3 -- python3 $ARGS > $OUT/bench_pmc_sq3.json
# LDS: extra cycles lost to bank conflicts against all LDS-array cycles (BASELINE.md 3 asks for it on C4; MI355X_MICROARCH.md: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_lds -- python3 $ARGS > $OUT/bench_pmc_lds.json
python3 tools/roofline_from_profile.py $OUT $TAG > $OUT/roofline_$TAG.json
cat $OUT/roofline_$TAG.json
# keep what is judged small: the stats / trace / counter CSVs, not the per-agent metadata
find $OUT -name "*agent_info.csv" -delete
