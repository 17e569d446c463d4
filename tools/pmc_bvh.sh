#!/bin/bash
# PMC counters of the BVH path kernel on C4 (10 000 spheres, 64 spp); separate passes, kernel trace only
set -e
OUT=gpurun_out/pmc_bvh${1:-}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
A="tools/configs_gpu.py c4b"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d $OUT/p1 -- python3 $A > $OUT/o1.txt
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/p2 -- python3 $A > $OUT/o2.txt
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM --output-format csv -d $OUT/p3 -- python3 $A > $OUT/o3.txt
python3 tools/pmc_sum.py $OUT/p1 $OUT/p2 $OUT/p3 > $OUT/summary.json
cat $OUT/o1.txt
