#!/bin/bash
# PMC passes for one config of tools/configs_gpu.py: tools/pmc_cfg.sh <cfg> <tag>
set -e
CFG=$1; OUT=gpurun_out/pmc_$2; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
A="tools/configs_gpu.py $CFG"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d $OUT/p1 -- python3 $A > $OUT/o1.txt
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/p2 -- python3 $A > $OUT/o2.txt
