#!/bin/bash
set -e
OUT=gpurun_out/pmc_c4; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
A="tools/configs_gpu.py c4m"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d $OUT/p1 -- python3 $A > $OUT/o1.txt
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/p2 -- python3 $A > $OUT/o2.txt
rocprofv3 --kernel-trace --pmc SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 --output-format csv -d $OUT/p3 -- python3 $A > $OUT/o3.txt
