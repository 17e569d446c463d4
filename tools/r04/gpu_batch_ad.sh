#!/bin/bash
# 512-thread workgroups for k_paths_regen: one free workgroup slot of a CU then holds RCCL's kernel (4 x 136 VGPRs)
set -o pipefail
mkdir -p gpurun_out/r04ad
O=gpurun_out/r04ad
export TMPDIR=/tmp
for v in default blk512 default blk512; do
  lib=$PWD/pathtrace_amd/libpt_$v.so; [ $v = default ] && lib=$PWD/pathtrace_amd/libpathtrace_amd.so
  echo "== $v" | tee -a $O/share_multi.txt
  PATHTRACE_AMD_LIB=$lib timeout -k 10 200 python tools/r04/share_multi.py 48 2>&1 | grep "^1024" | tee -a $O/share_multi.txt || exit 4
  PATHTRACE_AMD_LIB=$lib timeout -k 10 200 python tools/tile_scaling.py 2>&1 | grep "^N=" | tee -a $O/share_multi.txt || exit 4
done
export PATHTRACE_AMD_LIB=$PWD/pathtrace_amd/libpt_blk512.so
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/multi -o st -- python3 tools/r04/frames_trace.py multi 1024 128 30 > $O/multi.log 2>&1 || exit 6
python tools/r04/frames_summary.py $O/multi 30 | head -8
