#!/bin/bash
# depth of the send-buffer ring (pt_multi.cpp): 2 / 3 / 8 (default) / 16, same box
set -o pipefail
mkdir -p gpurun_out/r04aa
O=gpurun_out/r04aa
for v in default slots2 slots3 slots16 default; do
  lib=$PWD/pathtrace_amd/libpt_$v.so; [ $v = default ] && lib=$PWD/pathtrace_amd/libpathtrace_amd.so
  echo "== $v"
  PATHTRACE_AMD_LIB=$lib timeout -k 10 200 python tools/r04/share_multi.py 48 2>&1 | grep -v amdgpu | tee -a $O/share_multi.txt || exit 4
done
