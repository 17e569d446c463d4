#!/bin/bash
# smaller cores leave free workgroup slots in pairs on some CUs: does RCCL's gather kernel (4 x 136 VGPRs = two slots of one CU) then get placed
# in the windows between launches?  share_multi (plain vs pt_multi over RCCL), cores of 8 / 9 / 10 / 11 (default) 24ths
set -o pipefail
mkdir -p gpurun_out/r04as
O=gpurun_out/r04as
for round in 1 2; do
for v in default lg10 lg9 lg8; do
  lib=$PWD/pathtrace_amd/libpt_$v.so; [ $v = default ] && lib=$PWD/pathtrace_amd/libpathtrace_amd.so
  echo "== $v" | tee -a $O/ab.txt
  PATHTRACE_AMD_LIB=$lib timeout -k 10 200 python tools/r04/share_multi.py 48 rccl 2>&1 | grep "^1024" | tee -a $O/ab.txt || exit 4
done
done
