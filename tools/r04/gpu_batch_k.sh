#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04k
O=gpurun_out/r04k
timeout -k 10 600 tools/r04/ab_share.sh base=pathtrace_amd/libpathtrace_amd.so prio2=pathtrace_amd/libpt_prio2.so > $O/ab_share.txt 2>&1 || { tail $O/ab_share.txt; exit 4; }
cat $O/ab_share.txt
