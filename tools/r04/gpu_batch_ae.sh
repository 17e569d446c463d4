#!/bin/bash
# is it the scratch of RCCL's kernel (352 B per lane -> "use-once" scratch above HSA_SCRATCH_SINGLE_LIMIT) that makes it wait?
set -o pipefail
mkdir -p gpurun_out/r04ae
O=gpurun_out/r04ae
run() { echo "== $1" | tee -a $O/share_multi.txt; timeout -k 10 200 python tools/r04/share_multi.py 48 2>&1 | grep "^1024" | tee -a $O/share_multi.txt; }
run "default environment" || exit 4
HSA_SCRATCH_SINGLE_LIMIT=1073741824 run "HSA_SCRATCH_SINGLE_LIMIT=1 GiB" || exit 4
HSA_SCRATCH_SINGLE_LIMIT_ASYNC=0 run "HSA_SCRATCH_SINGLE_LIMIT_ASYNC=0" || exit 4
HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0 run "HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0" || exit 4
HSA_ENABLE_SCRATCH_ALT=0 run "HSA_ENABLE_SCRATCH_ALT=0" || exit 4
run "default environment" || exit 4
