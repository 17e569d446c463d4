#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04ah
for i in 1 2 3; do timeout -k 10 200 python tools/tile_scaling.py 2>&1 | grep "^N=" | cut -c1-40 || exit 6; done | tee gpurun_out/r04ah/ts.txt
