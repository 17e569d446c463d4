#!/bin/bash
# thresholds of traverse_segment with postponed leaves on C4 (10 000 spheres, 64 spp)
for cfg in "44 20" "44 32" "44 40" "44 48" "44 56" "52 40" "36 40" "52 56" "60 48" "44 64"; do
  set -- $cfg
  echo -n "refill $1 leaf $2: "
  TUNE_BVH_REFILL=$1 TUNE_BVH_LEAF=$2 python tools/configs_gpu.py c4b | tail -1 | cut -c1-90
done
