#!/bin/bash
# thresholds of traverse_segment on C4 (10 000 spheres, 64 spp)
for cfg in "40 10" "48 10" "56 10" "32 10" "24 10" "40 1" "40 4" "40 20" "40 32" "64 10" "1 10"; do
  set -- $cfg
  echo -n "refill $1 leaf $2: "
  TUNE_BVH_REFILL=$1 TUNE_BVH_LEAF=$2 python tools/configs_gpu.py c4b | tail -1
done
