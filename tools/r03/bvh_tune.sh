#!/bin/bash
# thresholds of traverse_segment on C4 (10 000 spheres, 64 spp), 4-wide tree
for cfg in "44 20" "36 20" "52 20" "60 20" "44 12" "44 28" "44 36" "52 28" "36 12" "28 20" "56 32"; do
  set -- $cfg
  echo -n "refill $1 leaf $2: "
  TUNE_BVH_REFILL=$1 TUNE_BVH_LEAF=$2 python tools/configs_gpu.py c4b | tail -1
done
