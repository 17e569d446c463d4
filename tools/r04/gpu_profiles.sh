#!/bin/bash
# rocprofv3 evidence of round 4: the bench workloads under kernel trace + counter passes (tools/profile_workload.sh)
set -o pipefail
for spec in "c2:" "c1:--workload c1" "c4_bvh:--workload c4 --accel 1"; do
  tag=${spec%%:*}; args=${spec#*:}
  echo "== $tag"
  timeout -k 10 500 bash tools/profile_workload.sh $tag $args > gpurun_out/prof_$tag.log 2>&1 || { tail -5 gpurun_out/prof_$tag.log; exit 5; }
  tail -3 gpurun_out/prof_$tag.log | cut -c1-300
done
