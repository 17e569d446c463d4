#!/bin/bash
# round 5 evidence, part A: rocprofv3 profiles (kernel trace + 7 counter passes each) of the headline, the reference's scene,
# the north_star-literal queue form on C2, and the BVH form of C4
set -u
for spec in "c2:" "c1:--workload c1" "c2_queue:--level0-form 1" "c4_bvh:--workload c4 --accel 1"; do
  tag=${spec%%:*}; args=${spec#*:}
  bash tools/profile_workload.sh $tag $args > gpurun_out/prof_$tag.log 2>&1; echo "$tag rc=$?"
  python3 -c "import json; d=json.load(open('gpurun_out/prof_$tag/roofline_$tag.json')); print('$tag', d['kernel'], d['avg_launch_ms_kernel_trace'], d['avg_launch_ms_in_order_run'], d.get('valu_issue_frac'), d.get('lane_utilisation'), d.get('hbm_bytes_per_launch'), d.get('lds_bank_conflict_frac'))"
done
