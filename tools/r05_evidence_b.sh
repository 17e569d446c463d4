#!/bin/bash
# round 5 evidence, part B: the LDS-tiled linear scan on C4 (profile), the bench lines of the commit, tile scaling, and the
# deliberately broken build that bench.py must refuse
set -u
O=gpurun_out/r05_lines; mkdir -p $O
bash tools/profile_workload.sh c4 --workload c4 --accel 0 > gpurun_out/prof_c4.log 2>&1; echo "c4 rc=$?"
python3 -c "import json; d=json.load(open('gpurun_out/prof_c4/roofline_c4.json')); print('c4', d['kernel'], d['avg_launch_ms_kernel_trace'], d.get('valu_issue_frac'), d.get('lane_utilisation'), d.get('lds_bank_conflict_frac'))"
python bench.py > $O/bench_default.json 2> $O/err.txt; echo "default rc=$?"
python bench.py --no-cpu-baseline --in-order > $O/bench_in_order.json 2>> $O/err.txt; echo "in_order rc=$?"
python bench.py --no-cpu-baseline --workload c1 > $O/bench_c1.json 2>> $O/err.txt; echo "c1 rc=$?"
python bench.py --no-cpu-baseline --workload c3 --steps 3 --warmup 1 > $O/bench_c3.json 2>> $O/err.txt; echo "c3 rc=$?"
python bench.py --no-cpu-baseline --workload ref > $O/bench_ref.json 2>> $O/err.txt; echo "ref rc=$?"
python bench.py --no-cpu-baseline --workload c4 --accel 1 --steps 5 --warmup 1 > $O/bench_c4_bvh.json 2>> $O/err.txt; echo "c4_bvh rc=$?"
python bench.py --no-cpu-baseline --level0-form 1 > $O/bench_c2_queue.json 2>> $O/err.txt; echo "c2_queue rc=$?"
python bench.py --no-cpu-baseline --force-multi > $O/bench_force_multi.json 2>> $O/err.txt; echo "force_multi rc=$?"
python bench.py --no-cpu-baseline --force-dist > $O/bench_force_dist.json 2>> $O/err.txt; echo "force_dist rc=$?"
python bench.py --no-cpu-baseline --gpus 2 --shared-device > $O/bench_shared_2.json 2>> $O/err.txt; echo "shared2 rc=$?"
python bench.py --no-cpu-baseline --gpus 8 --shared-device > $O/bench_shared_8.json 2>> $O/err.txt; echo "shared8 rc=$?"
python tools/tile_scaling.py > $O/tile_scaling.txt 2>> $O/err.txt; cat $O/tile_scaling.txt
# a build whose resolves do not wait for their lane's launch (pt_sched.h: kFaultResolveBeforeLaunch): bench.py must exit non-zero
PATHTRACE_AMD_LIB=$PWD/pathtrace_amd/libpt_broken.so python bench.py --no-cpu-baseline > $O/broken_stdout.txt 2> $O/broken_stderr.txt; echo "broken build: bench.py exit code $?" | tee $O/broken_rc.txt
tail -3 $O/broken_stderr.txt
for f in $O/bench_*.json; do python3 -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); c=d['config']
print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], c.get('frame_equals_single_gpu'), c.get('timed_region_counters_equal_steps_x_per_step'), (c.get('exchange_copy') or {}).get('ms_per_step'))"; done
