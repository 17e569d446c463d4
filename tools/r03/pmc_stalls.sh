#!/bin/bash
# stall / activity counters of one level-0 form: tools/r03/pmc_stalls.sh <scene> <form>
set -eu
SCENE=$1; F=$2
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
OUT=gpurun_out/pmc_stalls_${SCENE}_$F; rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/p1 -- python3 tools/r03/render_form.py $SCENE $F > $OUT/o1.txt
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p2 -- python3 tools/r03/render_form.py $SCENE $F > $OUT/o2.txt
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VSKIPPED SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_CYCLES --output-format csv -d $OUT/p3 -- python3 tools/r03/render_form.py $SCENE $F > $OUT/o3.txt
python3 tools/pmc_sum.py $OUT > $OUT/sum.json
find $OUT -name "*agent_info.csv" -delete
python3 - "$OUT/sum.json" <<'P'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d.items():
    if 'resolve' in k: continue
    print(k)
    for c,x in v.items(): print('   %-26s %.4e'%(c,x['per_dispatch']))
P
