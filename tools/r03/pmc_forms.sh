#!/bin/bash
# counter passes of tools/r03/render_form.py for several level0 forms: tools/r03/pmc_forms.sh <scene> <form> [<form> ...]
set -eu
SCENE=$1; shift
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for F in "$@"; do
  OUT=gpurun_out/pmc_form_${SCENE}_$F; rm -rf "$OUT"; mkdir -p "$OUT"
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d $OUT/p1 -- python3 tools/r03/render_form.py $SCENE $F > $OUT/o1.txt
  rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/p2 -- python3 tools/r03/render_form.py $SCENE $F > $OUT/o2.txt
  python3 tools/pmc_sum.py $OUT > $OUT/sum.json
  find $OUT -name "*agent_info.csv" -delete
  echo "== form $F"; cat $OUT/o1.txt | grep scene; python3 - "$OUT/sum.json" <<'P'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d.items():
    if 'resolve' in k: continue
    g=lambda c: v.get(c,{}).get('per_dispatch',0)
    iv=g('SQ_INSTS_VALU'); cyc=g('GRBM_GUI_ACTIVE')
    print(k, 'VALU %.3e SALU %.3e LDS %.3e VMEM_RD %.3e VMEM_WR %.3e lane_util %.3f cycles %.3e issue_frac %.3f' % (iv, g('SQ_INSTS_SALU'), g('SQ_INSTS_LDS'), g('SQ_INSTS_VMEM_RD'), g('SQ_INSTS_VMEM_WR'), g('SQ_THREAD_CYCLES_VALU')/max(64.0*g('SQ_ACTIVE_INST_VALU'),1), cyc, 2*iv/1024/max(cyc,1)))
P
done
