#!/bin/bash
# round 3, GPU call 2: whole GPU suite on the new f32 specification (plane-form triangle test, sphere-light direction from the
# sampler), then same-box A/B of the variants on C2 and C1
O=gpurun_out/r03; mkdir -p $O
step() { echo "== $*" >&2; timeout -k 10 900 "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed ($rc): stopping" >&2; exit $rc; fi; return 0; }
step python -m pytest tests -q -m gpu -p no:cacheprovider -x > $O/gpu_tests_call2.txt 2>&1
tail -8 $O/gpu_tests_call2.txt
P=pathtrace_amd
step tools/ab.sh base=$P/libpt_base.so new=$P/libpathtrace_amd.so ph7=$P/libpt_ph7.so rsq=$P/libpt_rsq.so > $O/ab2_c2.txt 2>&1
grep round $O/ab2_c2.txt
step tools/ab.sh "base=$P/libpt_base.so:--workload c1" "new=$P/libpathtrace_amd.so:--workload c1" "tribl=$P/libpt_tribl.so:--workload c1" "rsq=$P/libpt_rsq.so:--workload c1" > $O/ab2_c1.txt 2>&1
grep round $O/ab2_c1.txt
