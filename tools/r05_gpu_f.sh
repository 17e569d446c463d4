#!/bin/bash
# round 5 lease: kernels compiled without the SLP vectoriser (it packs f32 pairs into v_pk_* -- no throughput gain on gfx950, register
# pairs and copies), alone and with the flat sphere test (C2) / the replacement (C1)
set -u
tools/ab.sh "base=pathtrace_amd/libpathtrace_amd.so" "noslp=pathtrace_amd/libpt_noslp.so" "noslpsflat=pathtrace_amd/libpt_noslpsflat.so" 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_ab_noslp_c2.txt
cat gpurun_out/r05_ab_noslp_c2.txt
tools/ab.sh "base=pathtrace_amd/libpathtrace_amd.so:--workload c1" "replace=pathtrace_amd/libpt_replace.so:--workload c1" "noslp=pathtrace_amd/libpt_noslp.so:--workload c1" "noslprep=pathtrace_amd/libpt_noslprep.so:--workload c1" 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_ab_noslp_c1.txt
cat gpurun_out/r05_ab_noslp_c1.txt
