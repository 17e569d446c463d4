#!/bin/bash
# same-box A/B of library variants on one rank's share of C2 at N = 8 and N = 1 (tools/r04/share_trace.py: 30 renders back to
# back on one stream, HIP events): tools/r04/ab_share.sh <label>=<so> ...   (3 interleaved rounds)
for round in 1 2 3; do
  for spec in "$@"; do
    label=${spec%%=*}; so=${spec#*=}
    a=$(PATHTRACE_AMD_LIB=$PWD/$so python tools/r04/share_trace.py 8 30 0 2>/dev/null | sed 's/.*rows, \([0-9.]*\) ms.*/\1/')
    b=$(PATHTRACE_AMD_LIB=$PWD/$so python tools/r04/share_trace.py 1 12 0 2>/dev/null | sed 's/.*rows, \([0-9.]*\) ms.*/\1/')
    echo "round $round $label: N=8 share $a ms, N=1 $b ms"
  done
done
