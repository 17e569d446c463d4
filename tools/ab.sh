#!/bin/bash
# A/B of library variants: tools/ab.sh "<label>=<so>[:<bench args>]" ...   (3 interleaved rounds each)
for round in 1 2 3; do
  for spec in "$@"; do
    label=${spec%%=*}; rest=${spec#*=}; so=${rest%%:*}; args=""; [[ "$rest" == *:* ]] && args=${rest#*:}
    v=$(PATHTRACE_AMD_LIB=$PWD/$so python bench.py --no-cpu-baseline --steps 8 --warmup 2 $args | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['avg_launch_ms'])")
    echo "round $round $label: ms_per_step msamples avg_launch_ms = $v"
  done
done
