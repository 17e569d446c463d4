#!/bin/bash
# more seeds of the random pipelined sequence (tools/dbg/seq_debug.py): any film that differs from the job run alone?
set -o pipefail
mkdir -p gpurun_out/r04ar
for s in $(seq 1 30); do
  timeout -k 10 120 python tools/dbg/seq_debug.py $s > gpurun_out/r04ar/seq_$s.txt 2>&1 || { echo "seed $s: run failed"; tail -5 gpurun_out/r04ar/seq_$s.txt; exit 4; }
  echo "seed $s: $(grep -c BAD gpurun_out/r04ar/seq_$s.txt) bad of $(grep -c -E ' ok | BAD ' gpurun_out/r04ar/seq_$s.txt)"
done
