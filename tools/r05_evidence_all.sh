#!/bin/bash
# round 5, final evidence at one commit on one box: the whole GPU suite, the five profiles (tools/r05_evidence_a.sh + the LDS-tiled C4 scan),
# the bench lines of every mode, tile scaling, the deliberately broken build
set -u
python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputests_final.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r05_gputests_final.log
bash tools/r05_evidence_a.sh
bash tools/r05_evidence_b.sh
