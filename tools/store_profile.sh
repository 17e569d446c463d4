#!/bin/bash
# Copy what tools/profile_workload.sh <tag> left under gpurun_out/prof_<tag>/ into profiles/<round>/<tag>/ -- only the files of
# the LAST run of each pass (gpurun merges every call's output into gpurun_out/, so older runs of the same tag pile up there).
#   tools/store_profile.sh <tag> [round dir, default profiles/r04]
set -eu
TAG=${1:?usage: tools/store_profile.sh <tag> [round dir]}
DST=${2:-profiles/r04}
case "$TAG" in */*|.*|"") echo "bad tag: $TAG" >&2; exit 2;; esac
SRC="gpurun_out/prof_$TAG"
[ -d "$SRC" ] || { echo "$SRC does not exist" >&2; exit 2; }
rm -rf "${DST:?}/$TAG"; mkdir -p "$DST/$TAG"
cp "$SRC"/*.json "$DST/$TAG/"
for d in "$SRC"/*/runc; do
  pass=$(basename "$(dirname "$d")"); mkdir -p "$DST/$TAG/$pass/runc"
  newest=$(ls -t "$d"/*kernel_trace.csv | head -1); pid=$(basename "$newest" | cut -d_ -f1)
  cp "$d"/"${pid}"_*.csv "$DST/$TAG/$pass/runc/"
done
cp "$SRC/roofline_$TAG.json" "$DST/roofline_$TAG.json"
echo "stored $DST/$TAG ($(find "$DST/$TAG" -type f | wc -l) files)"
