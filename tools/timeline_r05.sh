#!/bin/bash
# Device timelines (rocprofv3 --kernel-trace --memory-copy-trace) of one ffx_score call at the headline size with page-locked
# and with ordinary buffers: tools/timeline_r05.sh <out-dir>.  Only the summaries are kept.
set -eu
OUT=${1:?out dir}
ROOT=$(pwd)
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
export TMPDIR=/tmp
cd /tmp
for m in pinned ordinary; do
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/trace_$m" -o t -- python3 "$ROOT/tools/score_timeline.py" $m > "$OUT/$m.json" 2> "$OUT/$m.err"
  python3 "$ROOT/tools/score_timeline.py" --read "$OUT/trace_$m" > "$OUT/${m}_timeline.json"
  find "$OUT/trace_$m" -name "*.csv" -delete
done
