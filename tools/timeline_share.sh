#!/bin/bash
# Kernel timeline of one rank's share of the bench frame: tools/timeline_share.sh <out_dir> <world> [back]   (GPU box, repository root)
# back: start at the k_generate launch that is the back-th from the end (2 where the batch goes down two frame lanes)
set -u
OUT=$1; W=$2
export TMPDIR=/tmp REPS=1
mkdir -p "$OUT"
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o t -- python3 tools/rank_share.py "$W" > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
CSV=$(find "$OUT/trace" -name '*kernel_trace.csv' | head -1)
python3 tools/timeline.py "$CSV" ${3:-1} > "$OUT/timeline.txt"
tail -1 "$OUT/trace.log" >> "$OUT/timeline.txt"
rm -rf "$OUT/trace"
