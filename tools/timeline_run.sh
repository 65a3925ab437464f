#!/bin/bash
# Kernel timeline of one bench step: tools/timeline_run.sh <out_dir> [bench options...]   (from the repository root on the GPU box)
set -u
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p "$OUT"
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o t -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dropin --no-kernel-timing "$@" > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
CSV=$(find "$OUT/trace" -name '*kernel_trace.csv' | head -1)
python3 tools/timeline.py "$CSV" > "$OUT/timeline.txt"
grep -h '^{"metric"' "$OUT/trace.log" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])" >> "$OUT/timeline.txt"
rm -rf "$OUT/trace"
