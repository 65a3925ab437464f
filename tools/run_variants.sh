#!/bin/bash
# A/B of library variants built with tools/build_variant.sh: for each name, the end-of-launch numbers (0.26 M and 2.07 M camera
# rays, tools/exp_share.py), the bench step and drop-in sequence, and an eighth of the image (tools/rank_share.py).
# Usage (GPU box, repository root): tools/run_variants.sh <name> [<name> ...]      (build_ab/librtcore_<name>.so)
for n in "$@"; do
  echo "== $n"
  export RTCORE_LIB=build_ab/librtcore_$n.so
  timeout -k 10 100 python tools/exp_share.py 8 1 2>&1 | grep "walks=1" | tail -1 | cut -c1-100 || exit 1
  timeout -k 10 100 python tools/exp_share.py 1 1 2>&1 | grep "walks=1" | tail -1 | cut -c1-100 || exit 1
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], 'dropin', d['per_frame_dropin']['ms_per_frame_1spp'])" || exit 1
  timeout -k 10 120 python tools/rank_share.py 8 2>/dev/null | tail -1 || exit 1
done
