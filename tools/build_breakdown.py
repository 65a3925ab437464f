#!/usr/bin/env python3
"""Per-kernel device time of rt_bvh_build on the triangle soup.  Usage: build_breakdown.py [n_triangles]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import native, scenes  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
sc = scenes.triangle_soup(n, seed=5, width=64, height=64)
core = native.RtCore(0)
core.upload(sc)
core.bvh_build()                      # warm-up (allocations)
core.enable_kernel_timing(True)
reps = 3
core.reset_stats()
tot = 0.0
for _ in range(reps):
    core.bvh_build()
    tot += core.stats()["buildMs"]
out = {"n": n, "build_ms": round(tot / reps, 3), "mtris_per_s": round(n / (tot / reps) / 1e3, 1)}
for k in ("k_calculate", "k_sort_hist", "k_sort_scatter", "k_scan", "k_dedup", "k_tree", "k_fit", "k_rotate", "k_emit_nodes", "k_fine_keys", "k_quantise", "k_pack_leaves", "k_visit_rank"):
    ms, cnt = core.kernel_time(k)
    out[k] = {"ms": round(ms / reps, 3), "launches": cnt // reps}
print(json.dumps(out))
