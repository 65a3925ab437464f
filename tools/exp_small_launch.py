#!/usr/bin/env python3
"""How long does a traversal launch of few rays take?  Rays start on a shell around the car and aim
at random points inside it (the kind of incoherent ray a late bounce has).  Prints the kernel time by
ray count, the visits of the longest ray and what a launch of zero rays costs."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import native, scenes  # noqa: E402

sc = scenes.car_standin(1920, 1080)
core = native.RtCore(0)
core.upload(sc); core.bvh_build()
core.enable_kernel_timing(True)
rng = np.random.default_rng(5)
v = np.asarray(sc.vertices["position"], np.float32).reshape(-1, 3)
lo, hi = np.percentile(v, 2, axis=0), np.percentile(v, 98, axis=0)
out = []
for n in (0, 64, 1024, 16384, 131072, 1048576, 8388608):
    m = max(n, 1)
    o = (lo + (hi - lo) * rng.random((m, 3))).astype(np.float32)
    t = (lo + (hi - lo) * rng.random((m, 3))).astype(np.float32)
    d = t - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o, d = o[:n], d[:n]
    best = None
    for rep in range(4):
        core.reset_stats()
        core.trace_rays(o, d)
        ms, calls = core.kernel_time("k_trace")
        best = ms / max(calls, 1) if best is None else min(best, ms / max(calls, 1))
    row = {"rays": n, "kernel_ms": round(best, 4)}
    if n:
        core.enable_counters(True); core.reset_stats(); core.trace_rays(o, d); s = core.stats(); core.enable_counters(False)
        row["mean_visits"] = round(s["internalVisits"] / n / 2, 1)
    out.append(row)
    print(row, flush=True)
print(json.dumps(out))
