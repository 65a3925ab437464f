#!/usr/bin/env python3
"""The per-frame drop-in sequence (RayTracing.cs:118-164: set seed, CSMain, ShadowMap, AddShader) on the
bench scene, a few frames one at a time; run under `rocprofv3 --kernel-trace` and feed the trace to
tools/timeline.py to see where a frame's time goes.  Prints the wall time per frame."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import native, scenes  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sc = scenes.car_standin(1920, 1080)
core = native.RtCore(0)
core.upload(sc); core.bvh_build()
p = sc.params.copy()
core.set_params(p)
for kv in filter(None, os.environ.get("RT_OPTS", "").split(",")):     # rt_set_option pairs, e.g. RT_OPTS=8=128
    k, v = kv.split("="); core.set_option(int(k), int(v))
seeds = scenes.frame_seeds(frames + 2, 1)
core.set_dirty()


def frame(k):
    p.seed = float(seeds[k])
    core.set_params(p)
    core.dispatch()
    core.dispatch_shadowmap()
    core.accumulate(k, True)


frame(0); frame(1)
core.synchronize()
t0 = time.perf_counter()
for k in range(2, frames + 2):
    frame(k)
core.synchronize()
print("ms per frame: %.3f" % ((time.perf_counter() - t0) * 1e3 / frames))
