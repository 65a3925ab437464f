#!/usr/bin/env python3
"""Phase profile of the one-ray-per-lane walker.  Needs a library built with -DTRV_PROFILE:
   hipcc ... -DTRV_PROFILE -shared -o build_ab/lib_prof.so api.hip build.hip render.hip
Usage: walker_profile.py build_ab/lib_prof.so [frames] [ranks: profile rank 0's share of the frame]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RTCORE_LIB"] = os.path.abspath(sys.argv[1])
from raytracing_amd import native, scenes  # noqa: E402

frames = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sc = scenes.car_standin(1920, 1080)
core = native.RtCore(0)
core.upload(sc); core.bvh_build()
params = sc.params.copy()
if len(sys.argv) > 3:
    params.tileRank, params.tileCount, params.tileSize = 0, int(sys.argv[3]), 64
core.set_params(params)
seeds = scenes.frame_seeds(frames, 1)
core.render_frames(seeds)
core.reset_stats()
core.set_dirty(); core.render_frames(seeds)
a, b = (C.c_ulonglong * 16)(), (C.c_ulonglong * 16)()
core.lib.rt_debug_walker_profile(core.ctx, a, b)
names = ["clk_refill", "clk_node", "clk_tri", "clk_retire", "outer_iters", "node_iters", "node_lane_steps", "tri_iters",
         "tri_lane_iters", "unused", "clk_node_fetch", "clk_node_test", "clk_node_next"]
for tag, v in (("path rays", a), ("shadow rays", b)):
    d = dict(zip(names, list(v)[:13]))
    tot = sum(d[k] for k in names[:4]) or 1
    d["share"] = {k: round(d[k] / tot, 3) for k in names[:4]}
    d["clk_per_node_iter"] = round(d["clk_node"] / max(d["node_iters"], 1), 1)
    for k in ("clk_node_fetch", "clk_node_test", "clk_node_next"):
        d[k + "_per_iter"] = round(d[k] / max(d["node_iters"], 1), 1)
    d["clk_per_tri_iter"] = round(d["clk_tri"] / max(d["tri_iters"], 1), 1)
    d["clk_refill_per_outer"] = round(d["clk_refill"] / max(d["outer_iters"], 1), 1)
    d["node_iters_per_outer"] = round(d["node_iters"] / max(d["outer_iters"], 1), 2)
    d["tri_iters_per_outer"] = round(d["tri_iters"] / max(d["outer_iters"], 1), 2)
    d["lanes_per_node_iter"] = round(d["node_lane_steps"] / max(d["node_iters"], 1), 1)
    d["lanes_per_tri_iter"] = round(d["tri_lane_iters"] / max(d["tri_iters"], 1), 1)
    print(tag, json.dumps(d))
