#!/usr/bin/env python3
"""Time one rank's share of the bench frame on one GPU: the tiles rank 0 of `world` would own.
Usage: rank_share.py world [spp]   (TILE: tile size, 8 for a share as in bench.py; RT_OPTS: rt_set_option pairs)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import native, scenes  # noqa: E402

world = int(sys.argv[1])
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
sc = scenes.car_standin(1920, 1080)
core = native.RtCore(0)
core.upload(sc); core.bvh_build()
p = sc.params.copy()
p.tileRank, p.tileCount, p.tileSize = 0, world, int(os.environ.get("TILE", "64" if world == 1 else "8"))
core.set_params(p)
for kv in filter(None, os.environ.get("RT_OPTS", "").split(",")):     # rt_set_option pairs, e.g. RT_OPTS=5=1
    k, v = kv.split("="); core.set_option(int(k), int(v))
seeds = scenes.frame_seeds(spp, 1)
core.render_frames(seeds)
out = []
for _ in range(int(os.environ.get("REPS", "3"))):
    core.set_dirty(); core.render_frames(seeds)
    out.append(round(core.stats()["renderMs"], 2))
print(json.dumps({"world": world, "spp": spp, "ms_per_image_of_this_rank": out}))
