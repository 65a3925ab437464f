#!/usr/bin/env python3
"""Time EVERY rank's share of the bench frame on one GPU, one after the other, and count its rays:
how even is the split, and how far is the slowest share from 1 / world of the whole frame.
Usage: rank_balance.py world [tile size] [spp]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import native, scenes  # noqa: E402

world = int(sys.argv[1])
tile = int(sys.argv[2]) if len(sys.argv) > 2 else (64 if world == 1 else 8)
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
sc = scenes.car_standin(1920, 1080)
core = native.RtCore(0)
core.upload(sc); core.bvh_build()
for kv in filter(None, os.environ.get("RT_OPTS", "").split(",")):
    k, v = kv.split("="); core.set_option(int(k), int(v))
seeds = scenes.frame_seeds(spp, 1)
rows = []
for rank in range(world):
    p = sc.params.copy()
    p.tileRank, p.tileCount, p.tileSize = rank, world, tile
    core.set_params(p)
    core.set_dirty(); core.render_frames(seeds)
    ms = []
    for _ in range(2):
        core.set_dirty(); core.render_frames(seeds)
        ms.append(core.stats()["renderMs"])
    st = core.stats()
    rows.append({"rank": rank, "ms": round(min(ms), 2), "rays": int(st.get("rays", 0))})
    print(rows[-1], flush=True)
worst = max(r["ms"] for r in rows)
print(json.dumps({"world": world, "tile": tile, "spp": spp, "slowest_ms": worst, "mean_ms": round(sum(r["ms"] for r in rows) / world, 2),
                  "ranks": rows}))
