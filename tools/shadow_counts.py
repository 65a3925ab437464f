#!/usr/bin/env python3
"""Work counters of the ShadowMap pass of the bench scene: rays answered, rays walked, how many of those hit."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import native, scenes
sc = scenes.car_standin(1920, 1080)
core = native.RtCore(0)
core.upload(sc); core.bvh_build(); core.set_params(sc.params)
seeds = scenes.frame_seeds(8, 1)
core.enable_counters(True); core.reset_stats()
core.set_dirty(); core.render_frames(seeds)
st = core.stats()
print(json.dumps({k: int(v) for k, v in st.items() if k.startswith("shadow") or k in ("rays", "raysWalked", "hits", "internalVisits")}))
