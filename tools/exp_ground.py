"""Experiment: how many of the box tests per ray are due to the two huge ground triangles sitting deep in the tree?"""
import json, sys
import numpy as np
sys.path.insert(0, ".")
from raytracing_amd import native, scenes, wire  # noqa: E402

core = native.RtCore(0)
full = scenes.car_standin(480, 270)
P = full.params
w, h = 480, 270
c2w = np.array(P.camera2World[:], np.float64).reshape(4, 4).T
inv = np.array(P.cameraInverseProjection[:], np.float64).reshape(4, 4).T
xs, ys = np.meshgrid((np.arange(w) + 0.5) / w * 2 - 1, (np.arange(h) + 0.5) / h * 2 - 1)
d = (inv @ np.stack([xs.ravel(), ys.ravel(), np.zeros(w * h), np.ones(w * h)]))[:3]
d = (c2w[:3, :3] @ d).T
d /= np.linalg.norm(d, axis=1, keepdims=True)
o = np.tile(c2w[:3, 3], (w * h, 1))
core.upload(full); core.bvh_build()
r = core.trace_rays(o.astype(np.float32), d.astype(np.float32))
hit = r["triangle"] != 0xFFFFFFFF
p = o[hit] + d[hit] * r["t"][hit, None] - d[hit] * 1e-3
L = -np.array(P.lightParameter[:3], np.float64)
rng = np.random.default_rng(1)
sd = L + rng.normal(size=(hit.sum(), 3)) * 0.08
sd /= np.linalg.norm(sd, axis=1, keepdims=True)
on_ground = r["triangle"][hit] < 2
out = []
for name in ("full", "no_ground"):
    sc = scenes.car_standin(480, 270)
    if name == "no_ground":
        sc.indices = sc.indices[6:]; sc.material_index = sc.material_index[2:]; sc.shadow_index = sc.shadow_index[2:]
    core.upload(sc); core.bvh_build()
    core.enable_counters(True)
    res = {}
    for label, sel in (("from_ground", on_ground), ("from_car", ~on_ground)):
        core.reset_stats()
        core.trace_rays(p[sel].astype(np.float32), sd[sel].astype(np.float32))
        st = core.stats()
        res[label] = {"rays": st["rays"], "boxes_per_ray": round(st["internalVisits"] / st["rays"], 1), "tris_per_ray": round(st["triTests"] / st["rays"], 2),
                      "hit_frac": round(st["hits"] / st["rays"], 3)}
    core.enable_counters(False)
    out.append({name: res})
print(json.dumps(out, indent=1))
