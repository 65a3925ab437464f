"""Experiment: how much of the walker's work is due to the coarse Morton grid of the reference's
build (10 bits per axis over a fixed [-125, 125]^3 world)?  The same scene and rays, scaled up so
the scene fills that world, get finer cells and therefore a better tree; box tests per ray are
counted for both."""
import json
import sys

import numpy as np

sys.path.insert(0, ".")
from raytracing_amd import native, scenes, wire  # noqa: E402

sc = scenes.car_standin(480, 270)
P = sc.params
core = native.RtCore(0)


def run(k):
    s2 = scenes.car_standin(480, 270)
    s2.vertices["position"] *= np.float32(k)
    core.upload(s2)
    core.bvh_build()
    # primary rays through pixel centres (pinhole)
    w, h = 480, 270
    c2w = np.array(P.camera2World[:], np.float64).reshape(4, 4).T
    inv = np.array(P.cameraInverseProjection[:], np.float64).reshape(4, 4).T
    xs, ys = np.meshgrid((np.arange(w) + 0.5) / w * 2 - 1, (np.arange(h) + 0.5) / h * 2 - 1)
    d = (inv @ np.stack([xs.ravel(), ys.ravel(), np.zeros(w * h), np.ones(w * h)]))[:3]
    d = (c2w[:3, :3] @ d).T
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = np.tile(c2w[:3, 3] * k, (w * h, 1))
    core.enable_counters(True)
    core.reset_stats()
    r = core.trace_rays(o.astype(np.float32), d.astype(np.float32))
    st1 = core.stats()
    hit = r["triangle"] != 0xFFFFFFFF
    p = o[hit] + d[hit] * r["t"][hit, None] - d[hit] * 1e-3 * k
    L = -np.array(P.lightParameter[:3], np.float64)
    rng = np.random.default_rng(1)
    sd = L + rng.normal(size=(hit.sum(), 3)) * 0.08
    sd /= np.linalg.norm(sd, axis=1, keepdims=True)
    core.reset_stats()
    r2 = core.trace_rays(p.astype(np.float32), sd.astype(np.float32))
    st2 = core.stats()
    core.enable_counters(False)
    f = lambda st: {"rays": st["rays"], "boxes_per_ray": round(st["internalVisits"] / st["rays"], 1), "leaves_per_ray": round(st["leafVisits"] / st["rays"], 2),
                    "tris_per_ray": round(st["triTests"] / st["rays"], 2), "hit_frac": round(st["hits"] / st["rays"], 3)}
    return {"scale": k, "primary": f(st1), "shadow": f(st2)}


print(json.dumps([run(1.0), run(4.0), run(8.0)], indent=1))
core.close()
