"""How the box tests are spread over the rays of the car scene (checker's walk of the traversal tree): camera rays and
random rays from their hit points; mean, median, tail."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import scenes
from oracle import oracle_py
sc = scenes.car_standin(480, 270, detail=1.0)
P = sc.params
w, h = 160, 90
c2w = np.array(P.camera2World[:], np.float64).reshape(4, 4).T
inv = np.array(P.cameraInverseProjection[:], np.float64).reshape(4, 4).T
xs, ys = np.meshgrid((np.arange(w) + 0.5) / w * 2 - 1, (np.arange(h) + 0.5) / h * 2 - 1)
d = (inv @ np.stack([xs.ravel(), ys.ravel(), np.zeros(w * h), np.ones(w * h)]))[:3]
d = (c2w[:3, :3] @ d).T
d /= np.linalg.norm(d, axis=1, keepdims=True)
o = np.tile(c2w[:3, 3], (w * h, 1))
orc = oracle_py.Oracle(sc).build()
o32, d32 = o.astype(np.float32), d.astype(np.float32)
v = np.zeros(len(o32), np.int64)
for i in range(len(o32)):
    v[i] = orc.trace(o32[i:i+1], d32[i:i+1], grid=True)["counters"]["internalVisits"]
print("primary box tests: mean %.1f median %.0f p99 %.0f p99.9 %.0f max %d" % (v.mean(), np.median(v), np.percentile(v, 99), np.percentile(v, 99.9), v.max()))
r = orc.trace(o32, d32, grid=True)
hit = r["triangle"] != 0xFFFFFFFF
p = o[hit] + d[hit] * r["t"][hit, None] - d[hit] * 1e-3
rng = np.random.default_rng(1)
bd = rng.normal(size=(hit.sum(), 3)); bd /= np.linalg.norm(bd, axis=1, keepdims=True)
p32, b32 = p.astype(np.float32), bd.astype(np.float32)
v2 = np.zeros(len(p32), np.int64)
for i in range(len(p32)):
    v2[i] = orc.trace(p32[i:i+1], b32[i:i+1], grid=True)["counters"]["internalVisits"]
print("bounce box tests: mean %.1f median %.0f p99 %.0f p99.9 %.0f max %d" % (v2.mean(), np.median(v2), np.percentile(v2, 99), np.percentile(v2, 99.9), v2.max()))
