"""Box tests per ray of the checker's walk of the traversal tree (the library's counts are the same, tests/test_gpu_parity.py)
for three kinds of ray on the car scene: camera rays, shadow-like rays from their hit points towards the light, random
rays from the hit points.  Used to compare builds of the tree (rebuild oracle/ with another RT_ROTATION_PASSES etc.).
Usage: exp_box_tests.py [detail]"""
import os, sys, json, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import scenes
from oracle import oracle_py

detail = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
sc = scenes.car_standin(480, 270, detail=detail)
P = sc.params
w, h = 240, 135
c2w = np.array(P.camera2World[:], np.float64).reshape(4, 4).T
inv = np.array(P.cameraInverseProjection[:], np.float64).reshape(4, 4).T
xs, ys = np.meshgrid((np.arange(w) + 0.5) / w * 2 - 1, (np.arange(h) + 0.5) / h * 2 - 1)
d = (inv @ np.stack([xs.ravel(), ys.ravel(), np.zeros(w * h), np.ones(w * h)]))[:3]
d = (c2w[:3, :3] @ d).T
d /= np.linalg.norm(d, axis=1, keepdims=True)
o = np.tile(c2w[:3, 3], (w * h, 1))
out = {}
for passes in (0,):
    t0 = time.time()
    orc = oracle_py.Oracle(sc).build()
    r = orc.trace(o.astype(np.float32), d.astype(np.float32), grid=True)
    hit = r["triangle"] != 0xFFFFFFFF
    p = o[hit] + d[hit] * r["t"][hit, None] - d[hit] * 1e-3
    L = -np.array(P.lightParameter[:3], np.float64)
    rng = np.random.default_rng(1)
    sd = L + rng.normal(size=(hit.sum(), 3)) * 0.08
    sd /= np.linalg.norm(sd, axis=1, keepdims=True)
    r2 = orc.trace(p.astype(np.float32), sd.astype(np.float32), grid=True)
    # bounce-like rays: random directions from the hit points
    bd = rng.normal(size=(hit.sum(), 3)); bd /= np.linalg.norm(bd, axis=1, keepdims=True)
    r3 = orc.trace(p.astype(np.float32), bd.astype(np.float32), grid=True)
    f = lambda c: round(c["internalVisits"] / c["rays"], 2)
    out = {"primary": f(r["counters"]), "shadow": f(r2["counters"]), "random": f(r3["counters"]), "s": round(time.time() - t0, 1)}
    print(json.dumps(out))
