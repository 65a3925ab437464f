#!/usr/bin/env python3
"""CPU experiment: which share of a ray's box tests falls on the top of the threaded traversal tree?
Walks the checker's threaded tree (oracle grid()) for sample rays of the car scene and tallies the steps by the
depth of the record, to size an LDS-resident tree top."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import scenes
from oracle import oracle_py

detail = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
sc = scenes.car_standin(1920, 1080, detail=detail)
orc = oracle_py.Oracle(sc).build()
ok, g, q = orc.grid()
n = len(q)
leaf = (q[:, 3] >> 31) != 0
link = (q[:, 3] // 16).astype(np.int64)
# depth of every record from the preorder + links
depth = np.zeros(n, np.int32)
stack = []   # (end, depth)
d = 0
ends = []
for i in range(n):
    while ends and ends[-1] == i:
        ends.pop()
    depth[i] = len(ends)
    if not leaf[i]:
        ends.append(link[i])
print("records", n, "max depth", depth.max())
half = lambda b: b.astype(np.uint16).view(np.float16).astype(np.float32)
lo = np.stack([half(q[:, a] & 0xFFFF) for a in range(3)], 1) * g[3:] + g[:3]
hi = np.stack([half(q[:, a] >> 16) for a in range(3)], 1) * g[3:] + g[:3]

def walk(o, dirs):
    hist = np.zeros(64, np.int64)
    for oo, dd in zip(o, dirs):
        r = 1.0 / dd
        i = 0
        mine = np.zeros(64, np.int64)
        while i < n:
            t1 = (lo[i] - oo) * r; t2 = (hi[i] - oo) * r
            tn = np.minimum(t1, t2).max(); tf = np.maximum(t1, t2).min()
            p = tf >= max(tn, 0.0)
            mine[depth[i]] += 1
            i = i + 1 if (leaf[i] or p) else link[i]
        if mine.sum() > 12:
            hist += mine
            walk.long += 1
    return hist
walk.long = 0

rs = np.random.default_rng(1)
P = sc.params
ld = -np.array(P.lightParameter[:3], np.float64)
ld /= np.linalg.norm(ld)
# shadow-like rays: from points on the ground plane around the car and on the car's box toward the light
pos = sc.vertices["position"]
mn, mx = pos.min(0), pos.max(0)
car = pos[np.abs(pos[:, 0]) < 4]
N = 600
o = np.stack([rs.uniform(-3, 3, N), np.full(N, mn[1] + 1e-3), rs.uniform(-2, 2, N)], 1)
t = np.cross(ld, [0, 1, 0]); t /= np.linalg.norm(t); b = np.cross(t, ld)
dirs = ld + P.lightParameter[3] * (rs.uniform(-1, 1, (N, 1)) * t + rs.uniform(-1, 1, (N, 1)) * b)
dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
hs = walk(o, dirs)
cnt = np.bincount(depth, minlength=64)
cum_nodes = np.cumsum(cnt)
cum_steps = np.cumsum(hs) / hs.sum()
print("shadow-like rays with more than 12 steps:", walk.long, "of", N, "steps per such ray", hs.sum() / max(walk.long, 1))
for dpt in range(0, 24):
    print(f"depth <= {dpt:2d}: {cum_nodes[dpt]:8d} records, {100 * cum_steps[dpt]:5.1f} % of the steps")
