#!/usr/bin/env python3
"""Compare the walk trees two builds of the library make (children links per node) on a few soups.
Usage: diff_walk_tree.py libA.so libB.so"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, numpy as np
sys.path.insert(0, %r)
from raytracing_amd import native, scenes
out = {}
core = native.RtCore(0)
for n in (3, 4, 5, 8, 16, 50, 200, 2000):
    sc = scenes.triangle_soup(n, seed=9)
    core.upload(sc); core.bvh_build()
    t = core.readback_walk_tree()
    out[n] = t["nodes"][:, 6:8].astype(int).tolist()
print(json.dumps(out))
''' % ROOT
res = []
for lib in sys.argv[1:3]:
    r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, RTCORE_LIB=os.path.abspath(lib)), capture_output=True, text=True, timeout=600)
    res.append(json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]))
for n in res[0]:
    a, b = res[0][n], res[1][n]
    d = [i for i in range(len(a)) if a[i] != b[i]]
    print(n, "differing nodes:", len(d), d[:8])
    if d and int(n) <= 16:
        print("  A", a)
        print("  B", b)
