#!/usr/bin/env python3
"""Device time of rt_bvh_build without per-kernel timers.  Usage: build_time.py [lib.so ...]  (RTCORE_LIB is set per child)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json
sys.path.insert(0, %r)
from raytracing_amd import native, scenes
out = {}
for name, sc in (("car", scenes.car_standin(64, 64)), ("soup10M", scenes.triangle_soup(10_000_000, seed=5, width=64, height=64))):
    core = native.RtCore(0)
    core.upload(sc)
    t = []
    for _ in range(5):
        core.bvh_build()
        t.append(core.stats()["buildMs"])
    out[name] = [round(x, 3) for x in t]
    core.close()
print(json.dumps(out))
''' % ROOT
for lib in sys.argv[1:] or [os.path.join(ROOT, "raytracing_amd", "librtcore.so")]:
    r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, RTCORE_LIB=os.path.abspath(lib)), capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    print(os.path.basename(lib), line[-1] if line else "FAILED " + r.stderr[-300:], flush=True)
