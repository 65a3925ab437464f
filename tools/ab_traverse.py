#!/usr/bin/env python3
"""A/B timing of traversal builds: for each library given (RTCORE_LIB), render
a few frames of the bench scene with per-kernel HIP-event timing and print the
per-frame device time of every kernel.  Usage: ab_traverse.py [--frames N] lib1.so lib2.so ..."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, json, numpy as np
sys.path.insert(0, %r)
from raytracing_amd import native, scenes, wire
frames = int(sys.argv[1]); workload = sys.argv[2]
sc = scenes.car_standin(1920, 1080) if workload == "car" else scenes.cornell_box(1920, 1080)
core = native.RtCore(0)
core.upload(sc); core.bvh_build(); core.set_params(sc.params)
import os
for kv in filter(None, os.environ.get("RT_OPTS", "").split(",")):     # e.g. RT_OPTS=0=0,1=0 (rt_set_option pairs)
    k, v = kv.split("="); core.set_option(int(k), int(v))
seeds = scenes.frame_seeds(frames, 1)
core.render_frames(seeds[:2])
core.reset_stats(); core.enable_kernel_timing(True)
rounds = []
for rnd in range(3):
    core.set_dirty(); core.render_frames(seeds)
    rounds.append(core.stats()["renderMs"] / frames)
out = {"ms_per_frame": rounds}
for k in ("k_extend", "k_shadow_rays", "k_shade", "k_generate", "k_sm_generate", "k_sm_setup", "k_sm_resolve", "k_accumulate"):
    ms, n = core.kernel_time(k)
    out[k] = round(ms / (3 * frames), 4)
print(json.dumps(out))
''' % ROOT

frames, workload = "8", "car"
libs = []
args = sys.argv[1:]
while args:
    a = args.pop(0)
    if a == "--frames":
        frames = args.pop(0)
    elif a == "--workload":
        workload = args.pop(0)
    else:
        libs.append(a)
for lib in libs:
    env = dict(os.environ, RTCORE_LIB=os.path.abspath(lib))
    r = subprocess.run([sys.executable, "-c", CHILD, frames, workload], env=env, capture_output=True, text=True, timeout=300)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    print(os.path.basename(lib), line[-1] if line else ("FAILED " + r.stderr[-400:]), flush=True)
