#!/usr/bin/env python3
"""The waves of one walker launch that end last, and what they did after the launch ran out of rays.
Needs a library built with -DTRV_SHARE_STATS in api.hip's and render.hip's objects (RTCORE_LIB).
Usage: exp_tail_waves.py [share] [share_long_walks]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import dist as rtdist, native, scenes, wire  # noqa: E402

share = int(sys.argv[1]) if len(sys.argv) > 1 else 8
slw = int(sys.argv[2]) if len(sys.argv) > 2 else 1
W, H = 1920, 1080
sc = scenes.car_standin(W, H)
core = native.RtCore(0)
core.upload(sc); core.bvh_build()
dev = torch.device("cuda:0")
P = sc.params
c2w = torch.tensor(np.array(P.camera2World[:], np.float32).reshape(4, 4).T, device=dev)
inv = torch.tensor(np.array(P.cameraInverseProjection[:], np.float32).reshape(4, 4).T, device=dev)
pm = rtdist.tile_pixel_map(W, H, 64, 0, share)
pm = torch.from_numpy(pm[pm != rtdist.PAD]).to(dev)
g = torch.Generator(device=dev); g.manual_seed(1)
jit = torch.rand(len(pm), 2, device=dev, generator=g)
x = (pm % W).float() + jit[:, 0]
y = (pm // W).float() + jit[:, 1]
ndc = torch.stack([x / W * 2 - 1, y / H * 2 - 1, torch.zeros_like(x), torch.ones_like(x)], 1)
d = (ndc @ inv.T)[:, :3]
d = d @ c2w[:3, :3].T
d = d / d.norm(dim=1, keepdim=True)
n = len(pm)
o4 = torch.zeros(n, 4, device=dev); o4[:, :3] = c2w[:3, 3]
d4 = torch.zeros(n, 4, device=dev); d4[:, :3] = d
hits = torch.zeros(n, 4, device=dev)
core.set_option(wire.OPT_SHARE_ACROSS_WAVES, 0)
core.set_option(wire.OPT_SHARE_LONG_WALKS, slw)
for _ in range(3):
    core.reset_stats()
    core.trace_rays_device(o4.data_ptr(), d4.data_ptr(), n, hits.data_ptr())
    ms = core.stats()["renderMs"]
nw = 6144
buf = np.zeros(nw * 12, np.uint64)
core.lib.rt_debug_share_area.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
assert core.lib.rt_debug_share_area(core.ctx, 3, buf.ctypes.data_as(C.c_void_p), buf.nbytes) == 0
rec = buf.reshape(nw, 12).astype(np.int64)
rec = rec[rec[:, 1] != 0]
t0 = rec[:, 0].min()
wall0 = rec[:, 11].min()
order = np.argsort(-rec[:, 1])
print(f"{n} rays, {ms:.3f} ms, {len(rec)} waves; kernel spans {rec[:, 1].max() - t0} shader clocks, {(rec[:, 11].max() - wall0) / 100.0:.1f} us of the 100 MHz clock")
print("wave  start  dry  end  (clocks since the first wave's start) | after dry: hand-over node tri rest | node trips  busy lanes/trip  main-loop trips  hand-overs")
for i in order[:16]:
    r = rec[i]
    print(f"{i:5d} {r[0] - t0:9d} {r[2] - t0 if r[2] else -1:9d} {r[1] - t0:9d} | {r[3]:8d} {r[4]:8d} {r[5]:8d} {r[6]:8d} | {r[7]:6d} {r[8] / max(r[7], 1):5.1f} {r[9]:6d} {r[10]:6d}")
ends = np.sort(rec[:, 1] - t0)
print("waves still running at 50/60/70/80/90/95/99 % of the span:", [int((ends > f * ends[-1]).sum()) for f in (0.5, 0.6, 0.7, 0.8, 0.9, 0.95, 0.99)])
