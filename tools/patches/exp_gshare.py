#!/usr/bin/env python3
"""What handing subtrees from wave to wave (RT_OPT_SHARE_ACROSS_WAVES) does to one walker launch.
Primary rays of the car scene, `frames` frames of 1/share of the image's tiles, traced with rt_trace_rays_device with the
option on and off.  With a library built with -DTRV_GSHARE_STATS (RTCORE_LIB) also the hand-over's counters.
Usage: exp_gshare.py [share] [frames]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import dist as rtdist, native, scenes, wire  # noqa: E402

share = int(sys.argv[1]) if len(sys.argv) > 1 else 1
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 1
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
nb = len(pm) // 64
pix = pm.view(nb, 1, 64).expand(nb, frames, 64).reshape(-1)
g = torch.Generator(device=dev); g.manual_seed(1)
jit = torch.rand(len(pix), 2, device=dev, generator=g)
x = (pix % W).float() + jit[:, 0]
y = (pix // W).float() + jit[:, 1]
ndc = torch.stack([x / W * 2 - 1, y / H * 2 - 1, torch.zeros_like(x), torch.ones_like(x)], 1)
d = (ndc @ inv.T)[:, :3]
d = d @ c2w[:3, :3].T
d = d / d.norm(dim=1, keepdim=True)
n = len(pix)
o4 = torch.zeros(n, 4, device=dev); o4[:, :3] = c2w[:3, 3]
d4 = torch.zeros(n, 4, device=dev); d4[:, :3] = d
hits = torch.zeros(n, 4, device=dev)
torch.cuda.synchronize()
NAMES = ["entries taken", "'over' entries written", "waits ended by 'over'", "entries given", "-", "results stored by way of a slot",
         "-", "waves that left without waiting", "clocks waited"]
ref = None
for on in (1, 0, 1, 0):
    core.set_option(wire.OPT_SHARE_ACROSS_WAVES, on)
    ts = []
    for _ in range(5):
        core.reset_stats()
        core.trace_rays_device(o4.data_ptr(), d4.data_ptr(), n, hits.data_ptr())
        ts.append(round(core.stats()["renderMs"], 3))
    torch.cuda.synchronize()
    got = hits.clone()
    if ref is None:
        ref = got
    same = bool((got.view(torch.int32) == ref.view(torch.int32)).all())
    line = f"share_across_waves={on}  {n / 1e6:.2f} M rays  ms {ts}  same bits as first run: {same}"
    a, b = (C.c_ulonglong * 16)(), (C.c_ulonglong * 16)()
    core.lib.rt_debug_walker_profile(core.ctx, a, b)
    if any(b):
        line += "\n    (last launch) " + ", ".join(f"{NAMES[k]} {b[k]}" for k in range(len(NAMES)))
    print(line, flush=True)
