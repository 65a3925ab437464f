#!/usr/bin/env python3
"""Does the order of the ray queue decide how long a launch of the persistent walker lasts?
Primary rays of the car scene (all frames of an 8x8 pixel block next to each other, as k_generate queues them)
for 1/share of the image's tiles, traced with rt_trace_rays_device in that order, with the 64-ray groups dealt
out by a stride permutation, and with the groups shuffled.  Usage: exp_group_permute.py [share] [frames]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import dist as rtdist, native, scenes  # noqa: E402

share = int(sys.argv[1]) if len(sys.argv) > 1 else 8
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 64
W, H = 1920, 1080
sc = scenes.car_standin(W, H)
core = native.RtCore(0)
core.upload(sc); core.bvh_build()
dev = torch.device("cuda:0")
P = sc.params
c2w = torch.tensor(np.array(P.camera2World[:], np.float32).reshape(4, 4).T, device=dev)
inv = torch.tensor(np.array(P.cameraInverseProjection[:], np.float32).reshape(4, 4).T, device=dev)
pm = rtdist.tile_pixel_map(W, H, 64, 0, share)
pm = torch.from_numpy(pm[pm != rtdist.PAD]).to(dev)            # tile-local order: 8x8 blocks of 64 consecutive slots
nb = len(pm) // 64
pix = pm.view(nb, 1, 64).expand(nb, frames, 64).reshape(-1)      # [block][frame][pixel]
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
G = n // 64


def run(name, order):
    if order is None:
        oo, dd = o4, d4
    else:
        idx = (order.view(-1, 1) * 64 + torch.arange(64, device=dev).view(1, -1)).reshape(-1)
        oo, dd = o4[idx].contiguous(), d4[idx].contiguous()
    torch.cuda.synchronize()
    ts = []
    for _ in range(4):
        core.trace_rays_device(oo.data_ptr(), dd.data_ptr(), n, hits.data_ptr())
        ts.append(round(core.stats()["renderMs"], 3))
    print(f"{name:28s} {n / 1e6:6.1f} M rays  ms {ts}  -> {n / min(ts) / 1e6:.2f} G rays/s", flush=True)


run("queue order", None)
for blocks in (1, 4, 16, 64, 256):
    units = nb // blocks
    perm = torch.randperm(units, device=dev, generator=g)
    per = blocks * frames                                        # 64-ray groups per unit
    order = (perm.view(-1, 1) * per + torch.arange(per, device=dev).view(1, -1)).reshape(-1)
    rest = torch.arange(units * per, G, device=dev)
    run(f"units of {blocks} blocks shuffled", torch.cat([order, rest]))
for blocks in (16, 64):
    units = nb // blocks
    stride = int(units * 0.618) | 1
    while np.gcd(stride, units) != 1:
        stride += 2
    perm = (torch.arange(units, device=dev) * stride) % units
    per = blocks * frames
    order = (perm.view(-1, 1) * per + torch.arange(per, device=dev).view(1, -1)).reshape(-1)
    run(f"units of {blocks} blocks by stride", torch.cat([order, torch.arange(units * per, G, device=dev)]))
