#!/usr/bin/env python3
"""Estimate: how many of the ShadowMap pass's walked rays that hit nothing could be answered by an occupancy grid over
the shadow casters (a ray that only crosses empty cells of a conservatively filled grid cannot meet a caster)?
Pinhole primary rays of the bench scene -> hit points -> 16 rays towards the light's disk -> for each ray: does it
pass the casters' box, does it hit (rt_trace_rays_device), does it cross a filled cell (dense sampling along the ray).
Usage: exp_occupancy.py [grid cells per axis] [image width]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import native, scenes
G = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W = int(sys.argv[2]) if len(sys.argv) > 2 else 480
H = W * 9 // 16
sc = scenes.car_standin(W, H)
core = native.RtCore(0); core.upload(sc); core.bvh_build()
dev = torch.device("cuda:0")
P = sc.params
c2w = torch.tensor(np.array(P.camera2World[:], np.float32).reshape(4, 4).T, device=dev)
inv = torch.tensor(np.array(P.cameraInverseProjection[:], np.float32).reshape(4, 4).T, device=dev)
ys, xs = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
ndc = torch.stack([(xs.flatten() + 0.5) / W * 2 - 1, (ys.flatten() + 0.5) / H * 2 - 1, torch.zeros(W * H, device=dev), torch.ones(W * H, device=dev)], 1)
d = (ndc @ inv.T)[:, :3] @ c2w[:3, :3].T
d = d / d.norm(dim=1, keepdim=True)
def trace(o, d):
    n = len(o)
    o4 = torch.zeros(n, 4, device=dev); o4[:, :3] = o
    d4 = torch.zeros(n, 4, device=dev); d4[:, :3] = d
    h = torch.zeros(n, 4, device=dev)
    torch.cuda.synchronize()
    core.trace_rays_device(o4.data_ptr(), d4.data_ptr(), n, h.data_ptr())
    return h
o = c2w[:3, 3].expand(W * H, 3).contiguous()
h = trace(o, d)
hit = torch.isfinite(h[:, 0])
pos = o[hit] + d[hit] * h[hit, 0:1]
L = torch.tensor([-P.lightParameter[0], -P.lightParameter[1], -P.lightParameter[2]], device=dev)
rad = P.lightParameter[3]
up = torch.tensor([0.0, 1.0, 0.0], device=dev)
T = torch.linalg.cross(L, up); T = T / T.norm()
B = torch.linalg.cross(T, L); B = B / B.norm()
g = torch.Generator(device=dev); g.manual_seed(3)
n = len(pos)
ang = torch.rand(n, 16, device=dev, generator=g) * 6.2831853
r = torch.sqrt(torch.rand(n, 16, device=dev, generator=g)) * rad
sd = L.view(1, 1, 3) + T.view(1, 1, 3) * (r * torch.cos(ang)).unsqueeze(-1) + B.view(1, 1, 3) * (r * torch.sin(ang)).unsqueeze(-1)
so = (pos.unsqueeze(1) + 0.001 * sd / sd.norm(dim=2, keepdim=True)).expand(n, 16, 3).reshape(-1, 3).contiguous()
sd = sd.reshape(-1, 3).contiguous()
# casters: triangles whose cast flag is set
idx = torch.from_numpy(sc.indices.astype(np.int64).reshape(-1, 3)).to(dev)
vp = torch.from_numpy(np.ascontiguousarray(sc.vertices["position"]).astype(np.float32)).to(dev)
cast = torch.from_numpy((sc.shadow_index[:, 0] != 0)).to(dev)
tri = vp[idx[cast]]
lo = tri.min(1).values - 0.001; hi = tri.max(1).values + 0.001
bmin, bmax = lo.min(0).values, hi.max(0).values
inv_d = 1.0 / sd
t1 = (bmin - so) * inv_d; t2 = (bmax - so) * inv_d
tn = torch.minimum(t1, t2).max(1).values; tf = torch.maximum(t1, t2).min(1).values
inbox = (tn <= tf) & (tf >= 0)
sh = trace(so, sd)
occl = torch.isfinite(sh[:, 0])
print(f"{len(so)} shadow rays from {n} hit pixels; pass the casters' box: {inbox.float().mean():.3f}; of those hit: {(occl & inbox).sum().item() / inbox.sum().item():.3f}")
# conservative occupancy: every cell a padded box touches, grown by one cell
cell = (bmax - bmin) / G
grid = torch.zeros(G, G, G, dtype=torch.bool, device=dev)
c0 = ((lo - bmin) / cell).floor().long().clamp(0, G - 1) - 1
c1 = ((hi - bmin) / cell).floor().long().clamp(0, G - 1) + 1
c0.clamp_(0, G - 1); c1.clamp_(0, G - 1)
span = (c1 - c0 + 1)
for dx in range(int(span[:, 0].max())):
    for dy in range(int(span[:, 1].max())):
        for dz in range(int(span[:, 2].max())):
            m = (dx < span[:, 0]) & (dy < span[:, 1]) & (dz < span[:, 2])
            cc = c0[m] + torch.tensor([dx, dy, dz], device=dev)
            grid[cc[:, 0], cc[:, 1], cc[:, 2]] = True
print(f"grid {G}^3 over {(bmax - bmin).tolist()}: {grid.float().mean():.3f} of the cells filled")
cand = inbox.nonzero().squeeze(1)
touch = torch.zeros(len(cand), dtype=torch.bool, device=dev)
steps = 4 * G
a, b = tn[cand].clamp(min=0), tf[cand]
for s in range(steps + 1):
    t = a + (b - a) * s / steps
    p = so[cand] + sd[cand] * t.unsqueeze(1)
    c = ((p - bmin) / cell).floor().long().clamp(0, G - 1)
    touch |= grid[c[:, 0], c[:, 1], c[:, 2]]
free = ~touch
miss = ~occl[cand]
print(f"rays in the box: {len(cand)}; miss everything: {miss.float().mean():.3f}; cross only empty cells: {free.float().mean():.3f}; "
      f"of the misses answered by the grid: {(free & miss).sum().item() / max(miss.sum().item(), 1):.3f}; wrongly free (must be 0): {(free & ~miss).sum().item()}")
