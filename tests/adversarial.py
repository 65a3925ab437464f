"""Scenes and rays built to break a walk that leaves out boxes beyond the closest hit it holds.

The reference's walk (BVHRayTracing.cginc:78-123) enters every box a ray passes and keeps the triangle with the
smallest COMPUTED distance.  A walk that skips a box because it begins beyond the hit it already holds returns the
same triangle unless Moller-Trumbore (BruteForceRayTracing.cginc:166-204) puts a triangle of that box far in FRONT
of the box -- which float32 does for a ray that runs nearly inside the triangle's plane: det = e1 . (d x e2) is then
the small difference of large products, and distance, u and v are all divided by it.

One cell of the scene:
  * a small well-shaped triangle H square to the cell's ray direction at distance t_h from the cell's origin;
  * behind it, fans of large triangles and slivers whose planes contain that direction up to a grazing angle of
    1e-7 .. 1e-4 rad, each starting 1.7 % .. 30 % of t_h behind H (so its own box begins beyond 65/64 t_h) and
    running away from the origin; the rays cross their planes near the corner that faces the origin.
The rays of a cell start at its origin and differ from its direction by a few grazing angles, so they meet the fans
at angles from below zero (back face, culled) to several times the fan's own."""
import numpy as np

from raytracing_amd import scenes, wire

F32 = np.float32


def _unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def grazing_cells(n_cells=2048, fans=3, seed=41, scale=1.0, theta_decades=(-7.0, -4.0), behind=(0.017, 0.3)):
    """-> (Scene, cells) with cells = dict of per-cell float64 arrays (origin, direction, a, b, theta, t_h).
    scale: every length of the scene times this (the reference's cut, determinant < 1e-8, is absolute: in a scene 256
    times smaller a ray has to meet a plane at 1e-4 rad or more for the triangle to survive it, and what Moller-Trumbore
    can be off by is bounded -- the scene for the rule that leaves boxes out with a bound); theta_decades: the grazing
    angles, as powers of ten; behind: how far behind H the fans start, as fractions of H's distance."""
    rs = np.random.Generator(np.random.PCG64(seed))
    side = int(np.ceil(n_cells ** (1 / 3)))
    grid = np.stack(np.meshgrid(*[np.arange(side)] * 3, indexing="ij"), -1).reshape(-1, 3)[:n_cells]
    origin = ((grid - (side - 1) / 2) * (180.0 / side) + rs.uniform(-1, 1, (n_cells, 3))) * scale
    d = _unit(rs.normal(size=(n_cells, 3)))
    helper = np.where(np.abs(d[:, :1]) > 0.9, np.array([[0, 1.0, 0]]), np.array([[1.0, 0, 0]]))
    a = _unit(np.cross(d, helper))
    b = np.cross(d, a)
    t_h = rs.uniform(0.5, 5.0, n_cells) * scale
    theta = 10.0 ** rs.uniform(theta_decades[0], theta_decades[1], n_cells)
    tris = []
    # H: an equilateral triangle of circumradius 0.05 t_h square to d, facing the origin (e1 x e2 against d)
    r = 0.05 * t_h[:, None]
    centre = origin + d * t_h[:, None]
    h0 = centre + r * a
    h1 = centre + r * (-0.5 * a - 0.8660254 * b)
    h2 = centre + r * (-0.5 * a + 0.8660254 * b)
    tris.append(np.stack([h0, h1, h2], 1))
    for f in range(fans):
        th = theta * (0.5, 1.0, 2.0)[f % 3]
        g = np.cos(th)[:, None] * d + np.sin(th)[:, None] * b                # in-plane direction, nearly the ray's
        behind_h = t_h * (1.0 + 10.0 ** rs.uniform(np.log10(behind[0]), np.log10(behind[1]), n_cells))   # where the cell's central ray crosses the plane: 1.7 % .. 30 % behind H
        p = origin + d * behind_h[:, None]
        L1 = rs.uniform(1.0, 3.0, n_cells)[:, None] * scale
        L2 = rs.uniform(0.3, 2.0, n_cells)[:, None] * scale
        sliver = rs.random(n_cells) < 0.5
        phi = np.where(sliver, 10.0 ** rs.uniform(-3, -1, n_cells), np.pi / 2)[:, None]
        side_sign = np.where(rs.random(n_cells) < 0.5, 1.0, -1.0)[:, None]
        e1 = L1 * g
        e2 = L2 * (np.cos(phi) * g + np.sin(phi) * side_sign * a)
        v0 = p - 1e-3 * behind_h[:, None] * (g + side_sign * a)                 # the crossing point sits just inside the corner
        # front face towards the origin: d . (e1 x e2) < 0
        flip = np.einsum("ij,ij->i", d, np.cross(e1, e2)) > 0
        e1f = np.where(flip[:, None], e2, e1)
        e2f = np.where(flip[:, None], e1, e2)
        tris.append(np.stack([v0, v0 + e1f, v0 + e2f], 1))
    pos = np.stack(tris, 1).reshape(-1, 3)                                    # cell-major: H, fan 0, fan 1, ...
    n_tri = len(pos) // 3
    v = np.zeros(len(pos), wire.MESH_VERTEX)
    v["position"] = pos.astype(F32)
    v["normal"] = (0, 1, 0)
    v["tangent"] = (1, 0, 0)
    mats = np.array([scenes.material_data((0.7, 0.7, 0.7, 1), 0, 0.3)], wire.MATERIAL)
    sc = scenes.Scene("grazing cells", v, np.arange(len(pos), dtype=np.uint32), np.zeros(n_tri, np.uint32), np.ones((n_tri, 2), np.int32), mats,
                      notes="adversarial: slivers and large triangles met at 1e-7 .. 1e-4 rad behind a well-shaped hit")
    sc.params = scenes.make_params(32, 32, (0, 0, -260 * scale), fov=50.0, sampling=wire.SAMPLING_COSINE)
    return sc, dict(origin=origin, direction=d, a=a, b=b, theta=theta, t_h=t_h)


def grazing_rays(cells, rays_per_cell, seed=43):
    """Rays of every cell: from its origin (moved sideways by up to a grazing angle's worth), along its direction turned
    towards / away from the fans' planes by -1 .. 3 grazing angles and sideways by a little."""
    rs = np.random.Generator(np.random.PCG64(seed))
    n = len(cells["theta"])
    k = rays_per_cell
    th = cells["theta"][:, None]
    eta = rs.uniform(-1.0, 3.0, (n, k)) * th * rs.choice([0.25, 1.0, 1.0, 2.0], (n, k))
    xi = rs.uniform(-1.0, 1.0, (n, k)) * 0.02
    zeta = rs.uniform(-1.0, 1.0, (n, k)) * th * cells["t_h"][:, None]
    d = cells["direction"][:, None, :] + eta[..., None] * cells["b"][:, None, :] + xi[..., None] * cells["a"][:, None, :]
    d = _unit(d)
    o = cells["origin"][:, None, :] + zeta[..., None] * cells["b"][:, None, :]
    return o.reshape(-1, 3).astype(F32), d.reshape(-1, 3).astype(F32)


def rays_in_the_planes_of(scene, n, seed=53, angle_decades=(-6.0, -1.5), start=(0.05, 6.0)):
    """n rays that run nearly inside the planes of triangles of `scene` itself: for a random triangle, a point in it, an
    in-plane direction turned out of the plane by +-10^angle_decades rad, and an origin `start` units back along that
    direction -- so the ray meets the triangle at a grazing angle with whatever else the scene holds in front of it and
    behind it.  The rays a rule that leaves boxes out has to survive on a scene of ordinary shapes."""
    rs = np.random.Generator(np.random.PCG64(seed))
    v = np.asarray(scene.vertices["position"], np.float64)
    idx = np.asarray(scene.indices, np.int64).reshape(-1, 3)
    t = rs.integers(0, len(idx), n)
    p0, p1, p2 = v[idx[t, 0]], v[idx[t, 1]], v[idx[t, 2]]
    e1, e2 = p1 - p0, p2 - p0
    nrm = np.cross(e1, e2)
    ok = np.linalg.norm(nrm, axis=1) > 0
    nrm[~ok] = (0, 1, 0)
    nrm = _unit(nrm)
    a, b = rs.random(n), rs.random(n)
    flip = a + b > 1
    a[flip], b[flip] = 1 - a[flip], 1 - b[flip]
    point = p0 + a[:, None] * e1 + b[:, None] * e2
    phi = rs.uniform(0, 2 * np.pi, n)
    x = _unit(np.where(np.linalg.norm(e1, axis=1, keepdims=True) > 0, e1, np.array([[1.0, 0, 0]])))
    y = np.cross(nrm, x)
    inplane = np.cos(phi)[:, None] * x + np.sin(phi)[:, None] * y
    theta = 10.0 ** rs.uniform(angle_decades[0], angle_decades[1], n) * rs.choice([-1.0, 1.0], n)
    d = _unit(np.cos(theta)[:, None] * inplane - np.sin(theta)[:, None] * nrm)      # (-normal: towards the front face for theta > 0)
    back = rs.uniform(start[0], start[1], n)
    o = point - d * back[:, None]
    return o.astype(F32), d.astype(F32)
