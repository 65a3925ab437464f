"""Shared helpers for the parity tests."""
import numpy as np

from raytracing_amd import scenes, wire


def bits(a):
    return np.ascontiguousarray(a).view(np.uint8).reshape(-1)


def assert_bits_equal(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, f"{what}: shape/dtype {a.shape}{a.dtype} vs {b.shape}{b.dtype}"
    if a.dtype.names:
        for name in a.dtype.names:
            x, y = a[name], b[name]
            bad = np.nonzero((x.view(np.uint32) != y.view(np.uint32)).reshape(len(a), -1).any(1))[0] if len(a) else []
            assert len(bad) == 0, f"{what}.{name}: {len(bad)} rows differ, first {bad[:5]}: {x[bad[:3]]} vs {y[bad[:3]]}"
        return
    av = a.view(np.uint32) if a.dtype.itemsize == 4 else a
    bv = b.view(np.uint32) if b.dtype.itemsize == 4 else b
    bad = np.nonzero(av.reshape(-1) != bv.reshape(-1))[0]
    assert len(bad) == 0, f"{what}: {len(bad)} of {av.size} words differ, first at {bad[:5]}: " \
                          f"{a.reshape(-1)[bad[:5]]} vs {b.reshape(-1)[bad[:5]]}"


def camera_rays(params, n_side=32):
    """Pinhole rays through an n_side x n_side grid of the view, as float32 arrays."""
    c2w = np.array(params.camera2World[:], np.float64).reshape(4, 4).T
    ip = np.array(params.cameraInverseProjection[:], np.float64).reshape(4, 4).T
    u = (np.arange(n_side) + 0.5) / n_side * 2 - 1
    uu, vv = np.meshgrid(u, u)
    clip = np.stack([uu.ravel(), vv.ravel(), np.zeros(uu.size), np.ones(uu.size)], 0)
    d = (ip @ clip)[:3]
    d = c2w[:3, :3] @ d
    d /= np.linalg.norm(d, axis=0, keepdims=True)
    o = np.repeat(c2w[:3, 3:4], uu.size, 1)
    return o.T.astype(np.float32), d.T.astype(np.float32)


def random_rays(n, seed, lo, hi):
    rs = np.random.Generator(np.random.PCG64(seed))
    o = rs.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rs.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d.astype(np.float32)


def relative_l2(a, b):
    """Per-pixel relative L2 over rgb: |a-b| / max(|b|, eps)."""
    a, b = a[..., :3].astype(np.float64), b[..., :3].astype(np.float64)
    num = np.sqrt(((a - b) ** 2).sum(-1))
    den = np.maximum(np.sqrt((b ** 2).sum(-1)), 1e-6)
    return num / den


def duplicate_heavy_scene(n=3000, seed=3):
    """Many triangles sharing Morton cells (and exact duplicates): stresses the
    duplicate-key path of the sort / DistributeMortonCode / tree."""
    sc = scenes.triangle_soup(n, seed=seed)
    v = sc.vertices.copy()
    pos = v["position"].reshape(n, 3, 3)
    pos[: n // 2] = pos[: n // 2] * np.float32(0.001)          # crowd half of them into a few cells
    pos[n // 2: n // 2 + 200] = pos[0]                          # 200 exact copies of triangle 0
    v["position"] = pos.reshape(-1, 3)
    sc.vertices = v
    return sc
