"""CPU checks of the brute-force primitive path (useBVH = false): the restatement of
BruteForceRayTracing.cginc in the checker against closed forms and against the BVH path,
and the host mirror of MeshCollector.  The reference holds no test or fixture for this path."""
import numpy as np
import pytest

from oracle import oracle_py
from raytracing_amd import host, scenes, wire

INVALID = 0xFFFFFFFF


def only(**kw):
    sc = scenes.Scene("prims", np.zeros(0, wire.MESH_VERTEX), np.zeros(0, np.uint32), np.zeros(0, np.uint32),
                      np.zeros((0, 2), np.int32), np.zeros(0, wire.MATERIAL), primitives=scenes.Primitives(**kw))
    return oracle_py.Oracle(sc)


def rays(n, seed=1, z0=-10.0):
    rng = np.random.default_rng(seed)
    o = np.stack([rng.uniform(-3, 3, n), rng.uniform(-3, 3, n), np.full(n, z0)], 1).astype(np.float32)
    d = np.stack([rng.uniform(-0.3, 0.3, n), rng.uniform(-0.3, 0.3, n), np.ones(n)], 1)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    return o, d


def test_plane_matches_closed_form_and_respects_its_box():
    pl = np.zeros(1, wire.PLANE)
    pl[0]["normal"], pl[0]["position"], pl[0]["size"] = (0, 0, -1), (0, 0, 2), (4, 2, 1)
    o, d = rays(5000)
    r = only(planes=pl).trace_primitives(o, d)
    t = (2.0 - o[:, 2].astype(np.float64)) / d[:, 2]
    p = o + d * t[:, None]
    inside = (np.abs(p[:, 0]) < 2) & (np.abs(p[:, 1]) < 1)
    edge = (np.abs(np.abs(p[:, 0]) - 2) < 1e-4) | (np.abs(np.abs(p[:, 1]) - 1) < 1e-4)
    hit = r["primitive"] != INVALID
    assert (hit == inside)[~edge].all()
    np.testing.assert_allclose(r["t"][hit], t[hit], rtol=1e-5)
    assert (r["normal"][hit] == np.array([0, 0, -1], np.float32)).all()
    assert np.isinf(r["t"][~hit]).all()


def test_sphere_distance_and_rays_starting_inside():
    sp = np.zeros(1, wire.SPHERE)
    sp[0]["center"], sp[0]["radius"] = (0.5, -0.25, 3.0), 1.5
    o, d = rays(5000, seed=2)
    r = only(spheres=sp).trace_primitives(o, d)
    oc = o.astype(np.float64) - sp[0]["center"]
    b = (oc * d).sum(1)
    disc = b * b - ((oc * oc).sum(1) - 1.5 ** 2)
    hit = r["primitive"] != INVALID
    sure = np.abs(disc) > 1e-3
    assert (hit == (disc > 0))[sure].all()
    np.testing.assert_allclose(r["t"][hit & sure], (-b - np.sqrt(np.maximum(disc, 0)))[hit & sure], rtol=2e-4, atol=2e-4)
    pos = o + d * r["t"][:, None]
    np.testing.assert_allclose(r["normal"][hit & sure], ((pos - sp[0]["center"]) / 1.5)[hit & sure], atol=2e-3)
    inside_o = np.tile(np.array([[0.5, -0.25, 3.2]], np.float32), (4, 1))
    assert (only(spheres=sp).trace_primitives(inside_o, d[:4])["primitive"] == INVALID).all()      # :105-107


def test_cube_hits_behind_the_origin_and_overwrites_a_nearer_plane():
    """IntersectCube has neither a `t > 0` nor a `t < hit.distance` test (BruteForceRayTracing.cginc:132-163)."""
    cu = np.zeros(1, wire.CUBE)
    cu[0]["min"], cu[0]["max"] = (-1, -1, 4), (1, 1, 6)
    pl = np.zeros(1, wire.PLANE)
    pl[0]["normal"], pl[0]["position"], pl[0]["size"] = (0, 0, -1), (0, 0, 1), (8, 8, 1)
    o = np.array([[0.2, 0.1, 0.0], [0.2, 0.1, 10.0]], np.float32)
    d = np.array([[0, 0, 1], [0.001, 0.002, 1]], np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = only(planes=pl, cubes=cu).trace_primitives(o, d)
    assert r["primitive"][0] == (wire.PRIM_CUBE << 28) and abs(r["t"][0] - 4.0) < 1e-5        # the plane at t = 1 lost
    assert (r["normal"][0] == np.array([0, 0, -1], np.float32)).all()
    assert r["primitive"][1] == (wire.PRIM_CUBE << 28) and r["t"][1] < 0                       # cube behind the ray
    r2 = only(planes=pl).trace_primitives(o[:1], d[:1])
    assert r2["primitive"][0] == (wire.PRIM_PLANE << 28) and abs(r2["t"][0] - 1.0) < 1e-6


def test_custom_mesh_triangles_agree_with_the_bvh_path():
    """The same world-space triangles through IntersectCustomMesh and through the BVH give the same
    distances; with an identity matrix they are bit-identical (same Moller-Trumbore on the same floats)."""
    soup = scenes.triangle_soup(3000, seed=4, extent=3.0, edge=0.8)
    pos = np.ascontiguousarray(soup.vertices["position"])
    me = np.zeros(1, wire.CUSTOM_MESH)
    me[0]["localToWorldMatrix"] = np.eye(4, dtype=np.float32).reshape(-1)
    me[0]["indicesOffset"], me[0]["indicesCount"] = 0, len(soup.indices)
    orc_p = only(meshes=me, vertices=pos, normals=np.ascontiguousarray(soup.vertices["normal"]), indices=soup.indices.astype(np.int32))
    orc_b = oracle_py.Oracle(soup).build()
    o, d = rays(4000, seed=5, z0=-6.0)
    a, b = orc_p.trace_primitives(o, d), orc_b.trace(o, d, brute=True)
    hit = a["primitive"] != INVALID
    assert (hit == (b["triangle"] != INVALID)).all() and hit.sum() > 500
    assert a["t"][hit].tobytes() == b["t"][hit].tobytes()
    # list order decides ties exactly as the strict `t < hit.distance` says: first triangle wins
    assert ((a["primitive"][hit] & 0x0FFFFFFF) == b["triangle"][hit]).all()


def test_render_with_brute_flag_is_deterministic_and_lit_by_the_sky_only():
    sc = scenes.primitive_room(48, 40)
    orc = oracle_py.Oracle(sc)
    a = orc.render(sc.params)
    b = orc.render(sc.params)
    assert a.tobytes() == b.tobytes() and np.isfinite(a).all() and a[..., :3].max() > 0
    # no primitive is emissive in this path (materialType stays 0, Shade never returns emissionColor):
    sc.primitives.planes["emissionColor"] = 5.0
    c = oracle_py.Oracle(sc).render(sc.params)
    assert c.tobytes() == a.tobytes()


# ---------------------------------------------------------------- MeshCollector mirror
def _renderer(pos, scale=(1, 1, 1), rot=scenes.IDENTITY_Q, mesh=None, active=True, color=(0.5, 0.5, 0.5, 1)):
    return host.MeshRenderer(mesh, host.Material(color=color, _Metallic=0.0, _Glossiness=0.5),
                             host.Transform(pos, rot, scale), activeInHierarchy=active)


def test_mesh_collector_sorts_far_to_near_and_pads_empty_lists():
    cam = host.Camera(host.Transform((0, 0, -10)))
    spheres = [_renderer((0, 0, 0), (2, 2, 2)), _renderer((0, 0, 20), (4, 4, 4)), _renderer((0, 0, 5), active=False),
               _renderer((0, 0, -9), (1, 1, 1))]
    mc = host.MeshCollector(cam, sphereParent=spheres)
    p = mc.Primitives()
    assert [float(c[2]) for c in p.spheres["center"]] == [20.0, 0.0, -9.0]            # OrderByDescending(distance), inactive skipped
    assert [float(r) for r in p.spheres["radius"]] == [2.0, 1.0, 0.5]                 # localScale.x / 2
    assert len(p.planes) == 1 and len(p.cubes) == 1 and len(p.meshes) == 1             # one zeroed element each (:299-300)
    assert not p.planes.view(np.uint8).any() and p.meshes[0]["indicesCount"] == 0
    assert p.spheres["metallic"][0] == np.float32(0.01) and p.spheres["transparent"][0] == -1
    np.testing.assert_allclose(p.spheres["albedo"][0], scenes.srgb_to_linear((0.5, 0.5, 0.5)), rtol=1e-6)


def test_mesh_collector_merges_custom_meshes_and_a_zeroed_pad_never_hits():
    cam = host.Camera(host.Transform((0, 0, -10)))
    box, quad = scenes.box_mesh(), scenes.quad_mesh()
    ms = [_renderer((0, 0, 0), (2, 2, 2), mesh=box), _renderer((0, 0, 8), (3, 3, 1), scenes.quat_from_euler_deg(0, 180, 0), mesh=quad)]
    p = host.MeshCollector(cam, customMeshParent=ms).Primitives()
    assert [int(m["indicesOffset"]) for m in p.meshes] == [0, quad.triangles.size]     # the far quad comes first
    assert len(p.vertices) == len(box.positions) + len(quad.positions) and p.indices.max() == len(p.vertices) - 1
    m = p.meshes[0]["localToWorldMatrix"].reshape(4, 4).T                              # Unity memory order is column-major
    np.testing.assert_allclose(m[:3, 3], (0, 0, 8), atol=1e-6)
    sc = scenes.Scene("mc", np.zeros(0, wire.MESH_VERTEX), np.zeros(0, np.uint32), np.zeros(0, np.uint32),
                      np.zeros((0, 2), np.int32), np.zeros(0, wire.MATERIAL), primitives=p)
    o = np.array([[0.1, 0.2, -10.0], [30.0, 0.0, -10.0]], np.float32)
    d = np.array([[0, 0, 1], [0, 0, 1]], np.float32)
    r = oracle_py.Oracle(sc).trace_primitives(o, d)
    assert r["primitive"][0] >> 28 == wire.PRIM_TRIANGLE and abs(r["t"][0] - 9.0) < 1e-5
    assert r["primitive"][1] == INVALID                                                # the padded zero plane / sphere / cube stay silent here


# ---------------------------------------------------------------- disc lights
def test_disc_light_function_restated_but_never_used():
    """DiscLightImportanceSampling (ImportanceSampling.cginc:147-160) is defined and its lights are
    uploaded (LightImportanceSampling.cs:94-106), but _LightImportanceSampling never calls it."""
    x = np.array([7, 3, 0.5, 1.0, -2.0, 0, 1, 0, 0.8, 0.6, 0.4, 1.0, 6.0, -1.0, 0.75], np.float32)
    out = oracle_py.disc_light(x, 0.25)
    np.testing.assert_allclose(out[:3], x[2:5] + x[5:8] * np.float32(0.001), rtol=0, atol=1e-7)
    assert abs(np.linalg.norm(out[3:6]) - 1) < 1e-5
    assert abs(out[9] - 1 / (np.pi * 0.75 ** 2)) < 1e-6
    np.testing.assert_allclose(out[6:9], x[8:11] / np.pi * max(float(out[3:6] @ x[5:8]), 0.0), rtol=1e-5)
    # a scene's disc lights change no pixel
    sc = scenes.cornell_box(24, 24, sphere_segments=8, sampling=wire.SAMPLING_LIGHT)
    a = oracle_py.Oracle(sc).build().render(sc.params)
    d = np.zeros(1, wire.DISC_LIGHT)
    d[0]["position"], d[0]["normal"], d[0]["radius"] = (0, 10, 0), (0, -1, 0), 3.0
    sc.disc_lights = d
    b = oracle_py.Oracle(sc).build().render(sc.params)
    assert a.tobytes() == b.tobytes()
