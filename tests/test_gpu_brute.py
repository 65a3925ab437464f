"""GPU parity of the brute-force primitive path (RT_FLAG_BRUTE_FORCE = the reference's
useBVH = false): first hits bit for bit, rendered frames to the radiance bound."""
import numpy as np
import pytest

from oracle import oracle_py
from raytracing_amd import host, native, scenes, wire
from tests import helpers as H

pytestmark = pytest.mark.gpu

RADIANCE_REL_L2 = 1e-3


def rel_l2(got, want):
    num = np.sqrt(((got[..., :3].astype(np.float64) - want[..., :3]) ** 2).sum(-1))
    den = np.maximum(np.sqrt((want[..., :3].astype(np.float64) ** 2).sum(-1)), 1e-6)
    return float((num / den).max())


def test_first_hits_bit_exact(core):
    sc = scenes.primitive_room(64, 64, mesh_segments=24)
    core.upload(sc)
    o, d = scenes.primitive_rays(200000)
    o[:64] = np.asarray(sc.primitives.spheres["center"][0]) + np.float32(0.01)       # origins inside a sphere
    o[64:128] = (np.asarray(sc.primitives.cubes["min"][0]) + np.asarray(sc.primitives.cubes["max"][0])) / 2   # inside a cube
    d[128:160] = (0, 0, 1)                                                            # axis-parallel: 1/0 in the cube test
    got = core.trace_primitives(o, d)
    want = oracle_py.Oracle(sc).trace_primitives(o, d)
    assert (got["primitive"] == want["primitive"]).all()
    H.assert_bits_equal(got["t"], want["t"], "hit.distance")
    gn, wn = np.isnan(got["normal"]), np.isnan(want["normal"])
    assert (gn == wn).all()
    H.assert_bits_equal(np.where(gn, np.float32(0), got["normal"]), np.where(wn, np.float32(0), want["normal"]), "hit.normal")
    kinds = got["primitive"][got["primitive"] != 0xFFFFFFFF] >> 28
    assert all((kinds == k).sum() > 1000 for k in range(4)) and (got["t"] < 0).any()


@pytest.mark.parametrize("sampling", [wire.SAMPLING_UNIFORM, wire.SAMPLING_COSINE, wire.SAMPLING_LIGHT, wire.SAMPLING_BSDF, wire.SAMPLING_MIS])
def test_frames_match_the_checker(core, sampling):
    sc = scenes.primitive_room(96, 80, sampling=sampling)
    seeds = scenes.frame_seeds(3, seed=7)
    core.upload(sc)
    core.set_params(sc.params)
    core.set_dirty()
    core.render_frames(seeds)
    got = core.readback(wire.IMAGE_CONVERGED)
    want = oracle_py.render_frames(oracle_py.Oracle(sc), sc.params, seeds)
    assert rel_l2(got, want) <= RADIANCE_REL_L2
    assert got[..., :3].max() > 0


def test_frame_by_frame_driver_and_mesh_collector(core):
    """RayTracing.Start with useBVH = false (RayTracing.cs:84-93) through the host mirror."""
    cam = host.Camera(host.Transform((0.0, 2.0, -9.0), scenes.quat_from_euler_deg(5, 0, 0)), fieldOfView=50.0)
    mat = lambda c, m, g: host.Material(color=c, _Metallic=m, _Glossiness=g)
    planes = [host.MeshRenderer(None, mat((0.7, 0.7, 0.7, 1), 0, 0.5), host.Transform((0, 0, 0), scenes.IDENTITY_Q, (16, 1, 16)))]
    spheres = [host.MeshRenderer(None, mat((0.9, 0.3, 0.2, 1), k / 3, 0.3 + 0.2 * k), host.Transform((-3 + 2 * k, 0.8, 1.5), scenes.IDENTITY_Q, (1.6,) * 3))
               for k in range(4)]
    meshes = [host.MeshRenderer(scenes.box_mesh(), mat((0.4, 0.8, 0.4, 1), 0, 0.4),
                                host.Transform((-2.5, 2.2, 3.0), scenes.quat_from_euler_deg(25, 40, 10), (1.2,) * 3))]
    mc = host.MeshCollector(cam, planeParent=planes, sphereParent=spheres, customMeshParent=meshes)
    lights = host.LightImportanceSampling(areaLightParent=[host.LightObject(host.Transform((0, 6.5, 1.0), scenes.quat_from_euler_deg(90, 0, 0), (4, 4, 1)))])
    rt = host.RayTracing(cam, None, lights, 80, 64, seed_source=scenes.PCG32(5), meshCollector=mc)
    rt.useBVH = False
    rt.samplingType = "MultipleImportance"
    rt.skybox = scenes.procedural_skybox(32)
    rt.Start()
    params = []
    for _ in range(3):
        rt.CustomRender()
        params.append(rt.params.copy())
    got = rt.convergedRT()
    rt.OnDisable()
    assert all(p.flags & wire.FLAG_BRUTE_FORCE for p in params)
    sc = scenes.Scene("mc", np.zeros(0, wire.MESH_VERTEX), np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros((0, 2), np.int32),
                      np.zeros(0, wire.MATERIAL), primitives=mc.Primitives(), skybox=rt.skybox)
    sc.sphere_lights, sc.area_lights, sc.disc_lights = lights.Lights()
    orc = oracle_py.Oracle(sc)
    conv = np.zeros((64, 80, 4), np.float32)
    for k, p in enumerate(params):
        oracle_py.accumulate(conv, orc.render(p), None, k, False)
    assert rel_l2(got, conv) <= RADIANCE_REL_L2


def test_brute_errors():
    c = native.RtCore(0)
    try:
        sc = scenes.primitive_room(32, 32)
        c.set_params(sc.params)
        with pytest.raises(native.RtError):
            c.dispatch()                                   # flag set, no primitive lists
        with pytest.raises(native.RtError):
            c.trace_primitives(np.zeros((1, 3), np.float32), np.ones((1, 3), np.float32))
        bad = scenes.Primitives(meshes=sc.primitives.meshes.copy(), vertices=sc.primitives.vertices, normals=sc.primitives.normals,
                                indices=sc.primitives.indices[:-1])
        with pytest.raises(native.RtError):
            c.primitives_upload(bad)                       # a mesh reaches past the index buffer
        c.upload(sc)
        c.dispatch()
        with pytest.raises(native.RtError):
            c.dispatch_shadowmap()                         # the ShadowMap kernel needs the BVH
    finally:
        c.close()


def test_disc_light_function_bit_exact_and_disc_lights_change_nothing(core):
    rng = np.random.default_rng(3)
    for k in range(16):
        n = rng.normal(size=3)
        x = np.concatenate([[rng.integers(0, 1920), rng.integers(0, 1080)], rng.uniform(-5, 5, 3), n / np.linalg.norm(n),
                            rng.uniform(0, 1, 3), rng.uniform(-8, 8, 3), [rng.uniform(0.05, 3)]]).astype(np.float32)
        seed = float(np.float32(rng.random()))
        got = core.debug_eval(8, x, np.array([seed], np.float32), n=10)
        H.assert_bits_equal(got, oracle_py.disc_light(x, seed), f"DiscLightImportanceSampling case {k}")
    sc = scenes.cornell_box(48, 48, sphere_segments=8, sampling=wire.SAMPLING_MIS)
    seeds = scenes.frame_seeds(2, seed=3)
    imgs = []
    for with_disc in (False, True):
        if with_disc:
            d = np.zeros(2, wire.DISC_LIGHT)
            d["position"], d["normal"], d["radius"] = (0, 10, 0), (0, -1, 0), 3.0
            sc.disc_lights = d
        core.upload(sc)
        core.bvh_build()
        core.set_params(sc.params)
        core.set_dirty()
        core.render_frames(seeds)
        imgs.append(core.readback(wire.IMAGE_CONVERGED))
    assert imgs[0].tobytes() == imgs[1].tobytes()


def load_primitives_fixture(name):
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"unity_{name}_primitives.npz"))
    prims = scenes.Primitives(z["planes"], z["spheres"], z["cubes"], z["meshes"], z["vertices"], z["normals"], z["indices"])
    sc = scenes.Scene(name, np.zeros(0, wire.MESH_VERTEX), np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros((0, 2), np.int32),
                      np.zeros(0, wire.MATERIAL), primitives=prims, skybox=scenes.procedural_skybox(32))
    sc.sphere_lights, sc.area_lights, sc.disc_lights = z["sphere_lights"], z["area_lights"], z["disc_lights"]
    sc.params = wire.RtParams.from_buffer_copy(z["params"].tobytes())
    return sc


@pytest.mark.parametrize("name,triangles", [("chapter3_4", 0), ("chapter4_1", 13444)])
def test_reference_scenes_without_bvh(core, name, triangles):
    """Assets/Scenes/Chapter3_4.unity and Chapter4_1.unity as the checkout holds them (useBVH = false; the buffers
    MeshCollector.Init uploads were exported by tests/golden/make_unity_golden.py): first hits bit for bit, frames
    to the radiance bound.  The reference's HDR sky is not in the checkout; a procedural one stands in."""
    sc = load_primitives_fixture(name)
    assert sc.primitives.n_triangles == triangles and sc.params.flags & wire.FLAG_BRUTE_FORCE
    sc.params.width, sc.params.height = 96, 60
    sc.params.cameraInverseProjection[:] = scenes.inverse_projection(60.0, 96 / 60).tolist()
    core.upload(sc)
    orc = oracle_py.Oracle(sc)
    cam = np.array(sc.params.camera2World[12:15], np.float32)
    rng = np.random.default_rng(2)
    d = rng.normal(size=(50000, 3)) * (1, 0.5, 1) + (0, -0.3, 0)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    o = np.tile(cam, (len(d), 1)) + rng.normal(size=(len(d), 3)).astype(np.float32)
    got, want = core.trace_primitives(o, d), orc.trace_primitives(o, d)
    assert (got["primitive"] == want["primitive"]).all() and (got["primitive"] != 0xFFFFFFFF).sum() > 1000
    H.assert_bits_equal(got["t"], want["t"], "hit.distance")
    seeds = scenes.frame_seeds(2, seed=4)
    core.set_params(sc.params)
    core.set_dirty()
    core.render_frames(seeds)
    assert rel_l2(core.readback(wire.IMAGE_CONVERGED), oracle_py.render_frames(orc, sc.params, seeds)) <= RADIANCE_REL_L2
