"""GPU parity tests: the HIP library (through its C ABI) against the CPU checker
on the same seeded inputs.  Integer / index / box outputs must be bit-exact;
radiance must agree to per-pixel relative L2 <= 1e-3."""
import numpy as np
import pytest

from raytracing_amd import scenes, wire
from tests import helpers as H

pytestmark = pytest.mark.gpu

RADIANCE_REL_L2 = 1e-3


def _specials():
    return np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 1e-30, 1e30, 3.14159265, 1e6, -1e6, np.inf, -np.inf, np.nan,
                     4294967296.0, 4294967040.0, 2147483648.0, 0.99999994, 1.0000001], np.float32)


def test_device_math_is_bit_identical_to_host(core, oracle):
    rs = np.random.Generator(np.random.PCG64(11))
    x_big = np.concatenate([rs.uniform(-3e5, 3e5, 200000), rs.uniform(-10, 10, 100000), _specials()]).astype(np.float32)
    x_unit = np.concatenate([rs.uniform(-1.0, 1.0, 200000), _specials()]).astype(np.float32)
    x_pos = np.concatenate([rs.uniform(0, 4, 200000), _specials()]).astype(np.float32)
    y_pow = np.concatenate([rs.uniform(-4, 8, 200000), _specials()[::-1]]).astype(np.float32)
    for which, x, y in ((0, x_big, None), (1, x_big, None), (2, x_unit, None), (3, x_pos, y_pow), (4, x_unit, None),
                        (5, x_unit, None), (6, x_big * np.float32(1e4), None)):
        want = oracle.math(which, x, y)
        got = core.debug_eval(which, x, y)
        H.assert_bits_equal(got, want, f"rt_math function {which}")


@pytest.mark.parametrize("px,py,seed", [(0, 0, 0.25), (1, 1, 0.25), (7, 3, 0.25), (1919, 1079, 0.25), (3839, 2159, 0.73)])
def test_device_rand_stream(core, oracle, px, py, seed):
    want, _ = oracle.rand_sequence(px, py, seed, 128)
    got = core.debug_eval(7, np.array([px, py], np.float32), np.array([seed], np.float32), n=128)
    H.assert_bits_equal(got, want, "Rand() stream")


def test_device_rand_stream_against_an_independent_restatement(core):
    """The same stream against tests/independent.py, which shares no source with the library (the checker does)."""
    from tests import independent
    for px, py, seed in ((1, 1, 0.25), (7, 3, 0.25), (1919, 1079, 0.25), (3839, 2159, 0.73), (640, 360, 17.0), (0, 5, 0.25)):
        want, _ = independent.rand_sequence(px, py, seed, 128)
        got = core.debug_eval(7, np.array([px, py], np.float32), np.array([seed], np.float32), n=128)
        H.assert_bits_equal(got, want, f"Rand() stream of pixel ({px}, {py}), seed {seed}")


def _scene(name):
    if name == "cornell":
        return scenes.cornell_box(64, 64, sphere_segments=32)
    if name == "soup20k":
        return scenes.triangle_soup(20000, seed=5)
    if name == "dups":
        return H.duplicate_heavy_scene()
    if name.startswith("tiny"):
        return scenes.triangle_soup(int(name[4:]), seed=9)
    raise KeyError(name)


@pytest.mark.parametrize("name", ["cornell", "soup20k", "dups", "tiny1", "tiny2", "tiny3", "tiny5", "tiny4097"])
def test_bvh_build_bit_exact(core, oracle, name):
    sc = _scene(name)
    want = oracle.Oracle(sc).build().readback_bvh()
    core.upload(sc)
    core.bvh_build()
    got = core.readback_bvh()
    for key in ("morton_unsorted", "morton_sorted_raw", "morton", "sorted_index", "tri_aabb", "triangles", "internal",
                "leaves", "bvh_data"):
        H.assert_bits_equal(got[key], want[key], f"{name}:{key}")


def test_empty_scene(core):
    sc = scenes.triangle_soup(4, seed=1)
    core.scene_upload(sc.vertices[:0], sc.indices[:0], sc.material_index[:0], sc.shadow_index[:0])
    core.materials_upload(sc.materials)
    core.bvh_build()
    o, d = H.random_rays(64, 1, -1, 1)
    r = core.trace_rays(o, d)
    assert np.isinf(r["t"]).all() and (r["triangle"] == 0xFFFFFFFF).all()


@pytest.mark.parametrize("packets", [False, True], ids=["single", "packets"])
@pytest.mark.parametrize("name", ["cornell", "soup20k", "dups", "tiny1", "tiny2"])
def test_first_hit_bit_exact(core, oracle, name, packets):
    """Both walkers (a ray per lane; 16 rays per packet) must take every ray through exactly the
    reference's decisions: same hit, same bits, same number of box and triangle tests."""
    sc = _scene(name)
    orc = oracle.Oracle(sc).build()
    core.upload(sc)
    core.bvh_build()
    if name == "cornell":
        o1, d1 = H.camera_rays(sc.params, 96)
        o2, d2 = H.random_rays(20000, 2, -9, 14)
    else:
        o1, d1 = H.random_rays(20000, 3, -110, 110)
        o2, d2 = scenes.stress_rays(20000, seed=4)
    # axis-parallel and degenerate directions exercise the inf / NaN slab cases
    o3 = np.tile(np.array([[0.5, 3.0, -20.0]], np.float32), (6, 1))
    d3 = np.array([[0, 0, 1], [1, 0, 0], [0, -1, 0], [0, 0, 0], [np.nan, 0, 1], [0, 1e-30, 1]], np.float32)
    o = np.concatenate([o1, o2, o3])
    d = np.concatenate([d1, d2, d3])
    core.set_option(wire.OPT_PACKET_TRACE, packets)
    # With RT_OPT_PRUNE_BY_DISTANCE 0 caller rays enter every box they pass: the walk whose work the checker counts.
    # The default (1) leaves boxes out by the bounded rule -- same hits, less work --, 2 by the 1.5 % rule.
    core.set_option(wire.OPT_PRUNE_BY_DISTANCE, 0)
    core.enable_counters(True)
    core.reset_stats()
    got = core.trace_rays(o, d)
    st = core.stats()
    core.set_option(wire.OPT_WIDE_NODES, 0)
    core.reset_stats()
    got_narrow = core.trace_rays(o, d)
    st_narrow = core.stats()
    core.set_option(wire.OPT_WIDE_NODES, 1)
    core.set_option(wire.OPT_PRUNE_BY_DISTANCE, 2)
    core.reset_stats()
    got_counting_fewer = core.trace_rays(o, d)
    st_fewer = core.stats()
    core.enable_counters(False)
    got_fewer = core.trace_rays(o, d)             # the build without counters is a different kernel
    core.set_option(wire.OPT_PRUNE_BY_DISTANCE, 0)
    got_every_box = core.trace_rays(o, d)
    core.set_option(wire.OPT_PRUNE_BY_DISTANCE, 1)
    got_plain = core.trace_rays(o, d)
    core.enable_counters(True)
    core.reset_stats()
    got_counting_default = core.trace_rays(o, d)
    st_default = core.stats()
    core.enable_counters(False)
    core.set_option(wire.OPT_PACKET_TRACE, False)
    for k in ("triangle", "t", "u", "v"):
        H.assert_bits_equal(got_plain[k], got[k], f"{k}: counting vs plain kernel")
        H.assert_bits_equal(got_narrow[k], got[k], f"{k}: two levels of the tree per step vs one")
        H.assert_bits_equal(got_every_box[k], got[k], f"{k}: option 0, plain kernel vs counting kernel")
        H.assert_bits_equal(got_counting_default[k], got[k], f"{k}: counting kernel with the default vs option 0")
        H.assert_bits_equal(got_fewer[k], got[k], f"{k}: boxes beyond the closest hit left out")
        H.assert_bits_equal(got_counting_fewer[k], got[k], f"{k}: counting kernel, boxes beyond the closest hit left out")
    assert st_fewer["rays"] == st["rays"] and st_fewer["hits"] == st["hits"]
    assert st_default["rays"] == st["rays"] and st_default["hits"] == st["hits"]
    assert st_default["internalVisits"] <= st["internalVisits"] and st_default["triTests"] <= st["triTests"]
    assert st_fewer["internalVisits"] <= st["internalVisits"] and st_fewer["triTests"] <= st["triTests"]
    if not packets and name in ("cornell", "soup20k"):
        assert st_fewer["internalVisits"] < st["internalVisits"]
    want = orc.trace(o, d)                        # the reference's walk
    # work counters: the packet walker takes the reference's walk node for node; the one-ray-per-lane walker tests
    # half-float outward-rounded boxes of its own tree, two levels per step by default (restated on the CPU by
    # trace(wide=True)) or one (trace(grid=True))
    want_counters = want["counters"] if packets else orc.trace(o, d, wide=True)["counters"]
    want_narrow = want["counters"] if packets else orc.trace(o, d, grid=True)["counters"]
    H.assert_bits_equal(got["triangle"], want["triangle"], "first-hit triangle id")
    H.assert_bits_equal(got["t"], want["t"], "hit distance")
    hit = want["triangle"] != 0xFFFFFFFF
    H.assert_bits_equal(got["u"][hit], want["u"][hit], "barycentric u")
    H.assert_bits_equal(got["v"][hit], want["v"][hit], "barycentric v")
    for k in ("rays", "internalVisits", "leafVisits", "triTests", "hits"):
        assert st[k] == want_counters[k], f"counter {k}: gpu {st[k]} vs oracle {want_counters[k]}"
        assert st_narrow[k] == want_narrow[k], f"counter {k}, one level per step: gpu {st_narrow[k]} vs oracle {want_narrow[k]}"
    for k in ("rays", "triTests", "hits"):        # the same whichever way the tree is walked
        assert st[k] == want["counters"][k]
    if name in ("cornell", "soup20k"):
        assert hit.sum() > 100


def test_first_hit_bit_exact_on_the_full_car(core, oracle):
    """The bench scene at full detail (495 842 triangles): camera rays, rays from the car's surfaces towards the light
    (what the soft-shadow rays are) and rays between random points of its box, against the checker's walk of the
    reference's tree; work counters of the default walker against the checker's restatement of it."""
    sc = scenes.car_standin(64, 64, detail=1.0)
    orc = oracle.Oracle(sc).build()
    core.upload(sc)
    core.bvh_build()
    o1, d1 = H.camera_rays(sc.params, 48)
    first = orc.trace(o1, d1)
    hit = first["triangle"] != 0xFFFFFFFF
    assert hit.sum() > 500
    rs = np.random.Generator(np.random.PCG64(77))
    light = -np.array(sc.params.lightParameter[:3], np.float64)
    o2 = (o1[hit].astype(np.float64) + d1[hit].astype(np.float64) * (first["t"][hit, None] - 1e-3)).astype(np.float32)
    d2 = light[None, :] + rs.normal(scale=0.1, size=(len(o2), 3))
    d2 = (d2 / np.linalg.norm(d2, axis=1, keepdims=True)).astype(np.float32)
    v = np.asarray(sc.vertices["position"], np.float32).reshape(-1, 3)
    lo, hi = v.min(axis=0), v.max(axis=0)
    lo[1] = max(lo[1], 0.0)
    a = (lo + (hi - lo) * rs.random((3000, 3))).astype(np.float32)
    b = (lo + (hi - lo) * rs.random((3000, 3))).astype(np.float32)
    d3 = b - a
    d3 /= np.linalg.norm(d3, axis=1, keepdims=True)
    o = np.concatenate([o1, o2, a])
    d = np.concatenate([d1, d2, d3.astype(np.float32)])
    want = orc.trace(o, d)
    core.set_option(wire.OPT_PRUNE_BY_DISTANCE, 0)       # every box a ray passes: the walk the checker counts
    core.enable_counters(True)
    core.reset_stats()
    got = core.trace_rays(o, d)
    st = core.stats()
    core.set_option(wire.OPT_PRUNE_BY_DISTANCE, 1)       # the default: boxes left out by the bounded rule
    core.reset_stats()
    got_default = core.trace_rays(o, d)
    st_default = core.stats()
    core.enable_counters(False)
    plain = core.trace_rays(o, d)
    for k in ("triangle", "t", "u", "v"):
        H.assert_bits_equal(got[k], want[k], f"{k}: counting kernel vs the reference's walk")
        H.assert_bits_equal(got_default[k], want[k], f"{k}: counting kernel, bounded rule vs the reference's walk")
        H.assert_bits_equal(plain[k], want[k], f"{k}: plain kernel vs the reference's walk")
    # the car's triangles are small ones (the ground's two are not): the rule has to leave a good part of the boxes out
    assert st_default["internalVisits"] < 0.8 * st["internalVisits"], (st_default, st)
    want_counters = orc.trace(o, d, wide=True)["counters"]
    for k in ("rays", "internalVisits", "leafVisits", "triTests", "hits"):
        assert st[k] == want_counters[k], f"counter {k}: gpu {st[k]} vs checker {want_counters[k]}"
    assert (want["triangle"] != 0xFFFFFFFF).sum() > 2000


@pytest.mark.parametrize("sampling", [wire.SAMPLING_MIS, wire.SAMPLING_BSDF, wire.SAMPLING_LIGHT, wire.SAMPLING_COSINE,
                                      wire.SAMPLING_UNIFORM])
def test_cornell_radiance(core, oracle, sampling):
    sc = scenes.cornell_box(96, 64, sphere_segments=24, sampling=sampling)
    seeds = scenes.frame_seeds(4, seed=1)
    orc = oracle.Oracle(sc).build()
    want = oracle.render_frames(orc, sc.params, seeds)
    core.upload(sc)
    core.bvh_build()
    core.set_params(sc.params)
    core.set_dirty()
    core.render_frames(seeds)
    got = core.readback(wire.IMAGE_CONVERGED)
    rel = H.relative_l2(got, want)
    assert np.isfinite(got[..., :3]).all()
    assert rel.max() <= RADIANCE_REL_L2, f"max per-pixel relative L2 {rel.max():.3e} (mean {rel.mean():.3e})"
    H.assert_bits_equal(got[..., 3], want[..., 3], "accumulated alpha")
    assert want[..., :3].mean() > 0.01


def test_single_frame_matches_oracle_bitwise(core, oracle):
    """One CSMain dispatch (1 spp): same arithmetic on both sides -> same bits."""
    sc = scenes.cornell_box(80, 48, sphere_segments=24)
    orc = oracle.Oracle(sc).build()
    p = sc.params.copy()
    p.seed = 0.625
    want = orc.render(p)
    core.upload(sc)
    core.bvh_build()
    core.set_params(p)
    core.dispatch()
    got = core.readback(wire.IMAGE_RT)
    H.assert_bits_equal(got, want, "1-spp frame")


def test_two_sphere_and_area_lights(core, oracle):
    sc = scenes.cornell_box(64, 48, sphere_segments=16)
    sc.sphere_lights = np.array([scenes.sphere_light((-3, 11, -2), 0.5, 2.0), scenes.sphere_light((4, 12, 1), 0.5, 1.0)],
                                wire.SPHERE_LIGHT)
    seeds = scenes.frame_seeds(3, seed=2)
    orc = oracle.Oracle(sc).build()
    want = oracle.render_frames(orc, sc.params, seeds)
    core.upload(sc)
    core.bvh_build()
    core.set_params(sc.params)
    core.set_dirty()
    core.render_frames(seeds)
    got = core.readback(wire.IMAGE_CONVERGED)
    assert H.relative_l2(got, want).max() <= RADIANCE_REL_L2


def test_shadowmap_dof_and_matte(core, oracle):
    sc = scenes.car_standin(64, 48, detail=0.02)
    seeds = scenes.frame_seeds(2, seed=3)
    orc = oracle.Oracle(sc).build()
    want = oracle.render_frames(orc, sc.params, seeds, use_shadow=True)
    p = sc.params.copy()
    p.seed = float(seeds[1])
    want_sm = orc.shadowmap(p)
    core.upload(sc)
    core.bvh_build()
    core.set_params(sc.params)
    core.set_dirty()
    core.render_frames(seeds)
    got = core.readback(wire.IMAGE_CONVERGED)
    got_sm = core.readback(wire.IMAGE_SHADOWMAP)
    H.assert_bits_equal(got_sm, want_sm, "shadow map")
    assert H.relative_l2(got, want).max() <= RADIANCE_REL_L2
    assert (want_sm[..., 0] < 1).any(), "scene should contain shadowed pixels"


def test_manual_dispatch_accumulate_equals_render_frames(core):
    sc = scenes.cornell_box(64, 64, sphere_segments=16)
    seeds = scenes.frame_seeds(3, seed=5)
    core.upload(sc)
    core.bvh_build()
    core.set_params(sc.params)
    core.set_dirty()
    core.render_frames(seeds)
    a = core.readback(wire.IMAGE_CONVERGED)
    assert core.sample_count == 3
    p = sc.params.copy()
    for k, s in enumerate(seeds):
        p.seed = float(s)
        core.set_params(p)
        core.dispatch()
        core.accumulate(k, False)
    b = core.readback(wire.IMAGE_CONVERGED)
    H.assert_bits_equal(a, b, "render_frames vs dispatch+accumulate")


@pytest.mark.parametrize("flags_clear", [0, wire.FLAG_DOF_PRIMARY, wire.FLAG_DOF_PRIMARY | wire.FLAG_DEPTH_OF_FIELD])
def test_batched_frames_equal_frame_by_frame_driver(core, flags_clear):
    """rt_render_frames keeps several frames in flight (and lets the ShadowMap pass reuse CSMain's
    primary hits when both use the same camera); the image must be the one CustomRender produces
    frame by frame (RayTracing.cs:118-164): CSMain, ShadowMap, Blit."""
    sc = scenes.car_standin(72, 40, detail=0.02)
    p = sc.params.copy()
    p.flags &= ~flags_clear
    seeds = scenes.frame_seeds(11, seed=9)        # one full batch and a ragged one
    core.upload(sc)
    core.bvh_build()
    core.set_params(p)
    core.set_dirty()
    core.set_option(wire.OPT_BATCH_FRAMES, 8)
    core.render_frames(seeds)
    by_eight = core.readback(wire.IMAGE_CONVERGED)
    core.set_option(wire.OPT_BATCH_FRAMES, 0)     # default: as many frames in flight as fit the path budget (all 11 here)
    core.set_dirty()
    core.render_frames(seeds)
    H.assert_bits_equal(by_eight, core.readback(wire.IMAGE_CONVERGED), "batches of 8 + 3 vs one batch of 11")
    a = core.readback(wire.IMAGE_CONVERGED)
    a_rt = core.readback(wire.IMAGE_RT)
    a_sm = core.readback(wire.IMAGE_SHADOWMAP)
    assert core.sample_count == 11
    core.set_dirty()
    for k, s in enumerate(seeds):
        p.seed = float(s)
        core.set_params(p)
        core.dispatch()
        core.dispatch_shadowmap()
        core.accumulate(k, True)
    H.assert_bits_equal(a, core.readback(wire.IMAGE_CONVERGED), "batched vs frame-by-frame")
    H.assert_bits_equal(a_rt, core.readback(wire.IMAGE_RT), "last frame of the batch")
    H.assert_bits_equal(a_sm, core.readback(wire.IMAGE_SHADOWMAP), "last shadow map of the batch")


def test_packet_walkers_do_not_change_a_bit(core):
    sc = scenes.car_standin(96, 64, detail=0.02)
    seeds = scenes.frame_seeds(3, seed=10)
    core.upload(sc)
    core.bvh_build()
    core.set_params(sc.params)
    images = []
    for on in (1, 0):
        core.set_option(wire.OPT_PACKET_SHADOW, on)
        core.set_option(wire.OPT_PACKET_PRIMARY, on)
        core.set_option(wire.OPT_MATERIAL_SORT, 1 - on)
        core.set_option(wire.OPT_TWO_STREAMS, on)
        core.set_dirty()
        core.render_frames(seeds)
        images.append((core.readback(wire.IMAGE_CONVERGED), core.readback(wire.IMAGE_SHADOWMAP)))
    core.set_option(wire.OPT_MATERIAL_SORT, 1)
    core.set_option(wire.OPT_TWO_STREAMS, -1)
    H.assert_bits_equal(images[0][0], images[1][0], "converged image, options on vs off")
    H.assert_bits_equal(images[0][1], images[1][1], "shadow map, packets on vs off")
    assert (images[0][1][..., 0] < 1).any()


def test_shadow_rays_walked_while_the_bounces_run_do_not_change_a_bit(core):
    """From some bounce on the ShadowMap pass's rays are walked on most of the chip while the remaining bounces
    run on the rest, and a second launch drains the same queue afterwards: whatever the bounce it starts
    at, the images are those of the plain order."""
    sc = scenes.car_standin(96, 64, detail=0.02)
    seeds = scenes.frame_seeds(5, seed=12)
    core.upload(sc)
    core.bvh_build()
    core.set_params(sc.params)
    images = []
    for on, after in ((0, 1), (1, 1), (1, 0), (1, 6), (1, 3)):
        core.set_option(wire.OPT_OVERLAP_SHADOW, on)
        core.set_option(wire.OPT_OVERLAP_AFTER, after)
        for _ in range(2):
            core.set_dirty()
            core.render_frames(seeds)
        images.append((core.readback(wire.IMAGE_CONVERGED), core.readback(wire.IMAGE_SHADOWMAP)))
    core.set_option(wire.OPT_OVERLAP_SHADOW, 1)
    core.set_option(wire.OPT_OVERLAP_AFTER, 0)
    assert (images[0][1][..., 0] < 1).any() and (images[0][1][..., 0] == 1).any()
    for k in range(1, len(images)):
        H.assert_bits_equal(images[0][0], images[k][0], f"converged image, overlap variant {k}")
        H.assert_bits_equal(images[0][1], images[k][1], f"shadow map, overlap variant {k}")


def test_long_walks_shared_between_lanes_do_not_change_a_bit(core, oracle):
    """At the end of a launch lanes that have run out of rays take over stacked subtrees of the lanes that still
    walk and hand their hit back (RT_OPT_SHARE_LONG_WALKS).  Launches of few rays are all end: closest hits of
    incoherent rays through the car and a soup, on against off against the checker's walk of the reference's tree,
    in the ordinary library and in the one that keeps two stack levels in LDS (where only the bottom level can be
    handed over, and stacks that run out restart on the stackless walk); then frames with the ShadowMap pass, whose
    rays may stop at their first hit."""
    import os
    from raytracing_amd import native
    libs = [("librtcore.so", core, False)]
    path = os.path.join(os.path.dirname(native.LIB_PATH), "librtcore_spill2.so")
    if os.path.exists(path):
        libs.append(("librtcore_spill2.so", native.RtCore(0, lib_path=path), True))
    try:
        for sc_name, sc in (("car", scenes.car_standin(96, 64, detail=0.05)), ("soup", scenes.triangle_soup(20000, seed=5))):
            v = np.asarray(sc.vertices["position"], np.float32).reshape(-1, 3)
            lo, hi = v.min(axis=0), v.max(axis=0)
            rs = np.random.Generator(np.random.PCG64(31))
            rays = []
            for n in (7, 64, 700, 5000):                       # from one wave with most lanes idle to a few waves per unit
                o = (lo + (hi - lo) * rs.random((n, 3))).astype(np.float32)
                t = (lo + (hi - lo) * rs.random((n, 3))).astype(np.float32)
                d = t - o
                d /= np.linalg.norm(d, axis=1, keepdims=True)
                rays.append((o, d.astype(np.float32)))
            orc = oracle.Oracle(sc).build()
            for lib_name, c, _ in libs:
                c.upload(sc)
                c.bvh_build()
                for o, d in rays:
                    want = orc.trace(o, d)
                    for prune, wide in ((2, 0), (0, 0), (2, 1), (1, 1), (0, 1)):   # (0 with the two-level walk: the step that sorts nothing)
                        c.set_option(wire.OPT_PRUNE_BY_DISTANCE, prune)
                        c.set_option(wire.OPT_WIDE_NODES, wide)
                        for on in (1, 0):
                            c.set_option(wire.OPT_SHARE_LONG_WALKS, on)
                            got = c.trace_rays(o, d)
                            for k in ("triangle", "t", "u", "v"):
                                H.assert_bits_equal(got[k], want[k], f"{sc_name}, {lib_name}, {len(o)} rays, sharing {on}, pruning {prune}, wide nodes {wide}: {k}")
                c.set_option(wire.OPT_PRUNE_BY_DISTANCE, 1)
                c.set_option(wire.OPT_SHARE_LONG_WALKS, 1)
                c.set_option(wire.OPT_WIDE_NODES, 1)
        sc = scenes.car_standin(96, 64, detail=0.05)
        seeds = scenes.frame_seeds(3, seed=5)
        images = []
        for lib_name, c, _ in libs:
            c.upload(sc)
            c.bvh_build()
            c.set_params(sc.params)
            for on, wide in ((1, 0), (0, 0), (1, 1), (0, 1)):
                c.set_option(wire.OPT_SHARE_LONG_WALKS, on)
                c.set_option(wire.OPT_WIDE_NODES, wide)
                for frames in (seeds, seeds[:1]):              # a batch, and one frame in flight
                    c.set_dirty()
                    c.render_frames(frames)
                    images.append((f"{lib_name}, sharing {on}, wide nodes {wide}, {len(frames)} frame(s)", len(frames),
                                   c.readback(wire.IMAGE_CONVERGED), c.readback(wire.IMAGE_SHADOWMAP)))
            c.set_option(wire.OPT_SHARE_LONG_WALKS, 1)
            c.set_option(wire.OPT_WIDE_NODES, 1)
        for label, n, conv, sm in images[1:]:
            ref = next(im for im in images if im[1] == n)
            H.assert_bits_equal(conv, ref[2], f"converged image, {label}")
            H.assert_bits_equal(sm, ref[3], f"shadow map, {label}")
        assert (images[0][3][..., 0] < 1).any()
    finally:
        for _, c, own in libs:
            if own:
                c.close()


@pytest.mark.parametrize("non_casters", ["ground only", "a third of the car", "none", "all", "a flat sheet under the car",
                                         "ground only, light in the xy plane", "ground only, wide light"])
def test_shadow_rays_that_stop_at_their_first_hit_do_not_change_a_bit(core, non_casters):
    """A ShadowMap ray that stays clear of the box round the non-casting triangles ends at its first accepted hit;
    the shadow map is the one of the full closest-hit walk, wherever the non-casters are."""
    import copy
    sc = copy.copy(scenes.car_standin(96, 64, detail=0.02))
    if non_casters == "a flat sheet under the car":
        # an axis-parallel non-casting quad between the ground and the car, facing down: the shadow rays of the
        # ground under the car meet it before any caster, and its own box is flat but for the build's padding
        extra = scenes.SceneBuilder()
        s = 0.7071068
        extra.add(scenes.quad_mesh(), scenes.material_data((0.2, 0.8, 0.2, 1), 0, 0.5), (0, 0.04, 0), (-s, 0, 0, s), (1.2, 0.6, 1),
                  cast_shadow=False)
        e = extra.build("sheet")
        nv = len(sc.vertices)
        sc.vertices = np.concatenate([sc.vertices, e.vertices])
        sc.indices = np.concatenate([sc.indices, e.indices + nv]).astype(np.uint32)
        sc.material_index = np.concatenate([sc.material_index, e.material_index + len(sc.materials)]).astype(np.uint32)
        sc.shadow_index = np.concatenate([sc.shadow_index, e.shadow_index]).astype(np.int32)
        sc.materials = np.concatenate([sc.materials, e.materials])
    sh = np.array(sc.shadow_index, np.int32, copy=True)
    if non_casters == "a flat sheet under the car":
        sh[:-2, 0] = 1                # the sheet alone: the box round the non-casters is as flat as it
    if non_casters == "a third of the car":
        sh[2::3, 0] = 0
    elif non_casters == "none":
        sh[:, 0] = 1
    elif non_casters == "all":
        sh[:, 0] = 0
    sc.shadow_index = sh
    if non_casters == "ground only, light in the xy plane":      # a direction component that is exactly zero
        p = sc.params.copy()
        p.lightParameter[0], p.lightParameter[1], p.lightParameter[2] = -0.6, -0.8, 0.0
        sc.params = p
    if non_casters == "ground only, wide light":                 # a wide cone of shadow-ray directions
        p = sc.params.copy()
        p.lightParameter[3] = 1.5
        sc.params = p
    seeds = scenes.frame_seeds(4, seed=13)
    core.upload(sc)
    core.bvh_build()
    core.set_params(sc.params)
    images = []
    # (first hit decides, rays that cannot meet a caster are not walked): none, each alone, both
    for first_hit, caster_box in ((0, 0), (1, 0), (0, 1), (1, 1)):
        core.set_option(wire.OPT_FIRST_HIT_SHADOW, first_hit)
        core.set_option(wire.OPT_CASTER_BOX, caster_box)
        core.set_dirty()
        core.render_frames(seeds)
        images.append((core.readback(wire.IMAGE_CONVERGED), core.readback(wire.IMAGE_SHADOWMAP)))
    for k in (1, 2, 3):
        H.assert_bits_equal(images[0][0], images[k][0], f"converged image, shadow-ray shortcuts, variant {k}")
        H.assert_bits_equal(images[0][1], images[k][1], f"shadow map, shadow-ray shortcuts, variant {k}")
    if non_casters in ("ground only", "none", "a flat sheet under the car") or non_casters.startswith("ground only,"):
        assert (images[0][1][..., 0] < 1).any()
    if non_casters == "all":
        assert (images[0][1][..., :3] == 1).all()


@pytest.mark.parametrize("world,tile", [(3, 32), (8, 8), (5, 64), (16, 64)])
def test_tiles_union_is_bit_identical_to_whole_frame(core, world, tile):
    """A share of the image queues its camera rays frame by frame, the whole image block by block, and every rank has
    one tile of every group of `world` (5 ranks, 64x64 tiles: 12 tiles, the last group holds two -- three ranks end on a
    tile beyond the image; 16 ranks: four of them have nothing but such a tile): the shares put together are the bits of the
    undivided render."""
    sc = scenes.cornell_box(200, 136, sphere_segments=16)     # not a multiple of the tile size
    seeds = scenes.frame_seeds(2, seed=6)
    core.upload(sc)
    core.bvh_build()
    core.set_params(sc.params)
    core.set_dirty()
    core.render_frames(seeds)
    whole = core.readback(wire.IMAGE_CONVERGED)
    union = np.zeros_like(whole)
    covered = np.zeros(whole.shape[:2], bool)
    for rank in range(world):
        p = sc.params.copy()
        p.tileRank, p.tileCount, p.tileSize = rank, world, tile
        core.set_params(p)
        core.set_dirty()
        core.render_frames(seeds)
        part = core.readback(wire.IMAGE_CONVERGED)
        pm = core.local_pixel_map()
        pm = pm[pm != 0xFFFFFFFF]
        mask = np.zeros(whole.shape[0] * whole.shape[1], bool)
        mask[pm] = True
        mask = mask.reshape(whole.shape[:2])
        assert not (covered & mask).any()
        covered |= mask
        union[mask] = part[mask]
        assert (part[~mask] == 0).all()
    assert covered.all()
    H.assert_bits_equal(union, whole, "tile union")


def test_two_live_contexts_split_one_frame(core):
    """Two contexts alive at once (the in-process form of two ranks), each rendering
    its half of the tiles from a thread of its own; both ways of assembling the frame
    -- rt_readback_merge on the host, TileGatherer.place on the device -- must give
    the bits of the undivided render."""
    import threading
    import torch
    from raytracing_amd import dist as rtdist, native
    sc = scenes.cornell_box(200, 136, sphere_segments=16)
    seeds = scenes.frame_seeds(3, seed=6)
    core.upload(sc)
    core.bvh_build()
    core.set_params(sc.params)
    core.set_dirty()
    core.render_frames(seeds)
    whole = core.readback(wire.IMAGE_CONVERGED)
    whole_shadow = core.readback(wire.IMAGE_SHADOWMAP)

    n = 2
    parts = [native.RtCore(0) for _ in range(n)]
    errors = []

    def run(rank):
        try:
            c = parts[rank]
            c.upload(sc)
            c.bvh_build()
            p = sc.params.copy()
            p.tileRank, p.tileCount, p.tileSize = rank, n, 32
            c.set_params(p)
            c.set_dirty()
            c.render_frames(seeds)
        except Exception as e:          # noqa: BLE001 -- reported by the main thread
            errors.append((rank, e))

    try:
        threads = [threading.Thread(target=run, args=(r,)) for r in range(n)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        merged = np.full_like(whole, np.float32(-1))
        merged_shadow = np.full_like(whole, np.float32(-1))
        for c in parts:
            c.readback_merge(wire.IMAGE_CONVERGED, merged)
            c.readback_merge(wire.IMAGE_SHADOWMAP, merged_shadow)
        H.assert_bits_equal(merged, whole, "rt_readback_merge of two contexts")
        H.assert_bits_equal(merged_shadow, whole_shadow, "rt_readback_merge of two contexts (shadow map)")
        # one context alone leaves the other's pixels untouched
        alone = np.full_like(whole, np.float32(-1))
        parts[0].readback_merge(wire.IMAGE_CONVERGED, alone)
        pm = parts[0].local_pixel_map()
        own = np.zeros(whole.shape[0] * whole.shape[1], bool)
        own[pm[pm != 0xFFFFFFFF]] = True
        own = own.reshape(whole.shape[:2])
        assert (alone[~own] == -1).all() and np.array_equal(alone[own].view(np.uint32), whole[own].view(np.uint32))
        with pytest.raises(ValueError):
            parts[0].readback_merge(wire.IMAGE_CONVERGED, np.zeros((4, 4, 4), np.float32))
        # device-side placement
        g = rtdist.TileGatherer(sc.params.width, sc.params.height, 32, torch.device("cuda:0"), contexts=n)
        for r, c in enumerate(parts):
            buf = g.rank_buffer(r)
            assert buf.numel() == c.local_pixel_count * 4
            c.readback_local_device(wire.IMAGE_CONVERGED, buf.data_ptr())
        torch.cuda.synchronize()
        H.assert_bits_equal(g.place().cpu().numpy(), whole, "TileGatherer.place of two contexts")
        # the C ABI's own device-side merge: both contexts write their tiles into ONE frame in device memory
        frame = torch.full((sc.params.height, sc.params.width, 4), -1.0, dtype=torch.float32, device="cuda:0")
        frame_shadow = torch.full_like(frame, -1.0)
        torch.cuda.synchronize()
        parts[0].merge_local_device(wire.IMAGE_CONVERGED, frame.data_ptr())
        torch.cuda.synchronize()
        one = frame.cpu().numpy()
        assert (one[~own] == -1).all() and np.array_equal(one[own].view(np.uint32), whole[own].view(np.uint32))
        parts[1].merge_local_device(wire.IMAGE_CONVERGED, frame.data_ptr(), 0)
        for c in parts:
            c.merge_local_device(wire.IMAGE_SHADOWMAP, frame_shadow.data_ptr())
        torch.cuda.synchronize()
        H.assert_bits_equal(frame.cpu().numpy(), merged, "rt_merge_local_device of two contexts vs rt_readback_merge")
        H.assert_bits_equal(frame_shadow.cpu().numpy(), merged_shadow, "rt_merge_local_device of two contexts (shadow map)")
        with pytest.raises(native.RtError):
            parts[0].merge_local_device(wire.IMAGE_CONVERGED, frame.data_ptr(), 99)        # no such device
        # rt_readback in the background into page-locked memory
        pinned = torch.empty((sc.params.height, sc.params.width, 4), dtype=torch.float32).pin_memory()
        pinned.fill_(-3.0)
        core.readback_async(wire.IMAGE_CONVERGED, pinned.data_ptr())
        core.set_dirty()
        core.render_frames(seeds[:1])                       # the next frame is rendered while the copy runs
        core.readback_wait()
        H.assert_bits_equal(pinned.numpy(), whole, "rt_readback_async")
        part_pinned = torch.empty_like(pinned).pin_memory()
        parts[1].readback_async(wire.IMAGE_CONVERGED, part_pinned.data_ptr())
        parts[1].readback_wait()
        H.assert_bits_equal(part_pinned.numpy(), parts[1].readback(wire.IMAGE_CONVERGED), "rt_readback_async of a tile share")
    finally:
        for c in parts:
            c.close()


def test_shadowmap_pass_started_by_dispatch_gives_the_same_bits(core):
    """rt_dispatch may start the ShadowMap pass of its frame (RT_OPT_EAGER_SHADOWMAP); with the option on or off, and
    when the prepared pass is dropped by a call in between, the three images keep their bits."""
    sc = scenes.car_standin(192, 128, detail=0.08)               # matte-mask ground under a shadow caster, depth of field
    assert sc.params.flags & wire.FLAG_MATTE_MASK
    seeds = scenes.frame_seeds(3, seed=12)
    core.upload(sc)
    core.bvh_build()

    def run(eager, disturb=None):
        core.set_option(wire.OPT_EAGER_SHADOWMAP, eager)
        p = sc.params.copy()
        core.set_params(p)
        core.set_dirty()
        for k, seed in enumerate(seeds):
            p.seed = float(seed)
            core.set_params(p)
            core.dispatch()
            if disturb == "dispatch_again":
                core.dispatch()
            elif disturb == "option":
                core.set_option(wire.OPT_CASTER_BOX, 1)
            elif disturb == "params":
                core.set_params(p)
            core.dispatch_shadowmap()
            core.accumulate(k, True)
        return [core.readback(w) for w in (wire.IMAGE_RT, wire.IMAGE_SHADOWMAP, wire.IMAGE_CONVERGED)]

    try:
        want = run(0)
        assert want[1].min() < want[1].max()                     # the shadow map is not flat
        for disturb in (None, "dispatch_again", "option", "params"):
            got = run(1, disturb)
            for name, a, b in zip(("rt", "shadow map", "converged"), got, want):
                H.assert_bits_equal(a, b, f"{name}, eager ShadowMap pass, {disturb}")
        # a ShadowMap pass on its own (no rt_dispatch before it) still traces its own primary rays
        core.set_option(wire.OPT_EAGER_SHADOWMAP, 1)
        p = sc.params.copy()
        p.seed = float(seeds[-1])
        core.set_params(p)
        core.dispatch_shadowmap()
        H.assert_bits_equal(core.readback(wire.IMAGE_SHADOWMAP), want[1], "stand-alone ShadowMap pass")
    finally:
        core.set_option(wire.OPT_EAGER_SHADOWMAP, 1)


@pytest.mark.parametrize("scene", ["car with ShadowMap pass", "cornell"])
def test_two_frames_in_flight_through_the_per_frame_calls_keep_the_bits(core, scene):
    """The per-frame call sequence keeps up to four frames in flight (RT_OPT_PIPELINE_FRAMES: 2, 3, or 1 = 4 lanes): the
    images after every frame and at the end are those of one frame after the other, and of the batch call."""
    if scene == "cornell":
        sc = scenes.cornell_box(320, 200, sphere_segments=24)
    else:
        sc = scenes.car_standin(384, 256, detail=0.15)
    use_shadow = bool(sc.params.flags & wire.FLAG_MATTE_MASK)
    seeds = scenes.frame_seeds(7, seed=33)
    core.upload(sc)
    core.bvh_build()

    def run(pipeline, peek=False, upto=None):
        core.set_option(wire.OPT_PIPELINE_FRAMES, pipeline)
        p = sc.params.copy()
        core.set_params(p)
        core.set_dirty()
        frames = []
        for k, seed in enumerate(seeds[:upto]):
            p.seed = float(seed)
            core.set_params(p)
            core.dispatch()
            if use_shadow:
                core.dispatch_shadowmap()
            core.accumulate(k, use_shadow)
            if peek:                                   # a readback between frames lets both lanes run out first
                frames.append([core.readback(w) for w in (wire.IMAGE_RT, wire.IMAGE_SHADOWMAP, wire.IMAGE_CONVERGED)])
        return frames if peek else [core.readback(w) for w in (wire.IMAGE_RT, wire.IMAGE_SHADOWMAP, wire.IMAGE_CONVERGED)]

    names = ("rt", "shadow map", "converged")
    try:
        want_frames = run(0, peek=True)
        for lanes in (1, 2, 3):                        # (1: as many as the library has, four)
            for upto in (7, 6, 5, 4, 1):               # the last frame on any lane
                got = run(lanes, upto=upto)
                for name, a, b in zip(names, got, want_frames[upto - 1]):
                    if name != "shadow map" or use_shadow:
                        H.assert_bits_equal(a, b, f"{name} after {upto} frames, RT_OPT_PIPELINE_FRAMES {lanes}")
        got_frames = run(1, peek=True)
        for k, (g, w) in enumerate(zip(got_frames, want_frames)):
            for name, a, b in zip(names, g, w):
                if name != "shadow map" or use_shadow:
                    H.assert_bits_equal(a, b, f"{name} of frame {k}, two frames in flight, read back after every frame")
        # every frame's running mean read back in the background, without letting the lanes run out
        import torch
        pinned = [torch.empty((sc.params.height, sc.params.width, 4), dtype=torch.float32).pin_memory() for _ in seeds]
        core.set_option(wire.OPT_PIPELINE_FRAMES, 1)
        p = sc.params.copy()
        core.set_params(p)
        core.set_dirty()
        for k, seed in enumerate(seeds):
            p.seed = float(seed)
            core.set_params(p)
            core.dispatch()
            if use_shadow:
                core.dispatch_shadowmap()
            core.accumulate(k, use_shadow)
            core.readback_async(wire.IMAGE_CONVERGED, pinned[k].data_ptr())
        core.readback_wait()
        for k, w in enumerate(want_frames):
            H.assert_bits_equal(pinned[k].numpy(), w[2], f"running mean after frame {k}, rt_readback_async, two frames in flight")
        # the batch call right after the per-frame calls, and the per-frame calls right after a batch
        run(1, upto=5)
        core.set_dirty()
        core.render_frames(seeds)
        H.assert_bits_equal(core.readback(wire.IMAGE_CONVERGED), want_frames[-1][2], "batch after per-frame calls")
        got = run(1)
        H.assert_bits_equal(got[2], want_frames[-1][2], "per-frame calls after a batch")
    finally:
        core.set_option(wire.OPT_PIPELINE_FRAMES, 1)


def test_spilled_stacks_of_concurrent_launches_do_not_meet(core):
    """A build of the library that keeps two stack levels per lane in LDS instead of sixteen sends nearly every
    ray's stack through the spill areas in device memory.  Launches that run at the same time -- the shadow rays
    under the bounces, the two streams of RT_OPT_TWO_STREAMS, the static-scratch query -- must each have an area
    of their own: the images are the ordinary library's, bit for bit, in every schedule."""
    import os
    import torch
    from raytracing_amd import native
    path = os.path.join(os.path.dirname(native.LIB_PATH), "librtcore_spill2.so")
    if not os.path.exists(path):
        pytest.skip("librtcore_spill2.so has not been built")
    sc = scenes.car_standin(192, 128, detail=0.08)
    seeds = scenes.frame_seeds(4, seed=21)
    core.upload(sc)
    core.bvh_build()
    core.set_params(sc.params)
    core.set_dirty()
    core.render_frames(seeds)
    want = [core.readback(w) for w in (wire.IMAGE_RT, wire.IMAGE_SHADOWMAP, wire.IMAGE_CONVERGED)]
    o, d = H.camera_rays(sc.params, 64)
    want_hits = core.trace_rays(o, d)

    spill = native.RtCore(0, lib_path=path)
    try:
        spill.upload(sc)
        spill.bvh_build()
        spill.set_params(sc.params)
        for name, options in (("shadow rays under the bounces", {}),
                              ("plain order", {wire.OPT_OVERLAP_SHADOW: 0}),
                              ("two streams", {wire.OPT_TWO_STREAMS: 1})):
            for k in (wire.OPT_OVERLAP_SHADOW, wire.OPT_TWO_STREAMS):
                spill.set_option(k, {wire.OPT_OVERLAP_SHADOW: 1, wire.OPT_TWO_STREAMS: 0}[k])
            for k, v in options.items():
                spill.set_option(k, v)
            spill.set_dirty()
            spill.render_frames(seeds)
            for label, which, ref in zip(("rt", "shadow map", "converged"), (wire.IMAGE_RT, wire.IMAGE_SHADOWMAP, wire.IMAGE_CONVERGED), want):
                H.assert_bits_equal(spill.readback(which), ref, f"{label}, stacks spilled, {name}")
        # the per-frame calls keep two frames in flight: four walker launches may be on the chip together
        for k in (wire.OPT_OVERLAP_SHADOW, wire.OPT_TWO_STREAMS):
            spill.set_option(k, {wire.OPT_OVERLAP_SHADOW: 1, wire.OPT_TWO_STREAMS: 0}[k])
        p = sc.params.copy()
        spill.set_dirty()
        for k, seed in enumerate(seeds):
            p.seed = float(seed)
            spill.set_params(p)
            spill.dispatch()
            spill.dispatch_shadowmap()
            spill.accumulate(k, True)
        for label, which, ref in zip(("rt", "shadow map", "converged"), (wire.IMAGE_RT, wire.IMAGE_SHADOWMAP, wire.IMAGE_CONVERGED), want):
            H.assert_bits_equal(spill.readback(which), ref, f"{label}, stacks spilled, per-frame calls with two frames in flight")
        got = spill.trace_rays(o, d)
        for k in ("triangle", "t", "u", "v"):
            H.assert_bits_equal(got[k], want_hits[k], f"{k}: stacks spilled, rt_trace_rays")
        # the query that allocates nothing spills into its static array
        n = len(o)
        o4 = np.zeros((n, 4), np.float32); o4[:, :3] = o
        d4 = np.zeros((n, 4), np.float32); d4[:, :3] = d
        to, td = torch.from_numpy(o4).cuda(), torch.from_numpy(d4).cuda()
        a = torch.full((n, 4), -1.0, dtype=torch.float32, device="cuda")
        b = torch.full((n, 4), -2.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        spill.trace_rays_device(to.data_ptr(), td.data_ptr(), n, a.data_ptr())
        spill.trace_rays_device_static(to.data_ptr(), td.data_ptr(), n, b.data_ptr())
        spill.synchronize()
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    finally:
        spill.close()


def test_resume_from_uploaded_accumulator(core):
    sc = scenes.cornell_box(64, 64, sphere_segments=16)
    seeds = scenes.frame_seeds(4, seed=8)
    core.upload(sc)
    core.bvh_build()
    core.set_params(sc.params)
    core.set_dirty()
    core.render_frames(seeds)
    full = core.readback(wire.IMAGE_CONVERGED)
    core.set_dirty()
    core.render_frames(seeds[:2])
    half = core.readback(wire.IMAGE_CONVERGED)
    core.set_dirty()
    core.upload_converged(half, 2)
    core.render_frames(seeds[2:])
    H.assert_bits_equal(core.readback(wire.IMAGE_CONVERGED), full, "resumed accumulation")


def test_error_codes(core):
    from raytracing_amd.native import RtError
    sc = scenes.triangle_soup(8, seed=1)
    bad = sc.indices.copy()
    bad[3] = 10 ** 6
    with pytest.raises(RtError) as e:
        core.scene_upload(sc.vertices, bad, sc.material_index, sc.shadow_index)
    assert e.value.code == -1 and "out of range" in str(e.value)
    core.upload(sc)
    with pytest.raises(RtError) as e:
        core.set_params(sc.params)
        core.dispatch()                       # scene uploaded but not built
    assert e.value.code == -4


@pytest.mark.parametrize("name", ["cornell", "soup20k", "dups", "tiny2", "tiny3", "tiny4097"])
def test_walk_tree_bit_exact(core, oracle, name):
    """The tree the rays walk (same build steps, finer keys) and every triangle's place in the reference's visiting
    order: device build against the checker's, word for word."""
    sc = _scene(name)
    orc = oracle.Oracle(sc).build()
    core.upload(sc)
    core.bvh_build()
    got, want = core.readback_walk_tree(), orc.readback_walk()
    n = sc.n_triangles
    assert (got["leaf_triangle"] == want["sorted_index"]).all()
    assert (got["leaf_rank"] == want["visit_rank"][want["sorted_index"]]).all()
    assert sorted(got["leaf_rank"].tolist()) == list(range(n))
    ok, _, qboxes = orc.grid()
    assert ok
    nodes = want["internal"]
    child = lambda side: nodes[side + "Node"] | np.where(nodes[side + "NodeType"] != 0, np.uint32(0x80000000), np.uint32(0))
    H.assert_bits_equal(got["nodes"][:, :6], qboxes, "grid boxes of the traversal tree")
    assert (got["nodes"][:, 6] == child("left")).all() and (got["nodes"][:, 7] == child("right")).all()


def test_equal_distances_go_to_the_reference_first_visited(core, oracle):
    """Coincident triangles give exactly equal distances; the reference keeps the one its own loop meets first.
    The library meets them in another order (another tree) and must still report that one."""
    rs = np.random.Generator(np.random.PCG64(5))
    base = scenes.triangle_soup(400, seed=3, extent=2.0, edge=0.8)
    reps = 6
    v = np.tile(base.vertices, reps)
    idx = np.concatenate([base.indices + k * len(base.vertices) for k in range(reps)])
    order = rs.permutation(400 * reps)                              # shuffle so that copies are far apart in the input
    sc = scenes.Scene("coincident", v, idx.reshape(-1, 3)[order].reshape(-1), np.tile(base.material_index, reps)[order],
                      np.tile(base.shadow_index, (reps, 1))[order], base.materials, params=base.params)
    orc = oracle.Oracle(sc).build()
    core.upload(sc)
    core.bvh_build()
    o, d = H.random_rays(50000, 17, -2, 2)
    got, want = core.trace_rays(o, d), orc.trace(o, d)
    hit = want["triangle"] != 0xFFFFFFFF
    assert hit.sum() > 5000
    assert (got["triangle"] == want["triangle"]).all()
    H.assert_bits_equal(got["t"], want["t"], "hit distance")
    # the tie-break really decided: in the library's own order another copy would have come first for many rays
    walk = orc.readback_walk()
    leaf_of = np.empty(len(order), np.int64)
    leaf_of[walk["sorted_index"]] = np.arange(len(order))
    groups = {}                                                    # input triangle i is a copy of base triangle order[i] % 400
    for i, t in enumerate(order):
        groups.setdefault(int(t) % 400, []).append(i)
    decided = 0
    for t in np.unique(want["triangle"][hit])[:200]:
        g = groups[int(order[t]) % 400]
        if min(g, key=lambda i: leaf_of[i]) != min(g, key=lambda i: walk["visit_rank"][i]):
            decided += 1
    assert decided > 20


def test_unusual_orders_of_the_per_frame_calls_with_frames_in_flight(core):
    """A host need not keep to dispatch / ShadowMap / accumulate: frames dispatched and never folded, two folds of one frame,
    a ShadowMap pass without its CSMain, a parameter change in the middle, a first-hit query between two frames.  With
    four frames in flight the images are those of one frame at a time."""
    sc = scenes.car_standin(256, 160, detail=0.1)
    seeds = scenes.frame_seeds(9, seed=71)
    core.upload(sc)
    core.bvh_build()
    o, d = H.camera_rays(sc.params, 16)

    def run(pipeline):
        core.set_option(wire.OPT_PIPELINE_FRAMES, pipeline)
        p = sc.params.copy()
        core.set_params(p)
        core.set_dirty()
        out = []
        k = 0
        for step, seed in enumerate(seeds):
            p.seed = float(seed)
            core.set_params(p)
            if step != 4:
                core.dispatch()                      # (step 4: a ShadowMap pass with no CSMain before it)
            if step == 1:
                continue                             # dispatched, never folded
            if step == 6:
                p.lightParameter[3] = p.lightParameter[3] * 1.5      # the light's radius changes between the two kernels
                core.set_params(p)
            core.dispatch_shadowmap()
            core.accumulate(k, True)
            k += 1
            if step == 3:
                core.accumulate(k, True)             # folded twice
                k += 1
            if step == 5:
                out.append(core.trace_rays(o, d)["t"].copy())        # lets the lanes run out, then goes on
        out += [core.readback(w) for w in (wire.IMAGE_RT, wire.IMAGE_SHADOWMAP, wire.IMAGE_CONVERGED)]
        return out

    try:
        want = run(0)
        for lanes in (1, 2):
            got = run(lanes)
            for name, a, b in zip(("first hits", "rt", "shadow map", "converged"), got, want):
                H.assert_bits_equal(a, b, f"{name}, RT_OPT_PIPELINE_FRAMES {lanes}")
    finally:
        core.set_option(wire.OPT_PIPELINE_FRAMES, 1)


def test_a_share_of_the_image_through_the_per_frame_calls_with_frames_in_flight(core):
    """One rank of three (8x8 tiles in hashed groups) drives its share frame by frame with four frames in flight, then
    with the batch call: both give, on the rank's pixels, the bits of the undivided frame; everything else reads 0."""
    sc = scenes.car_standin(200, 136, detail=0.1)
    use_shadow = bool(sc.params.flags & wire.FLAG_MATTE_MASK)
    seeds = scenes.frame_seeds(6, seed=71)
    core.upload(sc)
    core.bvh_build()
    core.set_params(sc.params)
    core.set_dirty()
    core.render_frames(seeds)
    whole = [core.readback(w) for w in (wire.IMAGE_CONVERGED, wire.IMAGE_SHADOWMAP)]
    p = sc.params.copy()
    p.tileRank, p.tileCount, p.tileSize = 1, 3, 8
    core.set_params(p)
    pm = core.local_pixel_map()
    mask = np.zeros(200 * 136, bool)
    mask[pm[pm != 0xFFFFFFFF]] = True
    mask = mask.reshape(136, 200)
    assert 0.3 < mask.mean() < 0.37

    def check(what):
        for name, got, want in zip(("converged", "shadow map"), [core.readback(w) for w in (wire.IMAGE_CONVERGED, wire.IMAGE_SHADOWMAP)], whole):
            if name == "shadow map" and not use_shadow:
                continue
            H.assert_bits_equal(got[mask], want[mask], f"{name}, {what}")
            assert (got[~mask] == 0).all(), f"{name}, {what}: pixels of other ranks"

    for lanes in (-1, 0, 1, 2, 3, 4):                  # the frames of the batch down one stream, two, or two to four frame lanes
        core.set_option(wire.OPT_TWO_STREAMS, lanes)
        core.set_dirty()
        core.render_frames(seeds)
        check(f"batch call, RT_OPT_TWO_STREAMS {lanes}")
    core.set_option(wire.OPT_TWO_STREAMS, -1)
    core.set_dirty()
    for k, seed in enumerate(seeds):
        p.seed = float(seed)
        core.set_params(p)
        core.dispatch()
        if use_shadow:
            core.dispatch_shadowmap()
        core.accumulate(k, use_shadow)
    check("per-frame calls, four frames in flight")


@pytest.mark.parametrize("scene", ["cornell", "pbr grid", "car with ShadowMap pass"])
def test_later_bounces_in_one_launch_do_not_change_a_bit(core, scene):
    """RT_OPT_CHAIN_BOUNCES: the bounces from the k-th on walked and shaded by one launch (a wave brings 64 paths to
    their ends) against every bounce a launch of its own -- batch call and per-frame calls, k = 1 .. 7."""
    if scene == "cornell":
        sc = scenes.cornell_box(256, 160, sphere_segments=24)
    elif scene == "pbr grid":
        sc = scenes.pbr_sphere_grid(256, 160, sphere_segments=16)
    else:
        sc = scenes.car_standin(256, 160, detail=0.1)
    use_shadow = bool(sc.params.flags & wire.FLAG_MATTE_MASK)
    seeds = scenes.frame_seeds(3, seed=17)
    core.upload(sc)
    core.bvh_build()
    core.set_params(sc.params)

    def images():
        return [core.readback(w) for w in (wire.IMAGE_RT, wire.IMAGE_CONVERGED)]

    try:
        core.set_option(wire.OPT_CHAIN_BOUNCES, 0)
        core.set_dirty()
        core.render_frames(seeds)
        want = images()
        assert np.isfinite(want[1]).all() and want[1][..., :3].max() > 0
        for k in (3, 1, 2, 5, 7):
            core.set_option(wire.OPT_CHAIN_BOUNCES, k)
            core.set_dirty()
            core.render_frames(seeds)
            for name, a, b in zip(("rt", "converged"), images(), want):
                H.assert_bits_equal(a, b, f"{name}, batch call, RT_OPT_CHAIN_BOUNCES {k}")
            core.set_dirty()
            p = sc.params.copy()
            for j, seed in enumerate(seeds):
                p.seed = float(seed)
                core.set_params(p)
                core.dispatch()
                if use_shadow:
                    core.dispatch_shadowmap()
                core.accumulate(j, use_shadow)
            for name, a, b in zip(("rt", "converged"), images(), want):
                H.assert_bits_equal(a, b, f"{name}, per-frame calls, RT_OPT_CHAIN_BOUNCES {k}")
        with pytest.raises(Exception):
            core.set_option(wire.OPT_CHAIN_BOUNCES, 8)
    finally:
        core.set_option(wire.OPT_CHAIN_BOUNCES, 3)


def test_a_build_after_a_build_of_as_many_triangles_reuses_its_buffers_and_keeps_the_bits(core, oracle):
    """A scene whose vertices move keeps its triangle count: the library then builds into the buffers and the scratch of
    the build before (no allocation in between).  Moved vertices, the same scene twice, another count and back again:
    every build's arrays, the two-level tree and first hits against the checker's."""
    import copy
    import torch
    keys = ("morton_unsorted", "morton_sorted_raw", "morton", "sorted_index", "tri_aabb", "triangles", "internal", "leaves", "bvh_data")
    base = scenes.triangle_soup(20000, seed=5)
    moved = copy.deepcopy(base)
    rs = np.random.Generator(np.random.PCG64(3))
    moved.vertices["position"] = (moved.vertices["position"] * np.float32(0.7) + rs.normal(0, 2.0, moved.vertices["position"].shape)).astype(np.float32)
    other = scenes.triangle_soup(7001, seed=8)
    o, d = H.random_rays(20000, 3, -110, 110)
    free = []
    for step, sc in enumerate((base, moved, moved, base, other, base)):
        orc = oracle.Oracle(sc).build()
        want = orc.readback_bvh()
        core.upload(sc)
        core.bvh_build()
        free.append(torch.cuda.mem_get_info()[0])
        got = core.readback_bvh()
        for key in keys:
            H.assert_bits_equal(got[key], want[key], f"build {step}: {key}")
        hits, ref = core.trace_rays(o, d), orc.trace(o, d)
        H.assert_bits_equal(hits["triangle"], ref["triangle"], f"build {step}: first-hit triangle")
        H.assert_bits_equal(hits["t"], ref["t"], f"build {step}: hit distance")
    # builds 1 .. 3 have as many triangles as the build before them: device memory stays as it was (the first build's figure
    # predates what the first trace and readback keep)
    assert free[1] == free[2] == free[3], free
