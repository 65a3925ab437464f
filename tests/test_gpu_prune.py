"""Leaving out boxes that begin beyond the closest hit (RT_OPT_PRUNE_BY_DISTANCE) against a scene built to break it.

The reference enters every box a ray passes (BVHRayTracing.cginc:78-123); the rule is exact only as far as
Moller-Trumbore's distance of a triangle is not far in front of the triangle's own box, which float32 does not
promise for a ray nearly in the triangle's plane (tests/adversarial.py).  So the rays whose hits are handed out --
rt_trace_rays*, camera rays -- do not use it unless told to (option value 2), and this test holds the default to the
checker's walk of the reference's tree, bit for bit, on ten million rays of which a known handful would change."""
import numpy as np
import pytest

from raytracing_amd import wire
from tests import adversarial
from tests import helpers as H

pytestmark = pytest.mark.gpu


def test_grazing_rays_and_slivers_first_hits_are_the_reference_walks(core, oracle):
    sc, cells = adversarial.grazing_cells(2048)
    o, d = adversarial.grazing_rays(cells, 4883)            # 10 000 384 rays
    assert len(o) >= 10_000_000
    orc = oracle.Oracle(sc).build()
    want = orc.trace(o, d)
    core.upload(sc)
    core.bvh_build()
    got = {}
    for value in (1, 0, 2):                                  # the default, the reference's walk box for box, every ray pruned
        core.set_option(wire.OPT_PRUNE_BY_DISTANCE, value)
        got[value] = core.trace_rays(o, d)
    core.set_option(wire.OPT_PRUNE_BY_DISTANCE, 1)
    for value in (1, 0):
        H.assert_bits_equal(got[value]["triangle"], want["triangle"], f"first-hit triangle, option {value}")
        H.assert_bits_equal(got[value]["t"], want["t"], f"hit distance, option {value}")
        hit = want["triangle"] != 0xFFFFFFFF
        H.assert_bits_equal(got[value]["u"][hit], want["u"][hit], f"u, option {value}")
        H.assert_bits_equal(got[value]["v"][hit], want["v"][hit], f"v, option {value}")
    # what the rule costs where it is applied to such rays: every changed hit is one whose reference hit lies on a
    # fan (triangle index % 4 != 0) that Moller-Trumbore puts in front of the well-shaped triangle before it
    moved = np.nonzero(got[2]["triangle"] != want["triangle"])[0]
    print(f"\n{len(o)} grazing rays: {len(moved)} first hits change when boxes beyond the closest hit are left out "
          f"(the reference's hit is a fan triangle in {(want['triangle'][moved] % 4 != 0).sum()} of them)")
    assert (want["triangle"] != 0xFFFFFFFF).mean() > 0.9
    assert (want["triangle"][moved] % 4 != 0).all()
    assert (got[2]["t"][moved] >= want["t"][moved]).all()
    assert len(moved) < len(o) // 1000


def test_frames_with_bounce_rays_pruned_are_the_checkers(core, oracle):
    """The rays that do use the rule by default -- bounce rays and soft-shadow rays, whose hits reach the caller only
    as radiance -- on a scene of ordinary shapes: the frame is the checker's bit for bit with the rule on (1, 2) and off."""
    from raytracing_amd import scenes
    sc = scenes.car_standin(96, 64, detail=0.05)
    seeds = scenes.frame_seeds(2, seed=9)
    orc = oracle.Oracle(sc).build()
    want = oracle.render_frames(orc, sc.params, seeds, use_shadow=bool(sc.params.flags & wire.FLAG_MATTE_MASK))
    core.upload(sc)
    core.bvh_build()
    core.set_params(sc.params)
    for value in (1, 0, 2):
        core.set_option(wire.OPT_PRUNE_BY_DISTANCE, value)
        core.set_dirty()
        core.render_frames(seeds)
        got = core.readback(wire.IMAGE_CONVERGED)
        H.assert_bits_equal(got, want, f"converged image, option {value}")
    core.set_option(wire.OPT_PRUNE_BY_DISTANCE, 1)
