"""GPU parity of the post-process chain and of SaveTexture's 8-bit conversion: the HIP
kernels (through the C ABI) against the CPU checker on the same frames, bit for bit
(the chain is float arithmetic in a fixed order on both sides, so the tolerance is zero)."""
import os

import numpy as np
import pytest

from raytracing_amd import host, native, scenes, wire
from raytracing_amd.ingest import png
from tests import helpers as H
from tests.test_post_cpu import hdr_image, luts

pytestmark = pytest.mark.gpu


def assert_frames_equal(got, want, what):
    """Bit equality, except that a NaN matches a NaN whatever its sign and payload bits
    (x86 and gfx950 generate different default NaNs, e.g. for inf - inf)."""
    assert got.shape == want.shape and got.dtype == want.dtype, what
    gn, wn = np.isnan(got), np.isnan(want)
    assert (gn == wn).all(), f"{what}: NaNs in different places"
    H.assert_bits_equal(np.where(gn, np.float32(0), got), np.where(wn, np.float32(0), want), what)


def special_frame(w, h):
    img = hdr_image(w, h, seed=9, peak=6.0)
    img[0, 0, :3] = (np.nan, np.inf, -1.0)
    img[1, 2, :3] = (0.0, 1.0, 1e30)
    img[2, 1, :3] = (-0.0, 0.99999994, 1.0000001)
    return img


@pytest.mark.parametrize("w,h", [(301, 203), (64, 64), (1, 1)])
def test_color_correction_bit_exact(core, oracle, w, h):
    img = special_frame(w, h) if w > 8 else hdr_image(w, h)
    core.image_upload(img)
    core.post_color_correction(wire.IMAGE_USER, 0.96, 1.33, 1.85)
    assert_frames_equal(core.readback(wire.IMAGE_POSTPROCESS), oracle.post_color_correction(img, 0.96, 1.33, 1.85), "colour correction")


@pytest.mark.parametrize("name", ["NeutralLUT_32", "Bleach_Bypass", "You_can_do_it"])
def test_color_grading_bit_exact(core, oracle, name):
    lut = luts()[name]
    img = special_frame(257, 129)
    core.image_upload(img)
    core.post_lut_upload(lut)
    core.post_color_grading(wire.IMAGE_USER)
    assert_frames_equal(core.readback(wire.IMAGE_POSTPROCESS), oracle.post_color_grading(img, lut), f"colour grading through {name}")


def test_published_gradings_agree_through_the_kernel(core):
    """rt_post_color_grading against the reference's own outputs: the five published images of one frame graded
    through five tables (README.md "Color grading") have their ungraded frame in common, so the colours the kernel
    maps onto one image, sent through another table, must give that table's image (tests/test_post_cpu.py,
    check_cross_table: median <= 2 and 90th percentile <= 6 of 255 counts; the wrong table is > 6 off at the median)."""
    from tests.test_post_cpu import check_cross_table
    tables = luts()

    def grade(lin, name):
        n, w = len(lin), 1024
        h = (n + w - 1) // w
        img = np.zeros((h * w, 4), np.float32)
        img[:n, :3] = lin
        img[:, 3] = 1.0
        core.image_upload(img.reshape(h, w, 4))
        core.post_lut_upload(tables[name])
        core.post_color_grading(wire.IMAGE_USER)
        return core.readback(wire.IMAGE_POSTPROCESS).reshape(-1, 4)[:n, :3].astype(np.float64)

    check_cross_table(grade)


BLOOM_CASES = [
    dict(),                                                                      # the reference's defaults
    dict(wrapMode=wire.WRAP_REPEAT),
    dict(mainTexST=(1.0 / 300, 1.0 / 200)),                                      # texel-spaced taps
    dict(downSampleStep=2, downSampleBlurSize=15, upSampleBlurSize=3, downSampleBlurSigma=4.0, upSampleBlurSigma=0.3),
    dict(downSampleStep=5, downSampleBlurSize=4, upSampleBlurSize=1, luminanceThreshold=0.3, bloomIntensity=3.0,
         wrapMode=wire.WRAP_REPEAT, mainTexST=(0.01, 0.02)),
]


@pytest.mark.parametrize("case", range(len(BLOOM_CASES)))
def test_bloom_bit_exact(core, oracle, case):
    bp = wire.RtBloomParams(**BLOOM_CASES[case])
    w, h = (517, 389) if bp.downSampleStep > 5 else (301, 203)
    img = hdr_image(w, h, seed=case, peak=5.0)
    core.image_upload(img)
    core.post_bloom(wire.IMAGE_USER, bp)
    assert_frames_equal(core.readback(wire.IMAGE_POSTPROCESS), oracle.post_bloom(img, bp), f"bloom case {case}")


def test_save_rgb24_bit_exact(core, oracle):
    img = special_frame(333, 77)
    img[5, :256, 0] = np.arange(256, dtype=np.float32) / np.float32(255)
    core.image_upload(img)
    got = core.save_rgb24(wire.IMAGE_USER)
    assert got.tobytes() == oracle.save_rgb24(img).tobytes()


def test_chain_on_a_rendered_frame_and_stack_semantics(core, oracle, tmp_path):
    """RayTracing.CustomRender with enablePostProcess (RayTracing.cs:154-158): every stage of the
    stack draws from convergedRT, so postProcessRT holds the last bound stage (here: bloom)."""
    sc = scenes.cornell_box(160, 128, sphere_segments=16)
    core.upload(sc)
    core.bvh_build()
    core.set_params(sc.params)
    core.set_dirty()
    core.render_frames(scenes.frame_seeds(4, seed=2))
    conv = core.readback(wire.IMAGE_CONVERGED)
    lut = luts()["Crisp_warm"]
    bloom = host.Bloom(downSampleStep=4, luminanceThreshold=0.8)
    st = host.PostProcessStack(host.ColorCorrection(), host.ColorGrading(lutTex=lut), bloom, None)
    st.Init(core)
    st.Render(wire.IMAGE_CONVERGED)
    assert_frames_equal(core.readback(wire.IMAGE_POSTPROCESS), oracle.post_bloom(conv, bloom.Parameters()), "stack: last stage wins")
    # the chained form a host may choose instead: each stage reads what the one before drew
    core.post_color_correction(wire.IMAGE_CONVERGED, 0.96, 1.33, 1.85)
    core.post_color_grading(wire.IMAGE_POSTPROCESS)
    core.post_bloom(wire.IMAGE_POSTPROCESS, bloom.Parameters())
    want = oracle.post_bloom(oracle.post_color_grading(oracle.post_color_correction(conv, 0.96, 1.33, 1.85), lut), bloom.Parameters())
    assert_frames_equal(core.readback(wire.IMAGE_POSTPROCESS), want, "chained stages")
    # plain copy and SaveTexture of both targets
    core.post_blit(wire.IMAGE_CONVERGED)
    assert_frames_equal(core.readback(wire.IMAGE_POSTPROCESS), conv, "blit")
    assert core.save_rgb24(wire.IMAGE_CONVERGED).tobytes() == oracle.save_rgb24(conv).tobytes()

    class RT:                                    # SaveTexture reads rayTracing.core
        pass
    rt = RT()
    rt.core = core
    path = host.SaveTexture(rt).Save(str(tmp_path / "out.png"))
    back = png.read_png(path)
    assert back.shape == (128, 160, 3) and back[::-1].tobytes() == oracle.save_rgb24(conv).tobytes()


def test_bloom_1080p_below_threshold_is_identity_and_4k_runs(core):
    img = hdr_image(1920, 1080, peak=0.9)
    core.image_upload(img)
    core.post_bloom(wire.IMAGE_USER, wire.RtBloomParams(luminanceThreshold=10.0))
    assert core.readback(wire.IMAGE_POSTPROCESS).tobytes() == img.tobytes()
    big = np.full((2160, 3840, 4), 1.5, np.float32)
    core.image_upload(big)
    core.post_bloom(wire.IMAGE_USER, wire.RtBloomParams(bloomIntensity=0.01))
    out = core.readback(wire.IMAGE_POSTPROCESS)
    b = 7 * 1.5 * 0.01
    b = (b * (2.51 * b + 0.03)) / (b * (2.43 * b + 0.59) + 0.14)
    np.testing.assert_allclose(out[..., :3], 1.5 + b ** (1 / 2.2), rtol=0, atol=1e-5)


def test_post_errors():
    c = native.RtCore(0)
    try:
        with pytest.raises(native.RtError):
            c.post_blit(wire.IMAGE_USER)                         # nothing uploaded
        with pytest.raises(native.RtError):
            c.post_blit(wire.IMAGE_CONVERGED)                    # no parameters yet
        c.image_upload(np.zeros((64, 64, 4), np.float32))
        with pytest.raises(native.RtError):
            c.post_color_grading(wire.IMAGE_USER)                # no table
        with pytest.raises(native.RtError):
            c.post_bloom(wire.IMAGE_USER, wire.RtBloomParams())  # 64 >> 7 = 0
        with pytest.raises(native.RtError):
            c.post_bloom(wire.IMAGE_USER, wire.RtBloomParams(downSampleStep=1))
        with pytest.raises(native.RtError):
            c.post_blit(17)
        with pytest.raises(native.RtError):
            c.readback(wire.IMAGE_POSTPROCESS)                   # no stage has drawn yet
    finally:
        c.close()
