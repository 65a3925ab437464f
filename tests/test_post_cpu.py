"""CPU checks of the post-process restatement (oracle/rt_post_oracle.cpp) and of the host
mirror of PostProcessStack.  The reference has no test or golden output for this chain
(parity unpinned); the anchors here are (1) the reference's own look-up tables
(tests/golden/reference_luts.npz, made by make_post_golden.py): grading through its
NeutralLUT_32 must be the identity up to the table's resolution; (2) an independent numpy
statement of every stage in float64; (3) the five published gradings of one frame
(tests/golden/reference_graded_pixels.npz), which must agree with each other through the
grading operator."""
import os

import numpy as np
import pytest

from oracle import oracle_py
from raytracing_amd import host, wire
from raytracing_amd.ingest import png

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def hdr_image(w, h, seed=3, peak=4.0):
    rng = np.random.default_rng(seed)
    img = rng.random((h, w, 4), np.float32) ** 3 * np.float32(peak)
    img[..., 3] = rng.random((h, w), np.float32)
    return img


def luts():
    z = np.load(os.path.join(GOLDEN, "reference_luts.npz"))
    return {k: png.texture_from_u8(z[k]) for k in z.files}


def graded_pixels():
    """The same 20 000 pixels of the five published gradings of one frame (tests/golden/make_post_golden.py)."""
    z = np.load(os.path.join(GOLDEN, "reference_graded_pixels.npz"))
    return {k: z[k] for k in z.files if k not in ("positions", "rows_skipped")}


def srgb_to_linear(c):
    c = np.asarray(c, np.float64)
    return np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)


def linear_to_srgb(c):
    c = np.clip(np.asarray(c, np.float64), 0.0, 1.0)
    return np.where(c <= 0.0031308, c * 12.92, 1.055 * c ** (1 / 2.4) - 0.055)


def cross_table_errors(grade, source="Urban_cowboy", grid=80):
    """The published gradings have their ungraded frame in common.  `grade(linear_rgb[n, 3], table) -> linear_rgb`
    is the operator under test: the colours it maps onto the `source` image are found (densest-grid nearest
    neighbour in display space), pushed through every other table, and compared with that table's published
    image.  Returns {table: (median, 90th percentile) of the absolute error in 8-bit counts}."""
    from scipy.spatial import cKDTree
    px = graded_pixels()
    g = (np.arange(grid) + 0.5) / grid
    cand = srgb_to_linear(np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3))
    shown = linear_to_srgb(grade(cand, source))
    _, idx = cKDTree(shown).query(px[source].astype(np.float64) / 255.0)
    frame = cand[idx]                                          # the ungraded frame at the sampled pixels, as estimated
    out = {}
    for name, want in px.items():
        err = np.abs(linear_to_srgb(grade(frame, name)) * 255.0 - want.astype(np.float64))
        out[name] = (float(np.median(err)), float(np.percentile(err, 90)))
    return out, frame


def check_cross_table(grade):
    errs, frame = cross_table_errors(grade)
    for name, (med, p90) in errs.items():
        assert med <= 2.0 and p90 <= 6.0, (name, med, p90)     # measured: median 0.6 - 1.2, 90th percentile 1.5 - 3.5 counts
    # the pairing matters: the wrong table for an image is far off
    px = graded_pixels()
    wrong = np.abs(linear_to_srgb(grade(frame, "Bleach_Bypass")) * 255.0 - px["Candlelight"].astype(np.float64))
    assert np.median(wrong) > 6.0, float(np.median(wrong))


def test_published_gradings_agree_through_the_checker():
    """oracle/rt_post_oracle.cpp against the reference's own outputs: five published images of one frame, each
    through another table (README.md "Color grading")."""
    tables = luts()

    def grade(lin, name):
        img = np.zeros((1, len(lin), 4), np.float32)
        img[0, :, :3] = lin
        img[..., 3] = 1.0
        return oracle_py.post_color_grading(img, tables[name])[0, :, :3].astype(np.float64)

    check_cross_table(grade)


# ---------------------------------------------------------------- independent numpy statements
def np_bilinear(img, u, v, wrap):
    """float64 bilinear fetch with texel centres at (i + .5)/n and 8 fractional weight bits."""
    h, w = img.shape[:2]

    def axis(t, n):
        x = np.floor((t * n - 0.5) * 256.0 + 0.5).astype(np.int64)
        i, f = x >> 8, (x & 255) / 256.0
        if wrap == wire.WRAP_REPEAT:
            return i % n, (i + 1) % n, f
        return np.clip(i, 0, n - 1), np.clip(i + 1, 0, n - 1), f

    x0, x1, fx = axis(np.asarray(u, np.float64), w)
    y0, y1, fy = axis(np.asarray(v, np.float64), h)
    fx, fy = fx[..., None], fy[..., None]
    lo = img[y0, x0] + (img[y0, x1] - img[y0, x0]) * fx
    hi = img[y1, x0] + (img[y1, x1] - img[y1, x0]) * fx
    return lo + (hi - lo) * fy


def np_gauss(img, dw, dh, n, sx, sy, sigma, wrap):
    u, v = np.meshgrid((np.arange(dw) + 0.5) / dw, (np.arange(dh) + 0.5) / dh)
    r = n // 2
    acc, wsum = 0.0, 0.0
    for i in range(-r, r + 1):
        for j in range(-r, r + 1):
            wgt = np.exp(-(i * i + j * j) / (2.0 * sigma * sigma)) / (2.0 * np.pi * sigma * sigma)
            acc = acc + np_bilinear(img, u + i * sx, v + j * sy, wrap)[..., :3] * wgt
            wsum += wgt
    out = np.zeros((dh, dw, 4))
    out[..., :3] = acc / wsum
    return out


def np_bloom(src, bp):
    src = src.astype(np.float64)
    h, w = src.shape[:2]
    n, wrap = bp.downSampleStep, bp.wrapMode
    sx, sy = bp.mainTexST[0], bp.mainTexST[1]
    lum = src[..., :3] @ np.array([0.2126, 0.7152, 0.0722])
    thr = np.where((lum > bp.luminanceThreshold)[..., None], src, np.array([0.0, 0.0, 0.0, 1.0]))
    downs, cur = [], thr
    for i in range(n):
        cur = np_gauss(cur, w // 2 ** (i + 1), h // 2 ** (i + 1), bp.downSampleBlurSize, sx, sy, bp.downSampleBlurSigma, wrap)
        downs.append(cur)
    prev = downs[n - 1]
    for i in range(n - 1):
        c = downs[n - 2 - i]
        ch, cw = c.shape[:2]
        prev = (np_gauss(prev, cw, ch, bp.upSampleBlurSize, 0.5 * sx, 0.5 * sy, bp.upSampleBlurSigma, wrap) +
                np_gauss(c, cw, ch, bp.upSampleBlurSize, sx, sy, bp.upSampleBlurSigma, wrap))
    u, v = np.meshgrid((np.arange(w) + 0.5) / w, (np.arange(h) + 0.5) / h)
    b = np_bilinear(prev, u, v, wrap)[..., :3] * bp.bloomIntensity
    b = (b * (2.51 * b + 0.03)) / (b * (2.43 * b + 0.59) + 0.14)
    out = src.copy()
    out[..., :3] += np.clip(b ** (1 / 2.2), 0.0, 1.0)
    return out


# ---------------------------------------------------------------- colour correction
def test_color_correction_matches_numpy():
    img = hdr_image(37, 23)
    got = oracle_py.post_color_correction(img, 0.96, 1.33, 1.85)
    rgb = img[..., :3].astype(np.float64)
    gray = rgb @ np.array([0.2125, 0.7154, 0.0721])
    c = rgb * 0.96
    c = gray[..., None] + 1.33 * (c - gray[..., None])
    c = 0.5 + 1.85 * (c - 0.5)
    np.testing.assert_allclose(got[..., :3], c, rtol=0, atol=2e-5)
    assert got[..., 3].tobytes() == img[..., 3].tobytes()            # alpha passes through


def test_color_correction_neutral_settings_are_identity():
    img = hdr_image(16, 9)
    got = oracle_py.post_color_correction(img, 1.0, 1.0, 1.0)
    np.testing.assert_allclose(got, img, rtol=0, atol=1e-6)


# ---------------------------------------------------------------- colour grading
def gamma(x):
    return np.maximum(1.055 * np.maximum(x, 0.0) ** 0.416666667 - 0.055, 0.0)


def test_reference_neutral_lut_is_identity_up_to_its_resolution():
    lut = luts()["NeutralLUT_32"]
    img = hdr_image(64, 48, peak=1.0)
    img[..., :3] = np.minimum(img[..., :3], np.float32(0.9))          # stay below the last blue block (see next test)
    got = oracle_py.post_color_grading(img, lut)
    gi = gamma(img[..., :3].astype(np.float64))
    poly = lambda s: s * (s * (s * 0.305306011 + 0.682171111) + 0.012522878)     # GammaToLinearSpace
    # red and green are point-sampled on a 32-step axis: the table returns floor(31 c) / 31 (within 1/31 of c,
    # stored in 8 bits); blue is interpolated over 32 blocks whose entries step by 1/31, i.e. b * 32 / 31
    want_rg = np.floor(gi[..., :2] * 31) / 31
    assert np.abs(want_rg - gi[..., :2]).max() <= 1.0 / 31
    np.testing.assert_allclose(got[..., :2], poly(want_rg), rtol=0, atol=0.008)
    np.testing.assert_allclose(got[..., 2], poly(gi[..., 2] * 32 / 31), rtol=0, atol=0.008)
    assert got[..., 3].tobytes() == img[..., 3].tobytes()


def test_grading_blue_at_or_above_one_wraps_to_block_zero():
    """floor(b * 32) = 32 addresses u >= 1, which the repeat-wrapped table folds back to the first
    (blue = 0) block (ColorGrading.shader:53-61): white comes out yellow.  Kept literally."""
    lut = luts()["NeutralLUT_32"]
    img = np.zeros((1, 2, 4), np.float32)
    img[0, 0] = (0.5, 0.5, 2.0, 1.0)       # gamma(2) = 1.35 -> block 43 -> wraps to 11, right block = 31
    img[0, 1] = (0.5, 0.5, 0.5, 1.0)
    got = oracle_py.post_color_grading(img, lut)
    g = gamma(np.float64(2.0))
    left = int(np.floor(g * 32)) % 32
    f = g * 32 - np.floor(g * 32)
    want_b = (left / 31) + ((31 / 31) - (left / 31)) * f
    poly = lambda s: s * (s * (s * 0.305306011 + 0.682171111) + 0.012522878)
    assert abs(got[0, 0, 2] - poly(want_b)) < 0.01
    assert abs(got[0, 1, 2] - poly(gamma(0.5) * 32 / 31)) < 0.01


@pytest.mark.parametrize("name", ["Bleach_Bypass", "Candlelight", "Crisp_warm", "Urban_cowboy", "You_can_do_it"])
def test_grading_returns_table_entries_at_table_nodes(name):
    """A colour that sits exactly on a table node (gamma-space r = i/31 + eps, g = j/31 + eps,
    b = k/32) must come back as that node's entry."""
    lut = luts()[name]
    rng = np.random.default_rng(11)
    i, j, k = rng.integers(0, 31, 200), rng.integers(0, 31, 200), rng.integers(0, 32, 200)
    gr, gg, gb = (i + 0.5) / 31.0, (j + 0.5) / 31.0, (k + 0.25) / 32.0      # inside the cell, away from its edges
    inv = lambda s: ((s + 0.055) / 1.055) ** 2.4                              # inverse of LinearToGammaSpace
    img = np.ones((1, 200, 4), np.float32)
    img[0, :, 0], img[0, :, 1], img[0, :, 2] = inv(gr), inv(gg), inv(gb)
    got = oracle_py.post_color_grading(img, lut)
    tx, ty = np.floor(gr * 31).astype(int), np.floor(gg * (31 / 32) * 32).astype(int)
    c0 = lut[ty, k * 32 + tx, :3].astype(np.float64)
    c1 = lut[ty, np.minimum(k + 1, 31) * 32 + tx, :3].astype(np.float64)
    s = c0 + (c1 - c0) * 0.25
    want = s * (s * (s * 0.305306011 + 0.682171111) + 0.012522878)
    np.testing.assert_allclose(got[0, :, :3], want, rtol=0, atol=3e-3)


# ---------------------------------------------------------------- bloom
def test_bloom_below_threshold_leaves_the_frame_untouched():
    img = hdr_image(256, 192, peak=0.9)
    got = oracle_py.post_bloom(img, wire.RtBloomParams(luminanceThreshold=5.0))
    assert got.tobytes() == img.tobytes()


def test_bloom_of_a_constant_bright_frame():
    """Every pyramid level of a constant frame is that constant, so bloomUps[n-2] = n * c."""
    c, n = 1.5, 4
    img = np.full((64, 96, 4), c, np.float32)
    for wrap in (wire.WRAP_CLAMP, wire.WRAP_REPEAT):
        got = oracle_py.post_bloom(img, wire.RtBloomParams(downSampleStep=n, wrapMode=wrap, bloomIntensity=0.05))
        b = n * c * 0.05
        b = (b * (2.51 * b + 0.03)) / (b * (2.43 * b + 0.59) + 0.14)
        np.testing.assert_allclose(got[..., :3], c + min(b ** (1 / 2.2), 1.0), rtol=0, atol=1e-5)
        np.testing.assert_array_equal(got[..., 3], img[..., 3])


@pytest.mark.parametrize("wrap,st", [(wire.WRAP_CLAMP, (1.0, 1.0)), (wire.WRAP_REPEAT, (1.0, 1.0)),
                                     (wire.WRAP_CLAMP, (1.0 / 40, 1.0 / 24)), (wire.WRAP_REPEAT, (0.013, 0.031))])
def test_bloom_matches_numpy(wrap, st):
    img = hdr_image(83, 50, seed=5)
    bp = wire.RtBloomParams(downSampleStep=3, downSampleBlurSize=5, upSampleBlurSize=3, downSampleBlurSigma=1.3,
                            upSampleBlurSigma=0.8, bloomIntensity=0.7, luminanceThreshold=0.8, wrapMode=wrap, mainTexST=st)
    got = oracle_py.post_bloom(img, bp)
    want = np_bloom(img, bp)
    np.testing.assert_allclose(got, want, rtol=0, atol=3e-5)


def test_bloom_default_tap_spacing_reads_the_frame_edges():
    """With `_MainTex_ST.xy` = (1, 1) every tap but the centre one lands a whole frame away and, on a
    clamped RenderTexture, reads the border texels (DownSample.shader:80): one bright pixel in the
    middle therefore changes the bloom much less than the same pixel on the border."""
    bp = wire.RtBloomParams(downSampleStep=2, luminanceThreshold=1.0)
    base = np.zeros((32, 32, 4), np.float32)
    mid, edge = base.copy(), base.copy()
    mid[16, 16, :3] = 50.0
    edge[0, 16, :3] = 50.0
    gm = oracle_py.post_bloom(mid, bp)[..., :3] - mid[..., :3]
    ge = oracle_py.post_bloom(edge, bp)[..., :3] - edge[..., :3]
    assert (ge > 0).mean() > 0.5 > (gm > 0).mean()


def test_bloom_rejects_too_small_a_frame():
    with pytest.raises(ValueError):
        oracle_py.post_bloom(np.zeros((64, 64, 4), np.float32), wire.RtBloomParams(downSampleStep=7))


# ---------------------------------------------------------------- SaveTexture
def test_save_rgb24_double_quantisation():
    lin = np.zeros((2, 256, 4), np.float32)
    lin[0, :, :3] = (np.arange(256, dtype=np.float32) / np.float32(255))[:, None]
    lin[1, :, :3] = np.linspace(-0.5, 1.5, 256, dtype=np.float32)[:, None]
    got = oracle_py.save_rgb24(lin)
    v = np.arange(256) / 255.0
    srgb = np.where(v < 0.0031308, v * 12.92, 1.055 * v ** (1 / 2.4) - 0.055)
    np.testing.assert_array_equal(got[0, :, 0], np.floor(srgb * 255 + 0.5).astype(np.uint8))
    assert got[0, 0, 0] == 0 and got[0, 255, 0] == 255 and (np.diff(got[0, :, 0].astype(int)) >= 0).all()
    assert got[1, 0, 0] == 0 and got[1, -1, 0] == 255                    # clamped
    # the linear 8-bit step in between loses the dark tones: 1/255 linear is already sRGB 13
    assert got[0, 1, 0] == 13


# ---------------------------------------------------------------- PostProcessStack (host mirror)
class FakeCore:
    def __init__(self):
        self.calls = []

    def __getattr__(self, name):
        return lambda *a: self.calls.append((name,) + a)


def test_stack_hands_the_same_source_to_every_stage_and_binds_dof_even_if_disabled():
    core = FakeCore()
    st = host.PostProcessStack(host.ColorCorrection(), host.ColorGrading(lutTex=np.zeros((32, 1024, 4), np.float32)),
                               host.Bloom(), host.DepthOfField(enabled=False))
    st.Init(core)
    st.Render(wire.IMAGE_CONVERGED)
    names = [c[0] for c in core.calls]
    assert names == ["post_lut_upload", "post_color_correction", "post_color_grading", "post_bloom", "post_blit"]
    assert all(c[1] == wire.IMAGE_CONVERGED for c in core.calls[1:])


def test_stack_skips_disabled_stages_and_falls_back_to_a_copy():
    core = FakeCore()
    st = host.PostProcessStack(host.ColorCorrection(enabled=False), None, host.Bloom(enabled=False), None)
    st.Init(core)
    st.Render()
    assert core.calls == [("post_blit", wire.IMAGE_CONVERGED)]
