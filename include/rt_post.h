/* rt_post.h -- numeric rules of the image post-process chain, shared by the HIP
 * kernels and the CPU checker (same role as rt_math.h for the path tracer).
 *
 * The reference's chain is fragment shaders drawn with Graphics.Blit
 * (the .shader files under Assets/Shaders/PostProcess); three things it leaves to the platform
 * are fixed here:
 *   - texture addressing and filtering (fixed-function samplers): texel centres
 *     at (i + 0.5) / n, bilinear weights with 8 fractional bits, lerp as
 *     a + t * (b - a);
 *   - pow(): rt_pow of rt_math.h;
 *   - UnityCG.cginc's LinearToGammaSpace / GammaToLinearSpace (Unity 2021.3.1f1
 *     built-in shader include, not part of the reference tree), restated from
 *     their published text.
 * All arithmetic is IEEE binary32 in the written order, no contraction.
 */
#ifndef RT_POST_H_
#define RT_POST_H_

#include "rt_math.h"

#ifndef RT_WRAP_CLAMP
#define RT_WRAP_CLAMP 0   /* RenderTexture default */
#define RT_WRAP_REPEAT 1  /* imported Texture2D default (the LUT .meta files: wrapU 0) */
#endif

/* ---------------------------------------------------------------- samplers */
RT_HD int32_t rt_tex_wrap(int32_t i, int32_t n, int wrap)
{
    if (wrap == RT_WRAP_REPEAT) {
        i %= n;
        return i < 0 ? i + n : i;
    }
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

typedef struct RtTexAxis {
    int32_t i0, i1;   /* the two texels */
    float f;          /* weight of i1 */
} RtTexAxis;

/* FilterMode.Bilinear along one axis of an n-texel image.  The texel-space
 * coordinate is snapped to 8 fractional bits (the subtexel precision Direct3D
 * requires of a sampler), so a coordinate that is a texel centre up to rounding
 * reads that texel alone. */
RT_HD RtTexAxis rt_tex_bilinear_axis(float u, int32_t n, int wrap)
{
    RtTexAxis a;
    float x = u * (float)n - 0.5f;
    if (!(x > -1.0e6f)) x = -1.0e6f;   /* also catches NaN */
    if (x > 1.0e6f) x = 1.0e6f;
    const int32_t fx = (int32_t)rt_floor(x * 256.0f + 0.5f);
    const int32_t i = fx >> 8;          /* arithmetic shift = floor */
    a.f = (float)(fx & 255) * 0.00390625f;
    a.i0 = rt_tex_wrap(i, n, wrap);
    a.i1 = rt_tex_wrap(i + 1, n, wrap);
    return a;
}

/* FilterMode.Point along one axis */
RT_HD int32_t rt_tex_point_axis(float u, int32_t n, int wrap)
{
    float x = u * (float)n;
    if (!(x > -1.0e6f)) x = -1.0e6f;
    if (x > 1.0e6f) x = 1.0e6f;
    return rt_tex_wrap((int32_t)rt_floor(x), n, wrap);
}

/* centre of destination pixel p of an n-pixel-wide Blit, in uv units */
RT_HD float rt_blit_uv(int32_t p, int32_t n) { return ((float)p + 0.5f) / (float)n; }

RT_HD float rt_bilerp(float t00, float t10, float t01, float t11, float fx, float fy)
{
    const float lo = rt_lerp(t00, t10, fx);
    const float hi = rt_lerp(t01, t11, fx);
    return rt_lerp(lo, hi, fy);
}

/* --------------------------------------------- UnityCG.cginc colour space */
RT_HD float rt_unity_linear_to_gamma1(float c)
{
    c = rt_max(c, 0.0f);
    return rt_max(1.055f * rt_pow(c, 0.416666667f) - 0.055f, 0.0f);
}
RT_HD float rt_unity_gamma_to_linear1(float c) { return c * (c * (c * 0.305306011f + 0.682171111f) + 0.012522878f); }

/* ------------------------------------------- DownSample / UpSample.shader */
/* GaussWeight2D (DownSample.shader:46-54, UpSample.shader:47-55) */
RT_HD float rt_gauss_weight2d(float x, float y, float sigma)
{
    const float PI = 3.14159265358f;
    const float E = 2.71828182846f;
    const float sigma_2 = rt_pow(sigma, 2.0f);
    const float a = -(x * x + y * y) / (2.0f * sigma_2);
    return rt_pow(E, a) / (2.0f * PI * sigma_2);
}

/* ------------------------------------------------------------ Bloom.shader */
/* ACESToneMapping (Bloom.shader:46-56), one channel */
RT_HD float rt_aces1(float color, float adapted_lum)
{
    const float A = 2.51f, B = 0.03f, C = 2.43f, D = 0.59f, E = 0.14f;
    color *= adapted_lum;
    return (color * (A * color + B)) / (color * (C * color + D) + E);
}

/* -------------------------------------------------------- SaveTexture.cs */
/* float -> 8-bit UNORM as ReadPixels / SetPixels store it: clamp, scale, round half up */
RT_HD uint32_t rt_unorm8(float c)
{
    if (!(c > 0.0f)) return 0u;          /* negative and NaN -> 0 */
    if (c >= 1.0f) return 255u;
    return (uint32_t)(c * 255.0f + 0.5f);
}

#endif
