/* rt_math.h -- canonical numeric rules of the rtcore path tracer.
 *
 * The reference shaders leave every transcendental (sin, cos, acos, pow),
 * float->uint conversion and min/max NaN rule to the GPU vendor
 * (Assets/Shaders/RayTracing/Header.cginc:61-96 feeds sin() into an integer
 * RNG state, so one differing ulp changes every later sample).  This header
 * fixes ONE definition of each of those operations, built only from IEEE-754
 * binary32/binary64 + - * / sqrt (all correctly rounded, no FMA contraction)
 * and integer bit manipulation, so that a host build (g++) and a device build
 * (hipcc, gfx950) produce the same bits.  Both the CPU checker and the HIP
 * kernels include this file; neither may be compiled with -ffast-math or with
 * FP contraction enabled.
 *
 * Rules fixed here:
 *   - (uint)f        : D3D ftou -- truncate; NaN or <0 -> 0; >= 2^32 -> 0xFFFFFFFF
 *   - min/max        : return the non-NaN operand (HLSL / IEEE minNum)
 *   - saturate(NaN)  : 0
 *   - normalize(v)   : v / sqrt(dot(v,v))
 *   - dot            : ((ax*bx) + (ay*by)) + (az*bz), left to right
 *   - lerp(a,b,t)    : a + t*(b-a)
 *   - sin/cos/acos/pow: evaluated in binary64 with fixed polynomials, rounded
 *                      once to binary32
 *   - half           : float
 */
#ifndef RT_MATH_H_
#define RT_MATH_H_

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ inline
#else
#define RT_HD inline
#endif

#define RT_PI 3.14159265f          /* Header.cginc:20 */
#define RT_EPSILON 1e-8f           /* Header.cginc:17 */
#define RT_NORMAL_BIAS 0.001f      /* Header.cginc:23 */
#define RT_MAX_BOUNCE 8            /* Header.cginc:5  */

/* ------------------------------------------------------------------ bits */
RT_HD uint32_t rt_f2u_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
RT_HD float rt_u2f_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
RT_HD uint64_t rt_d2u_bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
RT_HD double rt_u2d_bits(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }

RT_HD float rt_inf() { return rt_u2f_bits(0x7F800000u); }
RT_HD float rt_nan() { return rt_u2f_bits(0x7FC00000u); }
RT_HD bool rt_isnan(float x) { return x != x; }

/* D3D ftou. */
RT_HD uint32_t rt_ftou(float f)
{
    if (!(f > 0.0f)) return 0u;              /* NaN, -x, +-0 */
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;                       /* in range: truncation */
}

/* ------------------------------------------------------------ scalar ops */
RT_HD float rt_min(float a, float b)
{
    if (b != b) return a;
    if (a != a) return b;
    return a < b ? a : b;
}
RT_HD float rt_max(float a, float b)
{
    if (b != b) return a;
    if (a != a) return b;
    return a > b ? a : b;
}
RT_HD float rt_clamp(float x, float lo, float hi) { return rt_min(rt_max(x, lo), hi); }
RT_HD float rt_saturate(float x) { return rt_min(rt_max(x, 0.0f), 1.0f); }
RT_HD float rt_abs(float x) { return rt_u2f_bits(rt_f2u_bits(x) & 0x7FFFFFFFu); }
RT_HD float rt_lerp(float a, float b, float t) { return a + t * (b - a); }
RT_HD float rt_sqrt(float x) { return __builtin_sqrtf(x); }
RT_HD float rt_floor(float x) { return __builtin_floorf(x); }
RT_HD float rt_frac(float x) { return x - rt_floor(x); }
RT_HD float rt_sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

/* --------------------------------------------- binary64 transcendental core */
/* Round to nearest integer, ties to even, valid for |x| < 2^51. */
RT_HD double rt__rint(double x)
{
    const double M = 6755399441055744.0; /* 1.5 * 2^52 */
    const double t = x + M;              /* not folded: no fast-math anywhere */
    return t - M;
}

/* sin and cos of a finite binary64 x, |x| < 1e15: Cody-Waite reduction by
 * pi/2 in two pieces, then degree-13/14 minimax polynomials on [-pi/4,pi/4]. */
RT_HD void rt__sincos64(double x, double *s_out, double *c_out)
{
    const double k = rt__rint(x * 6.36619772367581382433e-01);
    const double r = (x - k * 1.57079632673412561417e+00) - k * 6.07710050650619224932e-11;
    const double z = r * r;
    const double ps = r + r * z * (-1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 +
                      z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 +
                      z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)))));
    const double pc = (1.0 - 0.5 * z) + z * z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 +
                      z * (2.48015872894767294178e-05 + z * (-2.75573143513906633035e-07 +
                      z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
    const long long ki = (long long)k;
    switch ((int)(ki & 3)) {
    case 0: *s_out = ps;  *c_out = pc;  break;
    case 1: *s_out = pc;  *c_out = -ps; break;
    case 2: *s_out = -ps; *c_out = -pc; break;
    default: *s_out = -pc; *c_out = ps; break;
    }
}

RT_HD void rt_sincos(float x, float *s, float *c)
{
    const double xd = (double)x;
    if (!(xd < 1e15 && xd > -1e15)) { *s = x - x; *c = x - x; return; }
    double sd, cd;
    rt__sincos64(xd, &sd, &cd);
    *s = (float)sd;
    *c = (float)cd;
}
RT_HD float rt_sin(float x) { float s, c; rt_sincos(x, &s, &c); return s; }
RT_HD float rt_cos(float x) { float s, c; rt_sincos(x, &s, &c); return c; }

/* acos on [-1,1]; NaN outside.  Rational approximation of asin in binary64. */
RT_HD double rt__asin_r(double z)
{
    const double p = z * (1.66666666666666657415e-01 + z * (-3.25565818622400915405e-01 +
                     z * (2.01212532134862925881e-01 + z * (-4.00555345006794114027e-02 +
                     z * (7.91534994289814532176e-04 + z * 3.47933107596021167570e-05)))));
    const double q = 1.0 + z * (-2.40339491173441421878e+00 + z * (2.02094576023350569471e+00 +
                     z * (-6.88283971605453293030e-01 + z * 7.70381505559019352791e-02)));
    return p / q;
}
RT_HD float rt_acos(float x)
{
    const double xd = (double)x;
    const double a = xd < 0.0 ? -xd : xd;
    const double pio2 = 1.57079632679489655800e+00;
    if (!(a <= 1.0)) return rt_nan();
    if (a <= 0.5) {
        const double z = xd * xd;
        return (float)(pio2 - (xd + xd * rt__asin_r(z)));
    }
    const double z = (1.0 - a) * 0.5;
    const double s = __builtin_sqrt(z);
    const double w = 2.0 * (s + s * rt__asin_r(z));   /* acos(|x|) */
    return (float)(xd > 0.0 ? w : (2.0 * pio2 - w));
}

/* log2 of a positive finite normal binary64. */
RT_HD double rt__log2_64(double x)
{
    uint64_t b = rt_d2u_bits(x);
    int e = (int)((b >> 52) & 0x7FF) - 1023;
    double m = rt_u2d_bits((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull); /* [1,2) */
    if (m > 1.41421356237309514547e+00) { m = m * 0.5; e += 1; }
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double poly = 1.0 / 3.0 + z * (1.0 / 5.0 + z * (1.0 / 7.0 + z * (1.0 / 9.0 + z * (1.0 / 11.0 +
                        z * (1.0 / 13.0 + z * (1.0 / 15.0 + z * (1.0 / 17.0 + z * (1.0 / 19.0))))))));
    const double ln_m = 2.0 * s + 2.0 * s * z * poly;
    return (double)e + ln_m * 1.44269504088896338700e+00;
}

/* 2^t in binary64 for any t (saturating to 0 / +inf). */
RT_HD double rt__exp2_64(double t)
{
    if (t != t) return t;
    if (t >= 1024.0) return rt_u2d_bits(0x7FF0000000000000ull);
    if (t <= -1000.0) return 0.0;
    const double n = rt__rint(t);
    const double u = (t - n) * 6.93147180559945286227e-01;
    double p = 1.0 / 6227020800.0;                 /* 1/13! */
    p = p * u + 1.0 / 479001600.0;
    p = p * u + 1.0 / 39916800.0;
    p = p * u + 1.0 / 3628800.0;
    p = p * u + 1.0 / 362880.0;
    p = p * u + 1.0 / 40320.0;
    p = p * u + 1.0 / 5040.0;
    p = p * u + 1.0 / 720.0;
    p = p * u + 1.0 / 120.0;
    p = p * u + 1.0 / 24.0;
    p = p * u + 1.0 / 6.0;
    p = p * u + 0.5;
    p = p * u + 1.0;
    p = p * u + 1.0;
    const long long ni = (long long)n;             /* in [-1000, 1023] */
    const double scale = rt_u2d_bits((uint64_t)(ni + 1023) << 52);
    return p * scale;
}

/* HLSL pow(x,y) = exp2(y * log2(x)); x < 0 -> NaN; x == 0 -> log2 = -inf. */
RT_HD float rt_pow(float x, float y)
{
    if (x != x || y != y) return rt_nan();
    if (x < 0.0f) return rt_nan();
    double l;
    if (x == 0.0f) l = rt_u2d_bits(0xFFF0000000000000ull);          /* -inf */
    else if (x == rt_inf()) l = rt_u2d_bits(0x7FF0000000000000ull); /* +inf */
    else l = rt__log2_64((double)x);
    const double t = (double)y * l;  /* 0 * inf -> NaN, as IEEE */
    return (float)rt__exp2_64(t);
}

/* --------------------------------------------------------------- vectors */
struct rt_f2 { float x, y; };
struct rt_f3 { float x, y, z; };

RT_HD rt_f3 rt_mk3(float x, float y, float z) { rt_f3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_HD rt_f3 rt_splat3(float v) { return rt_mk3(v, v, v); }
RT_HD rt_f3 rt_add3(rt_f3 a, rt_f3 b) { return rt_mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD rt_f3 rt_sub3(rt_f3 a, rt_f3 b) { return rt_mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD rt_f3 rt_mul3(rt_f3 a, rt_f3 b) { return rt_mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_HD rt_f3 rt_div3(rt_f3 a, rt_f3 b) { return rt_mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
RT_HD rt_f3 rt_scale3(rt_f3 a, float s) { return rt_mk3(a.x * s, a.y * s, a.z * s); }
RT_HD rt_f3 rt_divs3(rt_f3 a, float s) { return rt_mk3(a.x / s, a.y / s, a.z / s); }
RT_HD rt_f3 rt_neg3(rt_f3 a) { return rt_mk3(-a.x, -a.y, -a.z); }
RT_HD float rt_dot3(rt_f3 a, rt_f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
RT_HD rt_f3 rt_cross3(rt_f3 a, rt_f3 b)
{
    return rt_mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
RT_HD float rt_length3(rt_f3 a) { return rt_sqrt(rt_dot3(a, a)); }
RT_HD rt_f3 rt_normalize3(rt_f3 a) { return rt_divs3(a, rt_sqrt(rt_dot3(a, a))); }
RT_HD rt_f3 rt_lerp3(rt_f3 a, rt_f3 b, float t)
{
    return rt_mk3(rt_lerp(a.x, b.x, t), rt_lerp(a.y, b.y, t), rt_lerp(a.z, b.z, t));
}
RT_HD rt_f3 rt_lerp3v(rt_f3 a, rt_f3 b, rt_f3 t)
{
    return rt_mk3(rt_lerp(a.x, b.x, t.x), rt_lerp(a.y, b.y, t.y), rt_lerp(a.z, b.z, t.z));
}
RT_HD rt_f3 rt_saturate3(rt_f3 a) { return rt_mk3(rt_saturate(a.x), rt_saturate(a.y), rt_saturate(a.z)); }
RT_HD bool rt_any3(rt_f3 a) { return a.x != 0.0f || a.y != 0.0f || a.z != 0.0f; }
/* HLSL reflect(i,n) = i - 2*dot(i,n)*n */
RT_HD rt_f3 rt_reflect3(rt_f3 i, rt_f3 n)
{
    const float d2 = 2.0f * rt_dot3(i, n);
    return rt_sub3(i, rt_scale3(n, d2));
}
/* HLSL refract(i,n,eta): k = 1 - eta^2 (1 - dot(n,i)^2); k<0 -> 0 */
RT_HD rt_f3 rt_refract3(rt_f3 i, rt_f3 n, float eta)
{
    const float ni = rt_dot3(n, i);
    const float k = 1.0f - eta * eta * (1.0f - ni * ni);
    if (k < 0.0f) return rt_splat3(0.0f);
    const float a = eta * ni + rt_sqrt(k);
    return rt_sub3(rt_scale3(i, eta), rt_scale3(n, a));
}

/* ------------------------------------------------------------------- RNG */
/* Header.cginc:61-96.  State = {rngstate u32, seed f32}; pixel is the GLOBAL
 * pixel coordinate as two floats. */
struct rt_rng {
    uint32_t state;
    float seed;
    float px, py;
};

RT_HD uint32_t rt_xorshift(uint32_t s)
{
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    return s;
}

/* SetSeed, Header.cginc:79-83 */
RT_HD void rt_rng_init(rt_rng *r, uint32_t px, uint32_t py, float seed)
{
    r->px = (float)px;
    r->py = (float)py;
    r->seed = seed;
    uint32_t s = rt_ftou(r->px * r->py);
    s = rt_xorshift(s);
    s = rt_xorshift(s);
    s = rt_xorshift(s);
    r->state = s;
}

/* Rand, Header.cginc:86-92 with RandFloat :74-77 (float(1<<32 - 5) == 2^27). */
RT_HD float rt_rand(rt_rng *r)
{
    const float d = r->px * 12.9898f + r->py * 78.233f;
    const float result = rt_sin(r->seed / 100.0f * d) * 43758.5453f;
    r->seed = r->seed + 1.0f;
    r->state = rt_ftou((float)r->state + result);
    r->state = rt_xorshift(r->state);
    return rt_frac((float)r->state / 134217728.0f);
}

/* ------------------------------------------------------------ colour space */
/* Header.cginc:183-211 */
RT_HD float rt_srgb_to_linear1(float c)
{
    c = rt_clamp(c, 0.0f, 1.0f);
    const float a = rt_pow((c + 0.055f) / 1.055f, 2.4f);
    const float b = c / 12.92f;
    const float t = (c < 0.04045f) ? 1.0f : 0.0f;
    return rt_lerp(a, b, t);
}
RT_HD float rt_linear_to_srgb1(float c)
{
    c = rt_clamp(c, 0.0f, 1.0f);
    const float a = rt_pow(c, 1.0f / 2.4f) * 1.055f - 0.055f;
    const float b = c * 12.92f;
    const float t = (c < 0.0031308f) ? 1.0f : 0.0f;
    return rt_lerp(a, b, t);
}

#endif /* RT_MATH_H_ */
