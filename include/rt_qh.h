/* rt_qh.h -- half-float box planes of the traversal tree, shared by the BVH build, the
 * traversal kernels and the CPU checker.
 *
 * The tree the rays walk stores every node's box as six binary16 numbers h: plane
 * positions  org + value(h) * step  per axis, org = centre of the scene's box, step chosen so
 * that the box's half extent lands near 32768.  A stored box always CONTAINS the exact float
 * box it stands for (low planes rounded down, high planes rounded up, checked with the
 * expression below), so a ray that passes the exact box also passes the stored one.  Exact
 * boxes are kept only where the reference's result depends on them: at the triangles.
 *
 * Why binary16 and not a uniform 16-bit grid: gfx950 multiplies a binary16 operand inside a
 * binary32 fused multiply-add (v_fma_mix_f32), so a plane's slab value costs ONE vector
 * instruction instead of a conversion and a multiply-add.  The grid is finest at the centre
 * (where a scene usually has its detail) and has 11 significant bits everywhere: at the rim
 * of the scene a plane is off by at most 1/4096 of the scene's extent.
 *
 * All functions are plain IEEE operations in a fixed order (value(h) is exact in binary32;
 * the fused multiply-add is the correctly rounded one on both sides), so host and device
 * agree bit for bit.
 */
#ifndef RT_QH_H_
#define RT_QH_H_

#include "rt_math.h"

#define RT_QH_MAX_KEY 31743          /* 0x7BFF: the largest finite binary16 */
#define RT_QH_MAX_VALUE 65504.0f
#define RT_QH_SPAN 32768.0f          /* the half extent of the scene's box in grid units */

typedef struct RtQGrid {
    float org[3];    /* centre of the root box */
    float step[3];   /* world units per grid unit; 0 on a flat axis */
} RtQGrid;

/* Finite binary16 numbers in increasing order: key in [-31743, 31743], key 0 = +0. */
RT_HD uint32_t rt_qh_bits(int key) { return key >= 0 ? (uint32_t)key : (0x8000u | (uint32_t)(-key)); }
RT_HD int rt_qh_key(uint32_t bits) { return (bits & 0x8000u) ? -(int)(bits & 0x7FFFu) : (int)(bits & 0x7FFFu); }

/* value of a finite binary16 bit pattern, exact */
RT_HD float rt_qh_value(uint32_t bits)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (float)__builtin_bit_cast(_Float16, (unsigned short)bits);
#else
    const uint32_t e = (bits >> 10) & 31u, m = bits & 1023u;
    /* subnormal: m * 2^-24; normal: (1024 + m) * 2^(e - 25); both exact in binary32 */
    float v = e == 0u ? (float)m * 5.9604644775390625e-8f : rt_u2f_bits(((e + 112u) << 23) | (m << 13));
    return (bits & 0x8000u) ? -v : v;
#endif
}

/* the one expression that turns a stored number back into a plane position */
RT_HD float rt_qh_plane(float org, float step, int key) { return org + rt_qh_value(rt_qh_bits(key)) * step; }

/* a key near (v - org) / step: binary32 -> binary16 by truncation; the callers walk from there */
RT_HD int rt_qh_guess(float g)
{
    const float a = __builtin_fabsf(g);
    int key;
    if (!(a < RT_QH_MAX_VALUE)) key = RT_QH_MAX_KEY;                    /* also NaN */
    else if (a < 6.103515625e-5f) key = (int)(a * 16777216.0f);         /* subnormal range: units of 2^-24 */
    else key = (int)((rt_f2u_bits(a) >> 13) - (112u << 10));
    return g < 0.0f ? -key : key;
}

/* largest key with plane(key) <= v; -RT_QH_MAX_KEY - 1 if there is none */
RT_HD int rt_qh_floor(float org, float step, float v)
{
    if (!(step > 0.0f)) return org <= v ? 0 : -RT_QH_MAX_KEY - 1;       /* flat axis: every plane is org */
    int k = rt_qh_guess((v - org) / step);
    while (k >= -RT_QH_MAX_KEY && rt_qh_plane(org, step, k) > v) --k;
    while (k < RT_QH_MAX_KEY && rt_qh_plane(org, step, k + 1) <= v) ++k;
    return k;
}

/* smallest key with plane(key) >= v; RT_QH_MAX_KEY + 1 if there is none */
RT_HD int rt_qh_ceil(float org, float step, float v)
{
    if (!(step > 0.0f)) return org >= v ? 0 : RT_QH_MAX_KEY + 1;
    int k = rt_qh_guess((v - org) / step);
    while (k <= RT_QH_MAX_KEY && rt_qh_plane(org, step, k) < v) ++k;
    while (k > -RT_QH_MAX_KEY && rt_qh_plane(org, step, k - 1) >= v) --k;
    return k;
}

/* Grid of the scene box [mn, mx]: centre and the pitch that puts the farther face near RT_QH_SPAN
 * (at most twice that after rounding, well inside the binary16 range). */
RT_HD RtQGrid rt_qh_grid(const float mn[3], const float mx[3])
{
    RtQGrid g;
    for (int a = 0; a < 3; ++a) {
        const float c = (mn[a] + mx[a]) * 0.5f;
        const float half = rt_max(mx[a] - c, c - mn[a]);
        float s = half / RT_QH_SPAN;
        if (!(s > 0.0f)) s = 0.0f;
        g.org[a] = c;
        g.step[a] = s;
    }
    return g;
}

/* A box on the grid: three words, one per axis, low plane in bits 0..15, high plane in bits 16..31.
 * Returns false if a plane of the box lies outside the grid (the planes are clamped then). */
RT_HD bool rt_qh_box(const RtQGrid &g, const float mn[3], const float mx[3], uint32_t out[3])
{
    bool ok = true;
    for (int a = 0; a < 3; ++a) {
        int lo = rt_qh_floor(g.org[a], g.step[a], mn[a]), hi = rt_qh_ceil(g.org[a], g.step[a], mx[a]);
        if (lo < -RT_QH_MAX_KEY) { lo = -RT_QH_MAX_KEY; ok = false; }
        if (hi > RT_QH_MAX_KEY) { hi = RT_QH_MAX_KEY; ok = false; }
        ok = ok && rt_qh_plane(g.org[a], g.step[a], lo) <= mn[a] && rt_qh_plane(g.org[a], g.step[a], hi) >= mx[a];
        out[a] = rt_qh_bits(lo) | (rt_qh_bits(hi) << 16);
    }
    return ok;
}

RT_HD void rt_qh_unpack(const RtQGrid &g, const uint32_t w[3], float lo[3], float hi[3])
{
    for (int a = 0; a < 3; ++a) {
        lo[a] = g.org[a] + rt_qh_value(w[a] & 0xFFFFu) * g.step[a];
        hi[a] = g.org[a] + rt_qh_value(w[a] >> 16) * g.step[a];
    }
}

/* Per-ray constants of the grid-box test.  For a plane stored as h the reference's slab value
 * is  t* = RN(RN(plane(h) - o) / d);  the walker evaluates  t~ = fma(value(h), A, B)  with
 * A = RN(step * r),  B = RN(RN(org - o) * r),  r = RN(1/d).  With u = 2^-24 and every
 * operation rounding by at most u relative,
 *     |t~ - t*|  <=  u * (4 |h A| + 3 |B| + |org r| + 3 |t|)  <  12 u * (65504 |A| + |B| + |org r|) ,
 * so M = 16 u * (65504 |A| + |B| + |org r|) + eta covers it with a margin.  M is folded into
 * B: the plane a ray meets first gets B - M, the other one B + M (which is which depends only
 * on the sign of A), so the slab interval the walker sees always contains the reference's.
 * Then "near > far" or "far < 0" can only be true if the reference's test fails too. */
typedef struct RtRaySlab {
    float A[3], Blo[3], Bhi[3];      /* Blo goes with a box's low plane, Bhi with its high plane */
} RtRaySlab;

#define RT_SLAB_K 9.5367431640625e-7f   /* 16 * 2^-24 */
#define RT_SLAB_ETA 1.0e-30f

RT_HD RtRaySlab rt_ray_slab(const RtQGrid &g, const rt_f3 o, const rt_f3 r)
{
    RtRaySlab s;
    const float oo[3] = {o.x, o.y, o.z}, rr[3] = {r.x, r.y, r.z};
    for (int a = 0; a < 3; ++a) {
        const float A = g.step[a] * rr[a];
        const float B = (g.org[a] - oo[a]) * rr[a];
        const float M = (RT_QH_MAX_VALUE * __builtin_fabsf(A) + __builtin_fabsf(B) + __builtin_fabsf(g.org[a] * rr[a])) * RT_SLAB_K + RT_SLAB_ETA;
        s.A[a] = A;
        s.Blo[a] = A >= 0.0f ? B - M : B + M;
        s.Bhi[a] = A >= 0.0f ? B + M : B - M;
    }
    return s;
}

/* Which way the ray runs along each axis, as the slab constants see it: bit a set = A[a] < 0
 * (the box's HIGH plane is the one the ray meets first on that axis). */
RT_HD uint32_t rt_slab_octant(const RtRaySlab &s)
{
    return (s.A[0] < 0.0f ? 1u : 0u) | (s.A[1] < 0.0f ? 2u : 0u) | (s.A[2] < 0.0f ? 4u : 0u);
}

/* All nine constants finite and |A| small enough that no slab value can overflow: then every slab
 * value is finite, a low plane's value never exceeds the high plane's for A >= 0 (fma is monotone in
 * each operand) and the other way round for A < 0, and the test may pick near and far planes by the
 * sign of A instead of by min / max -- the same booleans either way.  Rays that fail this are
 * walked like those with a zero or non-finite direction component. */
RT_HD bool rt_slab_finite(const RtRaySlab &s)
{
    const float big = 1.0e30f;
    bool ok = true;
    for (int a = 0; a < 3; ++a)
        ok = ok && __builtin_fabsf(s.A[a]) < big && __builtin_fabsf(s.Blo[a]) < big && __builtin_fabsf(s.Bhi[a]) < big;
    return ok;
}

#define RT_QH_LO(w) rt_qh_value((w) & 0xFFFFu)
#define RT_QH_HI(w) rt_qh_value((w) >> 16)

/* false ONLY when the reference's test certainly fails for the grid box (wx, wy, wz); *tnear: where the ray
 * enters the box at the earliest (a lower bound, for ordering only) */
RT_HD bool rt_slab_maybe_near(const RtRaySlab &s, uint32_t wx, uint32_t wy, uint32_t wz, float *tnear)
{
    const float t1x = __builtin_fmaf(RT_QH_LO(wx), s.A[0], s.Blo[0]);
    const float t1y = __builtin_fmaf(RT_QH_LO(wy), s.A[1], s.Blo[1]);
    const float t1z = __builtin_fmaf(RT_QH_LO(wz), s.A[2], s.Blo[2]);
    const float t2x = __builtin_fmaf(RT_QH_HI(wx), s.A[0], s.Bhi[0]);
    const float t2y = __builtin_fmaf(RT_QH_HI(wy), s.A[1], s.Bhi[1]);
    const float t2z = __builtin_fmaf(RT_QH_HI(wz), s.A[2], s.Bhi[2]);
    const float tn = __builtin_fmaxf(__builtin_fminf(t1x, t2x), __builtin_fmaxf(__builtin_fminf(t1y, t2y), __builtin_fminf(t1z, t2z)));
    const float tf = __builtin_fminf(__builtin_fmaxf(t1x, t2x), __builtin_fminf(__builtin_fmaxf(t1y, t2y), __builtin_fmaxf(t1z, t2z)));
    const bool sureMiss = tn > tf || tf < 0.0f;
    *tnear = tn;
    return !sureMiss;
}

RT_HD bool rt_slab_maybe(const RtRaySlab &s, uint32_t wx, uint32_t wy, uint32_t wz)
{
    float tn;
    return rt_slab_maybe_near(s, wx, wy, wz, &tn);
}

#ifdef __cplusplus
/* The same boolean for a ray (rt_slab_finite) whose octant is OCT: near and far planes are picked
 * by the sign of A, no min / max per axis. */
template <uint32_t OCT>
RT_HD bool rt_slab_maybe_octant_near(const RtRaySlab &s, uint32_t wx, uint32_t wy, uint32_t wz, float *tnear)
{
    const float nx = (OCT & 1u) ? __builtin_fmaf(RT_QH_HI(wx), s.A[0], s.Bhi[0]) : __builtin_fmaf(RT_QH_LO(wx), s.A[0], s.Blo[0]);
    const float fx = (OCT & 1u) ? __builtin_fmaf(RT_QH_LO(wx), s.A[0], s.Blo[0]) : __builtin_fmaf(RT_QH_HI(wx), s.A[0], s.Bhi[0]);
    const float ny = (OCT & 2u) ? __builtin_fmaf(RT_QH_HI(wy), s.A[1], s.Bhi[1]) : __builtin_fmaf(RT_QH_LO(wy), s.A[1], s.Blo[1]);
    const float fy = (OCT & 2u) ? __builtin_fmaf(RT_QH_LO(wy), s.A[1], s.Blo[1]) : __builtin_fmaf(RT_QH_HI(wy), s.A[1], s.Bhi[1]);
    const float nz = (OCT & 4u) ? __builtin_fmaf(RT_QH_HI(wz), s.A[2], s.Bhi[2]) : __builtin_fmaf(RT_QH_LO(wz), s.A[2], s.Blo[2]);
    const float fz = (OCT & 4u) ? __builtin_fmaf(RT_QH_LO(wz), s.A[2], s.Blo[2]) : __builtin_fmaf(RT_QH_HI(wz), s.A[2], s.Bhi[2]);
    const float tn = __builtin_fmaxf(nx, __builtin_fmaxf(ny, nz));
    const float tf = __builtin_fminf(fx, __builtin_fminf(fy, fz));
    const bool sureMiss = tn > tf || tf < 0.0f;
    *tnear = tn;
    return !sureMiss;
}

/* The same, and *shifted = max over the axes of (near value on the axis - back[axis]).  (For OCT = 8: any ray, near and
 * far picked by min / max.)  A point of the ray within a distance rho of the box (on every axis: within rho of the
 * box's extent) has, on each axis a, a ray parameter of at least near[a] - rho |d| / |d[a]| in units of the ray
 * parameter; with back[a] that amount, *shifted is the smallest parameter at which the ray can be that near the box.
 * With back = 0 it is the entry distance itself.  The near values are lower bounds already (the slab constants carry
 * their margin). */
template <uint32_t OCT>
RT_HD bool rt_slab_maybe_shifted(const RtRaySlab &s, uint32_t wx, uint32_t wy, uint32_t wz, const float back[3], float *tnear, float *shifted)
{
    float nx, ny, nz, fx, fy, fz;
    if (OCT < 8u) {
        nx = (OCT & 1u) ? __builtin_fmaf(RT_QH_HI(wx), s.A[0], s.Bhi[0]) : __builtin_fmaf(RT_QH_LO(wx), s.A[0], s.Blo[0]);
        fx = (OCT & 1u) ? __builtin_fmaf(RT_QH_LO(wx), s.A[0], s.Blo[0]) : __builtin_fmaf(RT_QH_HI(wx), s.A[0], s.Bhi[0]);
        ny = (OCT & 2u) ? __builtin_fmaf(RT_QH_HI(wy), s.A[1], s.Bhi[1]) : __builtin_fmaf(RT_QH_LO(wy), s.A[1], s.Blo[1]);
        fy = (OCT & 2u) ? __builtin_fmaf(RT_QH_LO(wy), s.A[1], s.Blo[1]) : __builtin_fmaf(RT_QH_HI(wy), s.A[1], s.Bhi[1]);
        nz = (OCT & 4u) ? __builtin_fmaf(RT_QH_HI(wz), s.A[2], s.Bhi[2]) : __builtin_fmaf(RT_QH_LO(wz), s.A[2], s.Blo[2]);
        fz = (OCT & 4u) ? __builtin_fmaf(RT_QH_LO(wz), s.A[2], s.Blo[2]) : __builtin_fmaf(RT_QH_HI(wz), s.A[2], s.Bhi[2]);
    } else {
        const float t1x = __builtin_fmaf(RT_QH_LO(wx), s.A[0], s.Blo[0]), t2x = __builtin_fmaf(RT_QH_HI(wx), s.A[0], s.Bhi[0]);
        const float t1y = __builtin_fmaf(RT_QH_LO(wy), s.A[1], s.Blo[1]), t2y = __builtin_fmaf(RT_QH_HI(wy), s.A[1], s.Bhi[1]);
        const float t1z = __builtin_fmaf(RT_QH_LO(wz), s.A[2], s.Blo[2]), t2z = __builtin_fmaf(RT_QH_HI(wz), s.A[2], s.Bhi[2]);
        nx = __builtin_fminf(t1x, t2x); fx = __builtin_fmaxf(t1x, t2x);
        ny = __builtin_fminf(t1y, t2y); fy = __builtin_fmaxf(t1y, t2y);
        nz = __builtin_fminf(t1z, t2z); fz = __builtin_fmaxf(t1z, t2z);
    }
    const float tn = __builtin_fmaxf(nx, __builtin_fmaxf(ny, nz));
    const float tf = __builtin_fminf(fx, __builtin_fminf(fy, fz));
    *tnear = tn;
    *shifted = __builtin_fmaxf(nx - back[0], __builtin_fmaxf(ny - back[1], nz - back[2]));
    return !(tn > tf || tf < 0.0f);
}

template <uint32_t OCT>
RT_HD bool rt_slab_maybe_octant(const RtRaySlab &s, uint32_t wx, uint32_t wy, uint32_t wz)
{
    float tn;
    return rt_slab_maybe_octant_near<OCT>(s, wx, wy, wz, &tn);
}
#endif

/* The box an unused place of a two-level record holds: low plane +65504, high plane -65504 on every axis.  Where
 * near and far planes are picked by the sign of A, the ray's near value on an axis is then
 * fma(+-65504, A, B -+ M) = 65504 |A| + (B - M) or its mirror image and the far value 65504 |A| less on the other
 * side, so near > far -- a sure miss -- exactly when 65504 |A| outweighs the margin M on that axis. */
#define RT_QH_NO_BOX 0xFBFF7BFFu

/* Does the octant test (rt_slab_maybe_octant_near) turn the empty box down for this ray?  Evaluated with that
 * test's own multiply-adds: it rejects a box as soon as one axis has near > far.  It does for every ray whose
 * origin is within about a million scene extents of the scene; a ray for which it does not (M outweighs the grid
 * on all three axes) walks the exact nodes instead, so an unused place is never opened. */
RT_HD bool rt_slab_rejects_empty_box(const RtRaySlab &s)
{
    bool rejects = false;
    for (int a = 0; a < 3; ++a) {
        const float lo = __builtin_fmaf(RT_QH_MAX_VALUE, s.A[a], s.Blo[a]), hi = __builtin_fmaf(-RT_QH_MAX_VALUE, s.A[a], s.Bhi[a]);
        const float tnear = s.A[a] < 0.0f ? hi : lo, tfar = s.A[a] < 0.0f ? lo : hi;
        rejects = rejects || tnear > tfar;
    }
    return rejects;
}

/* Rays for which "a triangle's box passes => every enclosing box passes" is not
 * guaranteed by monotonicity of IEEE subtraction and division: a zero, infinite
 * or NaN direction component, or a non-finite origin.  Such rays are walked with
 * the exact boxes and the reference's own test at every node. */
RT_HD bool rt_ray_needs_exact_walk(const rt_f3 o, const rt_f3 d)
{
    const float big = 3.0e38f;
    const bool dOk = d.x != 0.0f && d.y != 0.0f && d.z != 0.0f && __builtin_fabsf(d.x) < big && __builtin_fabsf(d.y) < big &&
                     __builtin_fabsf(d.z) < big;
    const bool oOk = __builtin_fabsf(o.x) < big && __builtin_fabsf(o.y) < big && __builtin_fabsf(o.z) < big;
    return !(dOk && oOk);
}

#endif
