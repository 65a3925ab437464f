/* rt_q16.h -- 16-bit box grid shared by the BVH build, the traversal kernels and
 * the CPU checker.
 *
 * The traversal tree stores every child box as six 16-bit grid coordinates
 * relative to the root box.  A stored box always CONTAINS the exact float box
 * it stands for (low planes rounded down, high planes rounded up, checked with
 * the very expression the traversal evaluates), so a ray that passes the exact
 * box also passes the stored one.  Exact boxes are kept only where the
 * reference's result depends on them: at the triangles.
 *
 * All functions are plain IEEE binary32 operations in a fixed order; both
 * compilers build with -ffp-contract=off, so host and device agree bit for bit.
 */
#ifndef RT_Q16_H_
#define RT_Q16_H_

#include "rt_math.h"

#define RT_Q16_MAX 65535u

typedef struct RtQGrid {
    float org[3];    /* root box minimum */
    float step[3];   /* grid pitch per axis; org + 65535 * step >= root box maximum */
} RtQGrid;

/* the one expression that turns a grid coordinate back into a plane position */
RT_HD float rt_q16_plane(float org, float step, uint32_t q) { return org + (float)q * step; }

/* largest q with plane(q) <= v   (v >= org) */
RT_HD uint32_t rt_q16_floor(float org, float step, float v)
{
    if (!(step > 0.0f)) return 0u;               /* flat axis: every plane is org */
    const float g = (v - org) / step;
    uint32_t q = g >= (float)RT_Q16_MAX ? RT_Q16_MAX : (g > 0.0f ? (uint32_t)g : 0u);
    while (q > 0u && rt_q16_plane(org, step, q) > v) --q;
    while (q < RT_Q16_MAX && rt_q16_plane(org, step, q + 1u) <= v) ++q;
    return q;
}

/* smallest q with plane(q) >= v; RT_Q16_MAX + 1 if there is none (grid does not cover v) */
RT_HD uint32_t rt_q16_ceil(float org, float step, float v)
{
    uint32_t q = rt_q16_floor(org, step, v);
    while (q <= RT_Q16_MAX && rt_q16_plane(org, step, q) < v) ++q;
    return q;
}

/* Grid over the box [mn, mx]: pitch = extent / 65535, nudged up until the last plane covers mx. */
RT_HD RtQGrid rt_q16_grid(const float mn[3], const float mx[3])
{
    RtQGrid g;
    for (int a = 0; a < 3; ++a) {
        g.org[a] = mn[a];
        float s = (mx[a] - mn[a]) / (float)RT_Q16_MAX;
        if (!(s > 0.0f)) s = 0.0f;
        for (int k = 0; k < 64 && rt_q16_plane(mn[a], s, RT_Q16_MAX) < mx[a]; ++k)
            s = s > 0.0f ? s * 1.0000002f : 1.0e-30f;
        g.step[a] = s;
    }
    return g;
}

/* Grid box (six 16-bit coordinates in three words, see RtQNode) -> plane positions. */
RT_HD void rt_q16_unpack(const RtQGrid &g, uint32_t w0, uint32_t w1, uint32_t w2, float lo[3], float hi[3])
{
    lo[0] = rt_q16_plane(g.org[0], g.step[0], w0 & 0xFFFFu);
    lo[1] = rt_q16_plane(g.org[1], g.step[1], w0 >> 16);
    lo[2] = rt_q16_plane(g.org[2], g.step[2], w1 & 0xFFFFu);
    hi[0] = rt_q16_plane(g.org[0], g.step[0], w1 >> 16);
    hi[1] = rt_q16_plane(g.org[1], g.step[1], w2 & 0xFFFFu);
    hi[2] = rt_q16_plane(g.org[2], g.step[2], w2 >> 16);
}

/* Per-ray constants of the grid-box test.  For a plane with grid coordinate q the
 * reference's slab value is  t* = RN(RN(plane(q) - o) / d);  the walker evaluates
 * t~ = fma(q, A, B)  with  A = RN(step * r),  B = RN(RN(org - o) * r),  r = RN(1/d).
 * With u = 2^-24 and every operation rounding by at most u relative,
 *     |t~ - t*|  <=  u * (4 |q A| + 3 |B| + |org r| + 3 |t|)  <  12 u * (65535 |A| + |B| + |org r|) ,
 * so M = 16 u * (65535 |A| + |B| + |org r|) + eta covers it with a margin.  M is
 * folded into B: the plane a ray meets first gets B - M, the other one B + M
 * (which is which depends only on the sign of A), so the slab interval the walker
 * sees always contains the reference's.  Then "near > far" or "far < 0" can only
 * be true if the reference's test fails too; NaNs make both false ("maybe"). */
typedef struct RtRaySlab {
    float A[3], Blo[3], Bhi[3];
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
        const float M = (65535.0f * __builtin_fabsf(A) + __builtin_fabsf(B) + __builtin_fabsf(g.org[a] * rr[a])) * RT_SLAB_K + RT_SLAB_ETA;
        s.A[a] = A;
        s.Blo[a] = A >= 0.0f ? B - M : B + M;
        s.Bhi[a] = A >= 0.0f ? B + M : B - M;
    }
    return s;
}

/* false ONLY when the reference's test certainly fails for the grid box (w0, w1, w2) */
RT_HD bool rt_slab_maybe(const RtRaySlab &s, uint32_t w0, uint32_t w1, uint32_t w2)
{
    const float t1x = __builtin_fmaf((float)(w0 & 0xFFFFu), s.A[0], s.Blo[0]);
    const float t1y = __builtin_fmaf((float)(w0 >> 16), s.A[1], s.Blo[1]);
    const float t1z = __builtin_fmaf((float)(w1 & 0xFFFFu), s.A[2], s.Blo[2]);
    const float t2x = __builtin_fmaf((float)(w1 >> 16), s.A[0], s.Bhi[0]);
    const float t2y = __builtin_fmaf((float)(w2 & 0xFFFFu), s.A[1], s.Bhi[1]);
    const float t2z = __builtin_fmaf((float)(w2 >> 16), s.A[2], s.Bhi[2]);
    const float tn = __builtin_fmaxf(__builtin_fminf(t1x, t2x), __builtin_fmaxf(__builtin_fminf(t1y, t2y), __builtin_fminf(t1z, t2z)));
    const float tf = __builtin_fminf(__builtin_fmaxf(t1x, t2x), __builtin_fminf(__builtin_fmaxf(t1y, t2y), __builtin_fmaxf(t1z, t2z)));
    const bool sureMiss = tn > tf || tf < 0.0f;
    return !sureMiss;
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
