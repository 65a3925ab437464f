/* rt_finekey.h -- sort keys of the library's own traversal tree, shared by the BVH
 * build and the CPU checker.
 *
 * The reference sorts triangles by a 30-bit Morton code over a FIXED world box of
 * [-125, 125]^3 (MeshData.compute:62-101): cells of 0.24 units, so a 4-unit model
 * spans 16 cells per axis and thousands of triangles share a key.  Its tree is
 * reproduced bit for bit for readback, but walking it is wasteful.  A ray's result
 * depends only on WHICH triangles pass their own box (any tree over the same boxes
 * finds them) and, among equal distances, on the reference's visiting order (kept as
 * a per-triangle rank).  So rays walk a second tree built by the same pipeline
 * (stable radix sort, duplicate-key spreading, Karras tree, bottom-up fit) from the
 * keys below: 29 bits laid over the box of the scene itself, each next bit halving
 * whichever axis currently has the widest cell, so cells come out near-cubic
 * whatever the proportions of the scene -- and, above them, 2 bits for the SIZE class
 * of the triangle's box.  A Morton order places a triangle by its centre only: a
 * ground quad 28 units wide lands deep inside the subtree of the 5-millimetre triangles
 * around its centre and blows up every box on the way to the root (measured on the
 * car scene: 21 of the 45 box tests of a ray leaving the ground).  With the size class
 * on top, the few large triangles form subtrees of their own right under the root.
 *
 * Plain IEEE binary32 in a fixed order; identical on host and device.
 */
#ifndef RT_FINEKEY_H_
#define RT_FINEKEY_H_

#include "rt_math.h"

#define RT_FINEKEY_BITS 29   /* position bits; with the 2 size bits keys stay below 2^31: spreading duplicates
                              * (+1 per triangle at most, fewer than 2^31 triangles) cannot wrap */
#define RT_FINEKEY_TIERS 4   /* size classes: largest box extent > E/4, > E/16, > E/64, the rest (E = largest scene extent) */

typedef struct RtKeyPlan {
    float org[3];            /* minimum corner of the scene box */
    float scale[3];          /* cells per unit: 2^bits / extent (0 for a flat axis) */
    uint32_t bits[3];        /* bits per axis, sum = RT_FINEKEY_BITS */
    uint8_t axis[32];        /* axis of key bit k, most significant first */
    float tierLimit[3];      /* a box whose largest extent exceeds tierLimit[k] is of size class <= k */
} RtKeyPlan;

RT_HD RtKeyPlan rt_key_plan(const float mn[3], const float mx[3])
{
    RtKeyPlan p;
    float cell[3];
    for (int a = 0; a < 3; ++a) {
        p.org[a] = mn[a];
        p.bits[a] = 0;
        cell[a] = mx[a] - mn[a];
        if (!(cell[a] > 0.0f)) cell[a] = 0.0f;
    }
    for (int k = 0; k < 32; ++k) p.axis[k] = 0;
    for (int k = 0; k < RT_FINEKEY_BITS; ++k) {
        int a = 0;
        for (int c = 1; c < 3; ++c)
            if (cell[c] > cell[a] || (cell[c] == cell[a] && p.bits[c] < p.bits[a])) a = c;
        p.axis[k] = (uint8_t)a;
        p.bits[a] += 1;
        cell[a] *= 0.5f;
    }
    float largest = 0.0f;
    for (int a = 0; a < 3; ++a) {
        const float ext = mx[a] - mn[a];
        p.scale[a] = ext > 0.0f ? (float)(1u << p.bits[a]) / ext : 0.0f;
        if (ext > largest) largest = ext;
    }
    p.tierLimit[0] = largest * 0.25f;
    p.tierLimit[1] = largest * 0.0625f;
    p.tierLimit[2] = largest * 0.015625f;
    return p;
}

/* key of a triangle box [mn, mx] inside (or clamped into) the scene box: size class, then the cell of its centre */
RT_HD uint32_t rt_fine_key(const RtKeyPlan &p, const float mn[3], const float mx[3])
{
    const float c[3] = {(mn[0] + mx[0]) * 0.5f, (mn[1] + mx[1]) * 0.5f, (mn[2] + mx[2]) * 0.5f};
    const float size = rt_max(rt_max(mx[0] - mn[0], mx[1] - mn[1]), mx[2] - mn[2]);
    const uint32_t tier = size > p.tierLimit[0] ? 0u : (size > p.tierLimit[1] ? 1u : (size > p.tierLimit[2] ? 2u : 3u));
    uint32_t q[3], rem[3];
    for (int a = 0; a < 3; ++a) {
        const uint32_t top = (1u << p.bits[a]) - 1u;
        const uint32_t v = rt_ftou((c[a] - p.org[a]) * p.scale[a]);     /* negative and NaN -> 0 */
        q[a] = v > top ? top : v;
        rem[a] = p.bits[a];
    }
    uint32_t key = 0;
    for (int k = 0; k < RT_FINEKEY_BITS; ++k) {
        const uint32_t a = p.axis[k];
        rem[a] -= 1u;
        key = (key << 1) | ((q[a] >> rem[a]) & 1u);
    }
    return ((3u - tier) << RT_FINEKEY_BITS) | key;      /* the many small triangles first */
}

#endif
