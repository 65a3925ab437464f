// rt_traverse.h -- BVH traversal core for gfx950 (wave64).
//
// Semantics are those of the reference's BVHRayTrace
// (Assets/Shaders/BVH/BVHRayTracing.cginc:27-133 with IntersectTriangle_MT97,
// Assets/Shaders/RayTracing/BruteForceRayTracing.cginc:166-204): every node
// whose box the ray passes is visited (no distance pruning), a leaf's triangle
// is tested when its own box passes, the closest hit wins with strict '<', so
// on equal distances the triangle met first in the reference's depth-first
// order (left leaf, right leaf, right subtree, left subtree) is kept.
//
// What is different is only HOW the same decisions are computed:
//   * one 64-byte node fetch brings both children's boxes;
//   * the slab test is first evaluated with per-ray reciprocals and an error
//     margin; only when the margin cannot decide (a graze within a few ulp, an
//     infinity, a NaN) are the reference's six true divisions evaluated, so the
//     boolean is always the one the reference computes;
//   * a wave keeps fetching new rays for lanes whose ray has finished
//     (ballot + prefix count + one atomic per refill), and triangle tests are
//     queued per lane in LDS and run in batches, in per-ray order, so that the
//     64 lanes stay busy although rays take very different walks.
#ifndef RT_TRAVERSE_H_
#define RT_TRAVERSE_H_

#include "rt_internal.h"

#define TRV_THREADS 256
#define TRV_PEND_CAP 8          // queued triangle tests per lane
#define TRV_NODE_BURST 48       // node steps between two triangle batches
#define TRV_REFILL_MIN 24       // refill when at least this many lanes are idle
#define TRV_EARLY_OUT 40        // leave a burst when this many lanes have nothing to walk
#define TRV_DONE 0xFFFFFFFFu
#define TRV_LDS_WORDS ((RT_STACK_DEPTH + TRV_PEND_CAP) * TRV_THREADS)

// The reference's slab test, verbatim: true divisions, min/max that return the
// non-NaN operand (v_min_f32 / v_max_f32 in IEEE mode).
__device__ __forceinline__ bool box_hit_exact(float bminx, float bminy, float bminz, float bmaxx, float bmaxy, float bmaxz,
                                              const rt_f3 o, const rt_f3 d)
{
    const float t1x = (bminx - o.x) / d.x, t1y = (bminy - o.y) / d.y, t1z = (bminz - o.z) / d.z;
    const float t2x = (bmaxx - o.x) / d.x, t2y = (bmaxy - o.y) / d.y, t2z = (bmaxz - o.z) / d.z;
    const float tmin = __builtin_fmaxf(__builtin_fminf(t1x, t2x),
                                       __builtin_fmaxf(__builtin_fminf(t1y, t2y), __builtin_fminf(t1z, t2z)));
    const float tmax = __builtin_fminf(__builtin_fmaxf(t1x, t2x),
                                       __builtin_fminf(__builtin_fmaxf(t1y, t2y), __builtin_fmaxf(t1z, t2z)));
    return tmax > tmin && tmax > 0.0f;
}

// Same boolean, cheaper in the common case.  With r = RN(1/d), the product
// RN((b-o)*r) differs from the quotient RN((b-o)/d) by at most 3 * 2^-24
// relative (plus underflow noise far below ETA); min/max are monotone, so the
// same bound holds for tmin and tmax.  EPS = 8 * 2^-24 leaves a 2.6x safety
// factor.  Infinite or NaN intermediates make every comparison below false and
// fall through to the exact test.
#define TRV_EPS 4.76837158e-7f
#define TRV_ETA 1.0e-30f
__device__ __forceinline__ bool box_hit(float bminx, float bminy, float bminz, float bmaxx, float bmaxy, float bmaxz,
                                        const rt_f3 o, const rt_f3 d, const rt_f3 r)
{
    const float t1x = (bminx - o.x) * r.x, t1y = (bminy - o.y) * r.y, t1z = (bminz - o.z) * r.z;
    const float t2x = (bmaxx - o.x) * r.x, t2y = (bmaxy - o.y) * r.y, t2z = (bmaxz - o.z) * r.z;
    const float lo = __builtin_fmaxf(__builtin_fminf(t1x, t2x),
                                     __builtin_fmaxf(__builtin_fminf(t1y, t2y), __builtin_fminf(t1z, t2z)));
    const float hi = __builtin_fminf(__builtin_fmaxf(t1x, t2x),
                                     __builtin_fminf(__builtin_fmaxf(t1y, t2y), __builtin_fmaxf(t1z, t2z)));
    const float margin = (__builtin_fabsf(hi) + __builtin_fabsf(lo)) * TRV_EPS + 2.0f * TRV_ETA;
    const float diff = hi - lo;
    const bool sureHit = diff > margin && hi > 2.0f * TRV_ETA;
    const bool sureMiss = (-diff) > margin || hi < -2.0f * TRV_ETA;
    if (__builtin_expect(sureHit || sureMiss, 1)) return sureHit;
    return box_hit_exact(bminx, bminy, bminz, bmaxx, bmaxy, bmaxz, o, d);
}

// IntersectTriangle_MT97 + the acceptance test of CheckTriangle, on the stored
// (v0, e1 = v1 - v0, e2 = v2 - v0) record of sorted leaf `leaf`.
__device__ __forceinline__ void tri_test(const RtSceneView &sv, uint32_t leaf, const rt_f3 o, const rt_f3 d, RtHitRec &h)
{
    const float4 *p = (const float4 *)(sv.leafTri + leaf);
    const float4 a = p[0], b = p[1], c = p[2];
    const rt_f3 v0 = rt_mk3(a.x, a.y, a.z), e1 = rt_mk3(a.w, b.x, b.y), e2 = rt_mk3(b.z, b.w, c.x);
    const rt_f3 pvec = rt_cross3(d, e2);
    const float det = rt_dot3(e1, pvec);
    if (det < RT_EPSILON) return;
    const float inv_det = 1.0f / det;
    const rt_f3 tvec = rt_sub3(o, v0);
    const float u = rt_dot3(tvec, pvec) * inv_det;
    if (u < 0.0f || u > 1.0f) return;
    const rt_f3 qvec = rt_cross3(tvec, e1);
    const float v = rt_dot3(d, qvec) * inv_det;
    if (v < 0.0f || u + v > 1.0f) return;
    const float t = rt_dot3(e2, qvec) * inv_det;
    if (t > 0.0f && t < h.t) {
        h.t = t;
        h.u = u;
        h.v = v;
        h.leaf = leaf;
    }
}

__device__ __forceinline__ void flush_counters(const RtSceneView &sv, uint32_t rays, const uint32_t cnt[3], uint32_t hits)
{
    uint32_t v[5] = {rays, cnt[0], cnt[1], cnt[2], hits};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        uint32_t x = v[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if ((threadIdx.x & 63u) == 0 && x) atomicAdd(&sv.stats[k], (unsigned long long)x);
    }
}

// Persistent traversal of `src.count()` rays by one wave (all waves of all
// blocks share *fetchCounter, which must be 0 at launch).
//   Src: uint32_t count(); void load(uint32_t i, rt_f3 &o, rt_f3 &d); void store(uint32_t i, const RtHitRec &h);
// lds: TRV_LDS_WORDS words per block; this lane's stack entries are
// lds[level * TRV_THREADS + tid], its pending triangles lds[(RT_STACK_DEPTH + k) * TRV_THREADS + tid].
template <bool COUNT, class Src>
__device__ __forceinline__ void traverse_rays(const RtSceneView &sv, Src &src, uint32_t *__restrict__ fetchCounter, uint32_t *lds)
{
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t *stk = lds + threadIdx.x;
    uint32_t *pend = lds + RT_STACK_DEPTH * TRV_THREADS + threadIdx.x;
    const uint32_t total = src.count();

    bool hasRay = false;
    uint32_t rayIdx = 0, node = TRV_DONE, sp = 0, npend = 0;
    rt_f3 o = rt_splat3(0.0f), d = rt_splat3(0.0f), r = rt_splat3(0.0f);
    RtHitRec h;
    h.t = rt_inf(); h.u = 0.0f; h.v = 0.0f; h.leaf = RT_INVALID;
    bool exhausted = false;
    uint32_t cnt[3] = {0, 0, 0};
    uint32_t nRays = 0, nHits = 0;

    for (;;) {
        // ---- refill idle lanes: ballot, prefix count, one atomic per wave
        const unsigned long long idle = __ballot(!hasRay);
        const uint32_t nIdle = (uint32_t)__popcll(idle);
        if (!exhausted && nIdle >= TRV_REFILL_MIN) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(fetchCounter, nIdle);
            base = __shfl(base, 0, 64);
            if (base + nIdle >= total) exhausted = true;
            const uint32_t mine = base + (uint32_t)__popcll(idle & lt);
            if (!hasRay && mine < total) {
                rayIdx = mine;
                src.load(mine, o, d);
                r = rt_mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                h.t = rt_inf(); h.u = 0.0f; h.v = 0.0f; h.leaf = RT_INVALID;
                hasRay = true;
                sp = 0;
                npend = 0;
                node = TRV_DONE;
                if (COUNT) nRays++;
                if (sv.nTriangles != 0) {
                    if (COUNT) cnt[0]++;
                    const RtAABB &rb = sv.rootBox;
                    if (sv.nTriangles >= 2 && box_hit(rb.min[0], rb.min[1], rb.min[2], rb.max[0], rb.max[1], rb.max[2], o, d, r))
                        node = 0;
                }
            }
        }
        if (__ballot(hasRay) == 0ull) break;

        // ---- node phase: walk the tree, queue triangles whose box passes
        for (int it = 0; it < TRV_NODE_BURST; ++it) {
            const bool act = hasRay && node != TRV_DONE;
            const unsigned long long am = __ballot(act);
            if (am == 0ull) break;
            if (act) {
                const float4 *np = (const float4 *)(sv.nodes + node);
                const float4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
                const uint32_t l = __float_as_uint(q3.x), rr = __float_as_uint(q3.y);
                const bool lLeaf = (l & RT_LEAF_BIT) != 0, rLeaf = (rr & RT_LEAF_BIT) != 0;
                const bool hitL = box_hit(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, o, d, r);
                const bool hitR = box_hit(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, o, d, r);
                if (COUNT) {
                    cnt[0] += (lLeaf ? 0u : 1u) + (rLeaf ? 0u : 1u);
                    cnt[1] += (lLeaf ? 1u : 0u) + (rLeaf ? 1u : 0u);
                    cnt[2] += ((lLeaf && hitL) ? 1u : 0u) + ((rLeaf && hitR) ? 1u : 0u);
                }
                if (lLeaf && hitL) { pend[npend * TRV_THREADS] = l & ~RT_LEAF_BIT; npend++; }
                if (rLeaf && hitR) { pend[npend * TRV_THREADS] = rr & ~RT_LEAF_BIT; npend++; }
                const bool goL = !lLeaf && hitL, goR = !rLeaf && hitR;
                if (goR) {
                    if (goL) {
                        if (sp >= RT_STACK_DEPTH) { *sv.errorFlag = 1u; node = TRV_DONE; }
                        else { stk[sp * TRV_THREADS] = l; sp++; node = rr; }
                    } else {
                        node = rr;
                    }
                } else if (goL) {
                    node = l;
                } else if (sp == 0) {
                    node = TRV_DONE;
                } else {
                    sp--;
                    node = stk[sp * TRV_THREADS];
                }
            }
            // a node step can queue two triangles: stop while there is still room for two
            if (__ballot(npend > TRV_PEND_CAP - 2) != 0ull) break;
            // many lanes finished: leave early so that they can be refilled
            if (!exhausted && (uint32_t)__popcll(__ballot(!hasRay || node == TRV_DONE)) >= (uint32_t)TRV_EARLY_OUT) break;
        }

        // ---- triangle phase: every lane tests its queued triangles in the order it met them
        for (uint32_t k = 0; __ballot(k < npend) != 0ull; ++k)
            if (k < npend) tri_test(sv, pend[k * TRV_THREADS], o, d, h);
        npend = 0;

        // ---- retire finished rays
        if (hasRay && node == TRV_DONE) {
            src.store(rayIdx, h);
            if (COUNT && h.leaf != RT_INVALID) nHits++;
            hasRay = false;
        }
    }
    if (COUNT) flush_counters(sv, nRays, cnt, nHits);
}

#endif
