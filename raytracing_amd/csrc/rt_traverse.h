// rt_traverse.h -- BVH traversal core for gfx950 (wave64).
//
// Semantics are those of the reference's BVHRayTrace
// (Assets/Shaders/BVH/BVHRayTracing.cginc:27-133 with IntersectTriangle_MT97,
// Assets/Shaders/RayTracing/BruteForceRayTracing.cginc:166-204): every node
// whose box the ray passes is visited (the reference has no distance test), a leaf's triangle
// is tested when its own box passes, the closest hit wins with strict '<', so
// on equal distances the triangle met first in the reference's depth-first
// order (left leaf, right leaf, right subtree, left subtree) is kept.
// Boxes may be left out of that walk where it is certain that nothing in them can be computed nearer than the hit
// already held (RT_OPT_PRUNE_BY_DISTANCE 1, the default: traverse_rays, "leaving out boxes, with a bound") -- the result
// stays the reference's bit for bit.  One option value departs from it: with 2 (a finite pruneScale in the scene view)
// every box that begins 1.5 % beyond the closest hit is left out, the same hit except where Moller-Trumbore's distance
// of a triangle is far off (rtcore.h says when).
//
// What is different is only HOW the same result is computed:
//   * Inner nodes are a filter, not a decision.  With finite, non-zero direction
//     components IEEE subtraction and division are monotone, so a ray that passes
//     a triangle's own box passes every box that contains it: the set of triangles
//     the reference ends up testing is exactly "own box passes", and the order it
//     meets them in is a fixed permutation of the triangles (its loop's order does
//     not depend on which boxes pass).  Any walk of ANY tree over the same triangle
//     boxes that never skips a subtree holding such a triangle, and that lets the
//     triangle the reference meets first win among exactly equal distances
//     (leafRank), returns the same hit, bit for bit.  The one-ray-per-lane walker
//     therefore walks a second tree, built from finer keys than the reference's
//     (include/rt_finekey.h) and reshaped by tree rotations (include/rt_rotate.h),
//     through 32-byte nodes whose child boxes are rounded
//     OUTWARD onto half-float planes (rt_qh.h): a plane's slab value is one
//     v_fma_mix_f32, and when all rays of the wave run the same way along each axis
//     (the usual case for camera and shadow rays) near and far planes are picked
//     without min / max -- a test that may say "maybe" but never a wrong "no"; the
//     triangle's own box is exact and gets the reference's test.  Rays outside the
//     monotone case (a zero / infinite / NaN component) walk the reference's own
//     tree, exact 64-byte nodes with the reference's test at every node and no stack
//     (exact_walk), and so does the packet walker (with its own stacks).
//   * the reference's slab test is first evaluated with per-ray reciprocals and an
//     error margin; only when the margin cannot decide (a graze within a few ulp,
//     an infinity, a NaN) are the six true divisions evaluated, so the boolean is
//     always the one the reference computes;
//   * a wave keeps fetching new rays for lanes whose ray has finished
//     (ballot + prefix count + one atomic per refill), and triangle tests are
//     queued per lane in LDS and run in batches, in per-ray order, so that the
//     64 lanes stay busy although rays take very different walks;
//   * a ray source may declare that all it needs is WHETHER a hit of a certain kind exists
//     (Src::FIRST_HIT_MAY_DECIDE, the ShadowMap pass's rays): a ray for which the source can
//     show that its first accepted hit settles that ends its walk there.
#ifndef RT_TRAVERSE_H_
#define RT_TRAVERSE_H_

#include "rt_internal.h"

#define TRV_THREADS 256
#define PKT_THREADS 256
#ifndef TRV_PEND_CAP
#define TRV_PEND_CAP 10         // queued triangle tests per lane
#endif
#ifndef TRV_STACK_LDS
#define TRV_STACK_LDS 16        // stack levels kept in LDS; deeper levels spill to HBM (rare)
#endif
#ifndef TRV_REFILL_MIN
#define TRV_REFILL_MIN 40       // refill when at least this many lanes are idle (16 / 24 / 32 / 40 / 48 with two levels per step: 89.0 / 85.1 / 84.9 / 84.1 / 85.9 ms per image)
#endif
#ifndef TRV_WALK_NUM
#define TRV_WALK_NUM 1u         // leave the node phase when fewer than NUM/DEN of the lanes that still have nodes to visit
#define TRV_WALK_DEN 2u         // can go on (the rest wait for triangle tests); 3/4, 3/5, 1/2, 2/5, 1/3 with two levels per step: 87.7 / 85.8 / 85.1 / 85.3 / 86.8 ms per image
#endif
// RT_OPT_PRUNE_BY_DISTANCE 2: boxes that begin beyond the closest hit a ray holds are not entered, "beyond" with this much
// room, because the distance of a hit (Moller-Trumbore) and the entry distance of a box (slab test) are two different
// float computations and only agree to rounding -- to rounding of a quotient whose divisor can be nearly all rounding
// error (a ray nearly in a triangle's plane), which is why this is an option and the default is the rule with a bound.
#define RT_PRUNE_SCALE 1.015625f
// Long walks are shared out at the end of a launch (RT_OPT_SHARE_LONG_WALKS, traverse_rays): once the launch has no
// more rays to deal and at least this many lanes of a wave are idle, idle lanes take over stacked subtrees of lanes that
// are still walking; the node phase then ends after at most TRV_SHARE_TRIPS steps so that the hand-over comes round.
#ifndef TRV_TRI_MIN_LANES
#define TRV_TRI_MIN_LANES 8u    // the triangle phase ends when fewer lanes than this still have triangles queued (0 = when none has; 0 / 8 / 16 / 24 / 32: 84.3 / 82.8 / 83.0 / 83.6 / 85.4 ms per image)
#endif
#ifndef TRV_SHARE_MIN_IDLE
#define TRV_SHARE_MIN_IDLE 16u  // (32 / 16 / 8 with all pairs of a round handed over at once: 0.52 / 0.475 / 0.485 ms for 0.26 M camera rays, an eighth of the image 14.9 / 14.5 / 14.75 ms)
#endif
#ifndef TRV_SHARE_TRIPS
#define TRV_SHARE_TRIPS 16u
#endif
#ifndef TRV_SHARE_ROUNDS
#define TRV_SHARE_ROUNDS 3u     // levels a lane hands over per turn of the main loop, at most (1 / 3 / 6: 0.51 / 0.52 / 0.54 ms for 0.26 M camera rays, the whole image 90.5 / 89.7 / 90.8 ms)
#endif
#define TRV_DONE 0xFFFFFFFFu
#define TRV_REDO 0xFFFFFFFEu    // in WalkState::tos of a lane whose node is TRV_DONE: this ray takes the exact, stackless walk next
#define TRV_LDS_WORDS ((TRV_STACK_LDS + TRV_PEND_CAP) * TRV_THREADS)
static_assert(TRV_LDS_WORDS * 4 * 6 <= 160 * 1024, "six blocks per CU must fit the LDS");
#ifndef TRV_BLOCKS_PER_CU
#define TRV_BLOCKS_PER_CU 6u    // blocks launched per CU: as many as are resident at a time (80 registers per lane, 26 KB of
                                // LDS per block); with 8 the last two start when the others end and find the queue empty
                                // (95.5 instead of 94.9 ms per image)
#endif
static_assert(TRV_BLOCKS_PER_CU * TRV_THREADS * RT_STACK_DEPTH <= RT_SPILL_WORDS_PER_CU, "a spill area holds every stack level of every lane of a launch");

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

// IntersectTriangle_MT97 + the acceptance test of CheckTriangle.
// (returns whether h was replaced)
// The reference keeps the FIRST triangle it meets among equally distant ones (strict '<' in its
// own visiting order).  A walker that meets triangles in another order passes `leafRank`, the
// position of every leaf's triangle in that order, and lets the earlier one win a tie; a walker
// that follows the reference's order passes nullptr.
__device__ __forceinline__ bool tri_test_values(const rt_f3 v0, const rt_f3 e1, const rt_f3 e2, uint32_t leaf, const rt_f3 o,
                                                const rt_f3 d, RtHitRec &h, const uint32_t *__restrict__ leafRank = nullptr)
{
    const rt_f3 pvec = rt_cross3(d, e2);
    const float det = rt_dot3(e1, pvec);
    if (det < RT_EPSILON) return false;
    const float inv_det = 1.0f / det;
    const rt_f3 tvec = rt_sub3(o, v0);
    const float u = rt_dot3(tvec, pvec) * inv_det;
    if (u < 0.0f || u > 1.0f) return false;
    const rt_f3 qvec = rt_cross3(tvec, e1);
    const float v = rt_dot3(d, qvec) * inv_det;
    if (v < 0.0f || u + v > 1.0f) return false;
    const float t = rt_dot3(e2, qvec) * inv_det;
    bool closer = t > 0.0f && t < h.t;
    if (leafRank && t > 0.0f && t == h.t && h.leaf != RT_INVALID) closer = leafRank[leaf] < leafRank[h.leaf];
    if (closer) {
        h.t = t;
        h.u = u;
        h.v = v;
        h.leaf = leaf;
        return true;
    }
    return false;
}

// CheckTriangle (BVHRayTracing.cginc:41-76) on the record of sorted leaf `leaf`:
// the triangle's own exact box with the reference's test, then
// IntersectTriangle_MT97 on (v0, e1 = v1 - v0, e2 = v2 - v0) and the acceptance test.
// Returns whether the box passed (needed for the work counters only).
// A triangle changes the hit only if BOTH tests pass, so without counters the
// cheaper-to-fail one runs first: Moller-Trumbore rejects nearly every candidate,
// and the box (a fourth 16-byte load and the slab test) is looked at only for a
// triangle that would otherwise be accepted.
template <bool COUNT>
__device__ __forceinline__ bool tri_test(const RtSceneView &sv, uint32_t leaf, const rt_f3 o, const rt_f3 d, const rt_f3 r, RtHitRec &h)
{
    const float4 *p = (const float4 *)(sv.leafTri + leaf);
    const float4 a = p[0], b = p[1], c = p[2];
    const rt_f3 v0 = rt_mk3(a.x, a.y, a.z), e1 = rt_mk3(a.w, b.x, b.y), e2 = rt_mk3(b.z, b.w, c.x);
    if (COUNT) {
        const float4 e = p[3];
        if (!box_hit(c.z, c.w, e.x, e.y, e.z, e.w, o, d, r)) return false;
        tri_test_values(v0, e1, e2, leaf, o, d, h, sv.leafRank);
        return true;
    }
    RtHitRec cand = h;
    if (tri_test_values(v0, e1, e2, leaf, o, d, cand, sv.leafRank)) {              // Moller-Trumbore would accept it
        const float4 e = p[3];
        if (box_hit(c.z, c.w, e.x, e.y, e.z, e.w, o, d, r)) h = cand;
    }
    return true;
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

// Ray distribution: one cursor shared by all waves of the launch (zero at
// launch).  A wave reserves a run of consecutive rays with one atomic and deals
// them to its idle lanes over the following refills.  Atomics on one address
// complete at roughly one per 10 ns for the whole device: one atomic per refill
// (24-64 rays) held a 190 M-ray launch at 2.8 G rays/s.  Reservations shrink
// with the work that is left (short launches and the end of long ones would
// otherwise finish unevenly: fixed runs of 2048 rays tripled the time of the
// small bounce launches).
#ifndef TRV_CHUNK_MAX
#define TRV_CHUNK_MAX 512u      // most rays a wave reserves with one atomic
#endif
#ifndef TRV_CHUNK_MIN
#define TRV_CHUNK_MIN 64u
#endif
#ifndef TRV_CHUNK_SHARE
#define TRV_CHUNK_SHARE 2u      // a reservation is at most 1/SHARE of an even share of the rays that are left (1 / 2 / 4 with two levels per step: 84.0 / 84.3 / 84.9 ms per image, one frame in flight 4.2 / 4.1 / 4.3 ms; 2 / 4 / 8 / 32 before: 110.5 / 110.7 / 113.0 / 133.9)
#endif
// The feed's state is the same in every lane of a wave, and says so to the compiler (which then keeps it in scalar
// registers).
__device__ __forceinline__ uint32_t wave_uniform(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }

// How many lanes of a ballot are set, as a 32-bit scalar the compiler cannot see through: it knows the population
// count of a 64-bit mask as a 64-bit number, and scalar code has no ordered 64-bit comparison, so every `<` on such a
// count in a loop test became a vector instruction (three per step of the node phase).
__device__ __forceinline__ uint32_t lane_count(unsigned long long mask)
{
    uint32_t n;
    asm("s_bcnt1_i32_b64 %0, %1" : "=s"(n) : "s"(mask) : "scc");
    return n;
}

// the value lane `from` (wave-uniform) holds
__device__ __forceinline__ uint32_t lane_value(uint32_t x, uint32_t from) { return (uint32_t)__builtin_amdgcn_readlane((int)x, (int)from); }
__device__ __forceinline__ float lane_value(float x, uint32_t from) { return __uint_as_float(lane_value(__float_as_uint(x), from)); }

struct RayFeed {
    uint32_t *cursor;
    uint32_t total, next, end, chunk, waves2, own;    // wave-uniform (own: the cursor of a wave that draws alone, cursor == nullptr)
    bool lastChunk, exhausted;
    __device__ __forceinline__ uint32_t chunk_for(uint32_t remaining) const
    {
        // guided self-scheduling: big reservations while there is plenty of work, small ones
        // towards the end so that the waves finish together
        // (0 = no more reserving: from there on a wave draws exactly what its idle lanes need)
        const uint32_t c = (remaining / waves2) & ~15u;
        return c < TRV_CHUNK_MIN ? 0u : (c > TRV_CHUNK_MAX ? TRV_CHUNK_MAX : c);
    }
    __device__ __forceinline__ void init(uint32_t *c, uint32_t n)
    {
        cursor = c;
        total = wave_uniform(n);
        next = end = 0;
        own = 0;
        waves2 = gridDim.x * (blockDim.x >> 6) * TRV_CHUNK_SHARE;
        chunk = chunk_for(total);
        lastChunk = false;
        exhausted = total == 0;
    }
    // Wave-uniform call: hands up to `want` rays to the lanes with rank 0..want-1 (`rank` = this
    // lane's position among the asking lanes).  Returns this lane's ray or TRV_DONE.
    __device__ __forceinline__ uint32_t draw(uint32_t want, uint32_t rank)
    {
        if (next == end && !lastChunk) {
            const uint32_t take = chunk ? chunk : want;
            uint32_t base;
            if (cursor) {
                base = 0;
                if ((threadIdx.x & 63u) == 0) base = atomicAdd(cursor, take);
                base = wave_uniform(base);                     // lane 0's
            } else {                                           // rays that are this wave's alone: nothing to share a cursor with
                base = own;
                own += take;
            }
            next = base < total ? base : total;
            end = total - next < take ? total : next + take;
            lastChunk = end == total;
            chunk = chunk_for(total - end);
        }
        const uint32_t n = want < end - next ? want : end - next;
        const uint32_t mine = rank < n ? next + rank : TRV_DONE;
        next += n;
        exhausted = lastChunk && next == end;
        return mine;
    }
};

// The reference's own walk for ONE ray, exact 64-byte nodes of the reference's tree, the reference's test at
// every box, its order (left leaf, right leaf, right subtree, left subtree) -- for the rare rays outside the
// monotone case.  No stack: on the way back up the parent link says where the walk came from; from a right
// subtree the left sibling is next if it is internal and passes (its box is tested again, same boolean), from a
// left subtree the climb goes on.
template <bool COUNT>
__device__ __forceinline__ void exact_walk(const RtSceneView &sv, const rt_f3 o, const rt_f3 d, const rt_f3 r, RtHitRec &h, uint32_t cnt[3])
{
    if (sv.nTriangles == 0) return;
    if (COUNT) cnt[0]++;
    const RtAABB &rb = sv.rootBox;
    if (sv.nTriangles < 2 || !box_hit_exact(rb.min[0], rb.min[1], rb.min[2], rb.max[0], rb.max[1], rb.max[2], o, d)) return;
    uint32_t node = 0;
    for (;;) {
        const float4 *np = (const float4 *)(sv.nodes + node);
        const float4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
        const uint32_t l = __float_as_uint(q3.x), rr = __float_as_uint(q3.y);
        const bool lLeaf = (l & RT_LEAF_BIT) != 0, rLeaf = (rr & RT_LEAF_BIT) != 0;
        const bool hitL = box_hit_exact(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, o, d);
        const bool hitR = box_hit_exact(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, o, d);
        if (COUNT) cnt[0] += 2u;
        if (lLeaf && hitL) {
            const bool boxed = tri_test<COUNT>(sv, l & ~RT_LEAF_BIT, o, d, r, h);
            if (COUNT) { cnt[1]++; cnt[2] += boxed ? 1u : 0u; }
        }
        if (rLeaf && hitR) {
            const bool boxed = tri_test<COUNT>(sv, rr & ~RT_LEAF_BIT, o, d, r, h);
            if (COUNT) { cnt[1]++; cnt[2] += boxed ? 1u : 0u; }
        }
        if (!rLeaf && hitR) { node = rr; continue; }
        if (!lLeaf && hitL) { node = l; continue; }
        // nothing below: climb until a left sibling is still to be walked
        uint32_t parent = __float_as_uint(q3.z);
        for (;;) {
            if (parent == RT_INVALID) return;
            const float4 *pp = (const float4 *)(sv.nodes + parent);
            const float4 p0 = pp[0], p1 = pp[1], p3 = pp[3];
            const uint32_t pl = __float_as_uint(p3.x), pr = __float_as_uint(p3.y);
            const bool fromRight = pr == node;                         // `node` is internal, so this names the right child
            node = parent;
            parent = __float_as_uint(p3.z);
            if (fromRight && !(pl & RT_LEAF_BIT) && box_hit_exact(p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, o, d)) {
                node = pl;
                break;
            }
        }
    }
}

#ifdef TRV_SHARE_STATS            // diagnostics of the end of a launch (tools/exp_share.py): node-phase trips and busy lanes of a wave after the feed ran dry
#define TRV_SHARE_STATS_PARAM , unsigned long long *shStat
#define TRV_SHARE_STATS_ARG , shStat
#else
#define TRV_SHARE_STATS_PARAM
#define TRV_SHARE_STATS_ARG
#endif
// Per-lane state of the walk through the half-float nodes.
struct WalkState {
    uint32_t node, sp, npend;
    uint32_t tos;        // one level per step: the stack's top level, mirrored; both walks: TRV_REDO = this ray takes the exact walk next
};

// The node phase of a wave: walk the tree, queue triangles whose box may pass.  A lane whose pending list has no
// room for two more entries waits; the phase ends when too few lanes are still walking (the others wait for
// triangle tests, a refill or retirement).  OCT < 8: every ray of the wave that is in the tree runs the same way
// along each axis (rt_slab_octant) and the box test picks near and far planes without min / max; OCT == 8: any mix.
// NEAR: where both children have to be visited the walk goes on with the one the ray enters first (no result depends
// on the order: equal distances are settled by leafRank).  For rays that may stop at their first hit
// (Src::FIRST_HIT_MAY_DECIDE) the hit then comes sooner, and for the others the closest hit, which `limit` then uses:
// limit: this lane's ray does not enter a box whose entry distance (a lower bound: the slab interval the walker sees
// contains the reference's) exceeds it.  The caller passes (distance of the closest hit so far) x RT_PRUNE_SCALE, or
// +inf.  A triangle in a box left out lies, to rounding, behind the hit the ray already holds and could not replace
// it; "to rounding" is where this differs from the reference's walk, which enters every box the ray passes: the two
// return different hits only if Moller-Trumbore puts a triangle more than 1.5 % nearer than the box that holds it,
// which takes a ray within a few microradians of the triangle's plane (more for slivers: the relative error is about
// 6e-8 / (that angle x the sine of the angle between the triangle's edges)).
template <uint32_t OCT, bool COUNT, bool NEAR>
__device__ __forceinline__ void node_phase(const RtSceneView &sv, const RtRaySlab &slab, WalkState &w, uint32_t *stk, uint32_t *pend,
                                           uint32_t *spill, const uint32_t spillLanes, uint32_t cnt[3], const bool feedExhausted,
                                           const float limit, const uint32_t shareTrips TRV_SHARE_STATS_PARAM)
{
    // shareTrips (wave-uniform): steps this phase may take while lanes idle that could take over part of a walk
    // (0xFFFFFFFF: no such limit)
    uint32_t trips = 0;
    // After a pop the level below the old top becomes the mirrored one, at the lane's next step or when the phase ends
    // (the read of that level is then long back).  `popped` is a 32-bit lane value, not a boolean: a boolean lives as a
    // bit of a scalar lane mask, which the compiler rebuilds from a vector register when lanes leave and rejoin a loop --
    // from a register that holds nothing for a lane that left early; a word per lane is never rebuilt.
    uint32_t popped = 0u, below = TRV_DONE;
    for (;;) {
        // (one ballot per comparison, combined as scalar masks: the ballot of a conjunction is made by way of a vector register)
        const bool busy = w.node != TRV_DONE, room = w.npend <= TRV_PEND_CAP - 2;
        const bool act = busy && room;
        const unsigned long long busyMask = __ballot(busy);
        const uint32_t nAct = lane_count(busyMask & __ballot(room));
        const uint32_t nBusy = lane_count(busyMask);
        if (nAct == 0u || nAct * TRV_WALK_DEN < nBusy * TRV_WALK_NUM || (!feedExhausted && nBusy < 64u - TRV_REFILL_MIN)) break;
        if (nBusy <= 64u - TRV_SHARE_MIN_IDLE && ++trips > shareTrips) break;
#ifdef TRV_SHARE_STATS
        if (feedExhausted) { shStat[0]++; shStat[1] += nBusy; }
#endif
        if (act) {
            const uint4 *np = (const uint4 *)(sv.qnodes + w.node);
            uint4 qa = np[0], qb = np[1];
            w.tos = popped != 0u ? below : w.tos;                // after the loads are on their way
            popped = 0u;
            // keep the fetch at two 16-byte loads per lane (the compiler would otherwise split the second)
            asm volatile("" : "+v"(qb.x), "+v"(qb.y), "+v"(qb.z), "+v"(qb.w));
            const uint32_t l = qb.z, rr = qb.w;
            float nearL, nearR;
            const bool maybeL = OCT < 8u ? rt_slab_maybe_octant_near<(OCT & 7u)>(slab, qa.x, qa.y, qa.z, &nearL) : rt_slab_maybe_near(slab, qa.x, qa.y, qa.z, &nearL);
            const bool maybeR = OCT < 8u ? rt_slab_maybe_octant_near<(OCT & 7u)>(slab, qa.w, qb.x, qb.y, &nearR) : rt_slab_maybe_near(slab, qa.w, qb.x, qb.y, &nearR);
            const bool hitL = maybeL && nearL <= limit, hitR = maybeR && nearR <= limit;
            const bool lLeaf = (l & RT_LEAF_BIT) != 0, rLeaf = (rr & RT_LEAF_BIT) != 0;
            if (COUNT) cnt[0] += 2u;
            // Straight-line bookkeeping: every store below happens whether or not it is needed
            // (a slot past the end of a list is scratch), only the counters are conditional.
            pend[w.npend * TRV_THREADS] = l & ~RT_LEAF_BIT;
            w.npend += (lLeaf && hitL) ? 1u : 0u;
            pend[w.npend * TRV_THREADS] = rr & ~RT_LEAF_BIT;
            w.npend += (rLeaf && hitR) ? 1u : 0u;
            const bool goL = !lLeaf && hitL, goR = !rLeaf && hitR;
            // with both to visit: `first` now, `later` from the stack (the right one first unless NEAR says otherwise)
            const bool leftFirst = NEAR && nearL < nearR;
            const uint32_t first = leftFirst ? l : rr, later = leftFirst ? rr : l;
            if (__builtin_expect(w.sp < TRV_STACK_LDS, 1)) {
                // `tos` mirrors level sp-1 in a register, so a pop needs no LDS round trip before
                // the next node can be fetched; the level below is read now and taken over at the
                // start of this lane's next step (level 0 holds TRV_DONE).
                stk[w.sp * TRV_THREADS] = later;
                below = stk[(w.sp > 1u ? w.sp - 2u : 0u) * TRV_THREADS];
                const bool both = goL && goR, none = !(goL || goR);
                uint32_t next = goR ? rr : l;
                next = both ? first : next;
                next = none ? w.tos : next;
                w.node = next;
                w.tos = both ? later : w.tos;
                popped = none ? 1u : 0u;
                w.sp += both ? 1u : 0u;
                w.sp -= none ? 1u : 0u;
            } else if (__builtin_expect(w.sp > RT_STACK_DEPTH, 0)) {
                // No walk gets here.  A stack pointer out of range would index the spill area anywhere, so whatever
                // produced it, the ray starts over with the walk that has no stack.
                w.node = TRV_DONE;
                w.sp = 1;
                w.tos = TRV_REDO;
                w.npend = 0;
            } else {                                                 // deep levels live in HBM
                w.node = goR ? rr : l;
                if (goL && goR) {
                    w.node = first;
                    if (w.sp < RT_STACK_DEPTH) {
                        spill[(size_t)(w.sp - TRV_STACK_LDS) * spillLanes] = later;
                        w.sp++;
                        w.tos = later;
                    } else {
                        // Deeper than the stack (rotations may lengthen a path of the tree past the 32 levels a Karras
                        // tree has): the ray starts over on the reference's tree with the walk that needs no stack
                        // and returns the same hit.
                        w.node = TRV_DONE;
                        w.sp = 1;
                        w.tos = TRV_REDO;
                        w.npend = 0;
                    }
                } else if (!(goL || goR)) {
                    w.node = w.tos;
                    w.sp--;
                    const uint32_t lvl = w.sp ? w.sp - 1u : 0u;
                    w.tos = lvl < TRV_STACK_LDS ? stk[lvl * TRV_THREADS] : spill[(size_t)(lvl - TRV_STACK_LDS) * spillLanes];
                }
            }
        }
    }
    w.tos = popped != 0u ? below : w.tos;
}

// The node phase over the 64-byte records that hold two levels of the tree (RtQNode4): one step fetches the boxes of up
// to four grandchildren, queues the leaves among them whose box may pass, goes on with the nearest internal one that
// may pass and stacks the others.  Same stack as node_phase (level 0 holds TRV_DONE, entries are node numbers of the
// same tree), without the register copy of its top; same end test; same results, because inner nodes are only a filter
// (a box that is not tested -- the child's, between a node and its grandchildren -- cannot lose a triangle).
template <uint32_t OCT, bool ORDERED, bool BOUNDED>
__device__ __forceinline__ bool wide_box(const RtRaySlab &slab, uint32_t wx, uint32_t wy, uint32_t wz, uint32_t ref, const float limit, const float bw[3], const uint32_t largeBit, float *tnear)
{
    // an unused place holds a box whose low planes lie above its high planes: the octant test never passes it for a ray
    // that is walked here (rt_slab_rejects_empty_box, checked when the ray is dealt), the test that sorts a slab's two
    // planes would
    if (BOUNDED) {
        // the bounded rule: what is compared with the limit is the smallest ray parameter at which the ray can be within
        // the rule's distance of the box (bw: that distance per axis, in units of the ray parameter), and a box that holds
        // a large triangle is never left out
        float shifted;
        const bool maybe = rt_slab_maybe_shifted<OCT>(slab, wx, wy, wz, bw, tnear, &shifted);
        return maybe && (!(shifted > limit) || (ref & largeBit) != 0u) && (OCT < 8u || ref != RT_INVALID);
    }
    const bool maybe = OCT < 8u ? rt_slab_maybe_octant_near<(OCT & 7u)>(slab, wx, wy, wz, tnear) : rt_slab_maybe_near(slab, wx, wy, wz, tnear);
    return maybe && (!ORDERED || *tnear <= limit) && (OCT < 8u || ref != RT_INVALID);
}

// COUNT: every place of a record that holds a child is a box tested (cnt[0]).
// ORDERED: the walk goes front to back and leaves out what begins beyond `limit`.  A walk that enters every box a ray
// passes and ends only when all are visited (limit = +inf, no stop at a first hit) gains nothing from an order: without
// it the step goes on with the first place that is to be visited and stacks the others as they come, no distances
// sorted (a quarter of the step's vector instructions).
// BOUNDED (with ORDERED): boxes are left out by the bounded rule (traverse_rays, "leaving out boxes, with a bound"):
// `limit` is the closest hit's distance and bw the rule's distance per axis (or, for the 1.5 % rule: the hit's distance
// with that room, and zero).
template <uint32_t OCT, bool COUNT, bool ORDERED, bool BOUNDED = false>
__device__ __forceinline__ void node_phase_wide(const RtSceneView &sv, const RtRaySlab &slab, WalkState &w, uint32_t *stk, uint32_t *pend,
                                                uint32_t *spill, const uint32_t spillLanes, uint32_t cnt[3], const bool feedExhausted, const float limit,
                                                const uint32_t shareTrips, const float bw0 = 0.0f, const float bw1 = 0.0f, const float bw2 = 0.0f,
                                                const uint32_t largeBit = RT_REF_LARGE)
{
    // (largeBit: RT_REF_LARGE, or 0 where the 1.5 % rule goes through this phase: that rule knows no large triangles)
    const float bw[3] = {bw0, bw1, bw2};
    uint32_t trips = 0;
    for (;;) {
        const bool busy = w.node != TRV_DONE, room = w.npend + 4u <= TRV_PEND_CAP;
        const bool act = busy && room;
        const unsigned long long busyMask = __ballot(busy);
        const uint32_t nAct = lane_count(busyMask & __ballot(room));
        const uint32_t nBusy = lane_count(busyMask);
        if (nAct == 0u || nAct * TRV_WALK_DEN < nBusy * TRV_WALK_NUM || (!feedExhausted && nBusy < 64u - TRV_REFILL_MIN)) break;
        if (nBusy <= 64u - TRV_SHARE_MIN_IDLE && ++trips > shareTrips) break;
        if (act) {
            const uint4 *np = (const uint4 *)(sv.wnodes + (w.node & RT_REF_INDEX));     // (a reference may carry RT_REF_LARGE)
            const uint4 qa = np[0], qb = np[1], qc = np[2], qd = np[3];
            const uint32_t r0 = qd.x, r1 = qd.y, r2 = qd.z, r3 = qd.w;
            if (COUNT) cnt[0] += (r0 != RT_INVALID ? 1u : 0u) + (r1 != RT_INVALID ? 1u : 0u) + (r2 != RT_INVALID ? 1u : 0u) + (r3 != RT_INVALID ? 1u : 0u);
            float n0, n1, n2, n3;
            const bool h0 = wide_box<OCT, ORDERED, BOUNDED>(slab, qa.x, qa.y, qa.z, r0, limit, bw, largeBit, &n0);
            const bool h1 = wide_box<OCT, ORDERED, BOUNDED>(slab, qa.w, qb.x, qb.y, r1, limit, bw, largeBit, &n1);
            const bool h2 = wide_box<OCT, ORDERED, BOUNDED>(slab, qb.z, qb.w, qc.x, r2, limit, bw, largeBit, &n2);
            const bool h3 = wide_box<OCT, ORDERED, BOUNDED>(slab, qc.y, qc.z, qc.w, r3, limit, bw, largeBit, &n3);
            const bool l0 = (r0 & RT_LEAF_BIT) != 0, l1 = (r1 & RT_LEAF_BIT) != 0, l2 = (r2 & RT_LEAF_BIT) != 0, l3 = (r3 & RT_LEAF_BIT) != 0;
            // leaves whose box may pass: every store happens, the count moves on only where it is needed (the reference
            // goes in as it is, leaf bit and all: the triangle phase takes the bit off)
            pend[w.npend * TRV_THREADS] = r0;
            w.npend += (l0 && h0) ? 1u : 0u;
            pend[w.npend * TRV_THREADS] = r1;
            w.npend += (l1 && h1) ? 1u : 0u;
            pend[w.npend * TRV_THREADS] = r2;
            w.npend += (l2 && h2) ? 1u : 0u;
            pend[w.npend * TRV_THREADS] = r3;
            w.npend += (l3 && h3) ? 1u : 0u;
            // internal nodes to visit: the nearest one next, the others on the stack
            const bool g0 = !l0 && h0, g1 = !l1 && h1, g2 = !l2 && h2, g3 = !l3 && h3;
            const bool any = g0 || g1 || g2 || g3;
            uint32_t t0 = r0, t1 = r1, t2 = r2, t3 = r3;                 // t0 next; t3, t2, t1 stacked in that order where p0, p1, p2 say so
            bool p0, p1, p2;
            if (ORDERED) {
                const float k0 = g0 ? n0 : rt_inf(), k1 = g1 ? n1 : rt_inf(), k2 = g2 ? n2 : rt_inf(), k3 = g3 ? n3 : rt_inf();
                // all four in the order the ray enters them (a sorting network of five exchanges): the nearest next, the
                // others stacked farthest first
                float s0 = k0, s1 = k1, s2 = k2, s3 = k3;
#define TRV_EXCHANGE(a, b, ra, rb) { const bool x_ = b < a; const float lo_ = x_ ? b : a, hi_ = x_ ? a : b; const uint32_t rl_ = x_ ? rb : ra, rh_ = x_ ? ra : rb; a = lo_; b = hi_; ra = rl_; rb = rh_; }
                TRV_EXCHANGE(s0, s1, t0, t1)
                TRV_EXCHANGE(s2, s3, t2, t3)
                TRV_EXCHANGE(s0, s2, t0, t2)
                TRV_EXCHANGE(s1, s3, t1, t3)
                TRV_EXCHANGE(s1, s2, t1, t2)
#undef TRV_EXCHANGE
                p0 = s3 < rt_inf(); p1 = s2 < rt_inf(); p2 = s1 < rt_inf();
            } else {
                // the first place to be visited next, every later one stacked
                t0 = g0 ? r0 : (g1 ? r1 : (g2 ? r2 : r3));
                p2 = g1 && g0;
                p1 = g2 && (g0 || g1);
                p0 = g3 && (g0 || g1 || g2);
            }
            const uint32_t bestRef = t0;
            if (__builtin_expect(w.sp + 2u < TRV_STACK_LDS, 1)) {    // levels sp .. sp + 2 are all LDS levels
                stk[w.sp * TRV_THREADS] = t3;
                w.sp += p0 ? 1u : 0u;
                stk[w.sp * TRV_THREADS] = t2;
                w.sp += p1 ? 1u : 0u;
                stk[w.sp * TRV_THREADS] = t1;
                w.sp += p2 ? 1u : 0u;
                w.node = bestRef;
                if (!any) {
                    w.sp -= 1u;
                    w.node = stk[w.sp * TRV_THREADS];
                }
            } else {                                                 // near the end of the LDS levels: one push at a time, deep levels in HBM
                const uint32_t refs[3] = {t3, t2, t1};
                const bool push[3] = {p0, p1, p2};
                bool overflow = false;
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    if (push[k]) {
                        if (w.sp < TRV_STACK_LDS) stk[w.sp * TRV_THREADS] = refs[k];
                        else if (w.sp < RT_STACK_DEPTH) spill[(size_t)(w.sp - TRV_STACK_LDS) * spillLanes] = refs[k];
                        else overflow = true;
                        if (!overflow) w.sp++;
                    }
                w.node = bestRef;
                if (!any) {
                    w.sp -= 1u;
                    w.node = w.sp < TRV_STACK_LDS ? stk[w.sp * TRV_THREADS] : spill[(size_t)(w.sp - TRV_STACK_LDS) * spillLanes];
                }
                if (overflow) {                                      // deeper than the stack: the ray starts over on the stackless walk
                    w.node = TRV_DONE;
                    w.sp = 1;
                    w.tos = TRV_REDO;
                    w.npend = 0;
                }
            }
        }
    }
}

// Persistent traversal of `src.count()` rays by the waves of the launch (*cursor is 0 at launch).
//   Src: uint32_t count(); void load(uint32_t i, rt_f3 &o, rt_f3 &d); void store(uint32_t i, const RtHitRec &h);
// lds: TRV_LDS_WORDS words per block; this lane's first TRV_STACK_LDS stack levels are
// lds[level * TRV_THREADS + tid], its pending triangles lds[(TRV_STACK_LDS + k) * TRV_THREADS + tid];
// stack levels from TRV_STACK_LDS up live in sv.stackSpill[(level - TRV_STACK_LDS) * lanes + global lane].
template <bool COUNT, class Src>
__device__ __forceinline__ void traverse_rays(const RtSceneView &sv, Src &src, uint32_t *__restrict__ cursor, uint32_t *lds)
{
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t *stk = lds + threadIdx.x;
    uint32_t *pend = lds + TRV_STACK_LDS * TRV_THREADS + threadIdx.x;
    const uint32_t spillLanes = gridDim.x * TRV_THREADS;
    uint32_t *spill = sv.stackSpill + (blockIdx.x * TRV_THREADS + threadIdx.x);
    RayFeed feed;
    feed.init(cursor, src.count());

    bool hasRay = false;
    uint32_t firstHit = 0;      // does this ray's first hit settle its answer: 0 not asked yet, 1 yes, 2 no
    uint32_t rayIdx = 0;
    // Sharing a long walk (see share_out below).  Per lane: bits 0-5 the lowest stack level that still holds a subtree
    // of its own (the levels from 1 up to there were given away and hold TRV_DONE), bits 8-14 how many lanes still
    // walk a subtree it gave them, bit 24 "this lane walks a subtree another lane gave it", bits 16-21 that lane.
    uint32_t share = 1u;
#ifdef TRV_SHARE_STATS
    unsigned long long shT0 = 0ull, shGiven = 0ull, shIters = 0ull;   // wave-uniform: clock when the feed ran dry, hand-overs, trips of the main loop since
    unsigned long long shStat[2] = {0ull, 0ull}, shClk[4] = {0ull, 0ull, 0ull, 0ull}, shMark = 0ull;   // clocks after the feed ran dry: hand-over, node phase, triangle phase, the rest
#define TRV_SHARE_CLK(k) if (feed.exhausted) { const unsigned long long now_ = clock64(); shClk[k] += now_ - shMark; shMark = now_; }
#else
#define TRV_SHARE_CLK(k)
#endif
    const bool sharing = !COUNT && sv.shareWalks != 0u;
    WalkState w;
    w.node = TRV_DONE; w.sp = 0; w.npend = 0; w.tos = TRV_DONE;
    rt_f3 o = rt_splat3(0.0f), d = rt_splat3(0.0f), r = rt_splat3(0.0f);
    RtRaySlab slab;
    memset(&slab, 0, sizeof slab);
    RtHitRec h;
    h.t = rt_inf(); h.u = 0.0f; h.v = 0.0f; h.leaf = RT_INVALID;
    uint32_t cnt[3] = {0, 0, 0};
    uint32_t nRays = 0, nHits = 0;

#ifdef TRV_PROFILE
    // phase clocks of this wave (shader clock), reported through sv.stats[16..]
    unsigned long long pT[4] = {0, 0, 0, 0}, pN[6] = {0, 0, 0, 0, 0, 0}, pMark = clock64();
#define TRV_PHASE(k) { const unsigned long long now_ = clock64(); pT[k] += now_ - pMark; pMark = now_; }
#else
#define TRV_PHASE(k)
#endif
    for (;;) {
        // ---- refill idle lanes: ballot, prefix count, one atomic per wave
        const unsigned long long idle = __ballot(!hasRay);
        const uint32_t nIdle = lane_count(idle);
        if (!feed.exhausted && nIdle >= TRV_REFILL_MIN) {
            const uint32_t mine = feed.draw(nIdle, hasRay ? 64u : (uint32_t)__popcll(idle & lt));
            if (mine != TRV_DONE) {
                rayIdx = mine;
                src.load(mine, o, d);
                r = rt_mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                slab = rt_ray_slab(sv.grid, o, r);
                h.t = rt_inf(); h.u = 0.0f; h.v = 0.0f; h.leaf = RT_INVALID;
                hasRay = true;
                firstHit = 0;
                share = 1u;
                stk[0] = TRV_DONE;                             // what the last pop of a ray returns
                w.sp = 1;
                w.tos = TRV_DONE;
                w.npend = 0;
                w.node = TRV_DONE;
                if (COUNT) nRays++;
                if (__builtin_expect(rt_ray_needs_exact_walk(o, d) || !rt_slab_finite(slab) || !rt_slab_rejects_empty_box(slab), 0)) {
                    w.tos = TRV_REDO;                          // the whole walk at once, after the node phase; retired below
                } else if (sv.nTriangles != 0) {
                    if (COUNT) cnt[0]++;
                    const RtAABB &rb = sv.rootBox;
                    if (sv.nTriangles >= 2 && box_hit(rb.min[0], rb.min[1], rb.min[2], rb.max[0], rb.max[1], rb.max[2], o, d, r))
                        w.node = 0;
                }
            }
        }
        // ---- the end of a launch: idle lanes take over stacked subtrees of the lanes that are still walking.
        // A launch lasts as long as its longest ray, a chain of dependent fetches (the longest camera rays of a
        // finely tessellated scene take thousands of steps, most of a millisecond during which the chip idles), but
        // the entries of a ray's stack are independent subtrees and no result depends on the order they are walked in:
        // the closest hit wins, equal distances go by leafRank, and for a ray that may stop at its first hit any hit
        // will do.  So once the launch has no rays left to deal, a lane without a ray copies the ray and the hit so
        // far of a lane that walks, takes the BOTTOM entry of its stack (the subtree it would reach last) and leaves
        // TRV_DONE there -- the giver's walk then ends when it pops down to that level, as all below was given away
        // too.  The level taken is never the giver's top one, which it mirrors in a register.  A taker may give
        // in turn.  When a taker's walk ends its hit goes back to the giver (deliver, below), and a giver's ray
        // retires when its own walk and all it gave away are done.
        uint32_t shareTrips = 0xFFFFFFFFu;
#ifdef TRV_SHARE_STATS
        if (feed.exhausted && shT0 == 0ull) shT0 = shMark = clock64();
        if (feed.exhausted) shIters++;
        TRV_SHARE_CLK(3)
#endif
        if (sharing && feed.exhausted) {
            shareTrips = TRV_SHARE_TRIPS;
            // All pairs of a round at once: the r-th lane that gives with the r-th lane that takes.  Lane r learns which
            // lane is the r-th giver (a forward permute: every giver sends its number to lane "its rank"; the others send
            // to lane 63, which is the rank of no giver unless all 64 give, and then nobody else sends), a taker fetches its
            // giver's ray and hit with backward permutes and the subtree from the giver's stack in LDS.  In a further
            // round the givers hand over their next level, as long as lanes are left without a ray.
            for (uint32_t round = 0; round < TRV_SHARE_ROUNDS; ++round) {
                const uint32_t lowest = share & 63u;
                const bool canGive = hasRay && w.node != TRV_DONE && lowest + 2u <= w.sp && lowest < TRV_STACK_LDS;
                const unsigned long long give = __ballot(canGive), take = __ballot(!hasRay);
                const uint32_t nTake = lane_count(take);
                if (give == 0ull || nTake < (round == 0u ? TRV_SHARE_MIN_IDLE : 1u)) break;
                const uint32_t gRank = (uint32_t)__popcll(give & lt), tRank = (uint32_t)__popcll(take & lt);
                // (a lane number comes with bit 7 set: a lane that nobody sends to receives zero)
                const uint32_t giverOfRank = (uint32_t)__builtin_amdgcn_ds_permute((int)((canGive ? gRank : 63u) << 2), (int)(canGive ? lane | 0x80u : 0u));
                const uint32_t fromWord = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(tRank << 2), (int)giverOfRank);
                const uint32_t from = fromWord & 63u;
                const bool taking = !hasRay && (fromWord & 0x80u) != 0u, giving = canGive && gRank < nTake;
                const int at = (int)(from << 2);
                const float gox = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)__float_as_uint(o.x)));
                const float goy = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)__float_as_uint(o.y)));
                const float goz = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)__float_as_uint(o.z)));
                const float gdx = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)__float_as_uint(d.x)));
                const float gdy = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)__float_as_uint(d.y)));
                const float gdz = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)__float_as_uint(d.z)));
                const float gt = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)__float_as_uint(h.t)));
                const float gu = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)__float_as_uint(h.u)));
                const float gv = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)__float_as_uint(h.v)));
                const uint32_t gleaf = (uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)h.leaf);
                const uint32_t gfirst = (uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)firstHit);
                const uint32_t level = (uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)share) & 63u;
                if (taking) {
                    o = rt_mk3(gox, goy, goz);
                    d = rt_mk3(gdx, gdy, gdz);
                    r = rt_mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                    slab = rt_ray_slab(sv.grid, o, r);
                    h.t = gt; h.u = gu; h.v = gv; h.leaf = gleaf;
                    firstHit = gfirst;
                    uint32_t *entry = lds + level * TRV_THREADS + (threadIdx.x & ~63u) + from;
                    w.node = *entry;
                    *entry = TRV_DONE;
                    stk[0] = TRV_DONE;
                    w.sp = 1;
                    w.tos = TRV_DONE;
                    w.npend = 0;
                    hasRay = true;
                    share = 1u | (from << 16) | (1u << 24);
                }
                if (giving) share += 1u + (1u << 8);
#ifdef TRV_SHARE_STATS
                shGiven += (unsigned long long)__popcll(__ballot(giving));
#endif
            }
        }
        TRV_SHARE_CLK(0)
        if (__ballot(hasRay) == 0ull) break;
        TRV_PHASE(0)
#ifdef TRV_PROFILE
        pN[0]++;
#endif

        // ---- node phase, in the variant for the way the wave's rays run
        {
            const unsigned long long in = __ballot(w.node != TRV_DONE);
            const unsigned long long sx = __ballot(slab.A[0] < 0.0f) & in, sy = __ballot(slab.A[1] < 0.0f) & in, sz = __ballot(slab.A[2] < 0.0f) & in;
            const bool uniform = (sx == 0ull || sx == in) && (sy == 0ull || sy == in) && (sz == 0ull || sz == in);
            const uint32_t oct = uniform ? ((sx ? 1u : 0u) | (sy ? 2u : 0u) | (sz ? 4u : 0u)) : 8u;
            constexpr bool NEAR = true;                              // every kind of ray gains: a hit found sooner, or a closer one
            const float limit = h.t * sv.pruneScale;                 // +inf without a hit, or with the option off
            // front to back, with the boxes beyond `limit` left out, where that can end a walk sooner; as the boxes come otherwise
            const bool ordered = COUNT || Src::FIRST_HIT_MAY_DECIDE || sv.pruneScale < rt_inf();
            bool bounded = false;
            // (camera, caller and bounce rays: where the 1.5 % rule is asked for it goes through the same phase, with the
            //  support distance in place of the entry distance -- one ordered phase per kernel instead of two)
            if constexpr (!Src::FIRST_HIT_MAY_DECIDE) bounded = sv.wnodes != nullptr && (sv.boundQ > 0.0f || sv.pruneScale < rt_inf() || COUNT);
            if (bounded) {
                if constexpr (!Src::FIRST_HIT_MAY_DECIDE) {
                // ---- leaving out boxes, with a bound.  For a triangle the reference accepts (determinant >= 1e-8,
                // u, v, u + v in range -- all as computed), the point o + t'd of the COMPUTED distance t' lies within
                //     |r| <= 28.9 u |o - v0| |e1| |e2| |d| / determinant <= RT_BOUND_C |e1||e2| |o - v0| |d|      (u = 2^-24)
                // of the point v0 + u'e1 + v'e2 of the triangle that the computed u', v' name: t', u', v' are Cramer
                // quotients of three numerators and one determinant, each a dot product with a cross product (a cross
                // product's components are off by 2u |a||b| each, 3.47u in norm, a three-term dot product by 3u |a||b|,
                // o - v0 by u): each numerator is off by at most 7.47u |o - v0||e1||e2||d|, the determinant by 6.47u
                // |e1||e2||d|, and M (t', u', v') - (o - v0) with M = [-d, e1, e2] collects exactly those errors over the
                // determinant: 6.47 + 3 x 7.47 = 28.9.  (u' >= 0, v' >= 0 and fl(u' + v') <= 1 put that point in the
                // triangle up to a rounding of u' + v' and of the stored edges -- parts in 10^8 of an edge, which the
                // 0.001 by which every triangle's box is padded absorbs.)  The triangle's point is in
                // every box that holds the triangle, so o + t'd is within |r| of the box: on each axis a, t' is at
                // least (the ray's near value on a) - |r| / |d[a]|.  A triangle that could still change the answer has
                // t' <= the closest hit t, so its v0 is within t|d| + |r| + (|e1| + |e2|) of o; with
                // Q = RT_BOUND_C max|e1||e2| and E = max(|e1| + |e2|) over the scene's SMALL triangles and q = Q|d|
                // that gives |r| <= |d| (q t + Q E) / (1 - q) =: |d| rho, and
                //     max over a of (near value on a - rho |d| / |d[a]|) > t   =>   t' > t for every small triangle in
                // the box: nothing in it can replace the hit or tie with it.  Boxes that hold a LARGE triangle carry a
                // flag and are never left out.  (The near values are lower bounds that include their own roundings;
                // rho is rounded up with room for the few operations that make it and for the second-order term
                // 2 ulp (t + E / |d|) of the quotients' own roundings.)
                // The 1.5 % rule (RT_OPT_PRUNE_BY_DISTANCE 2) goes through the same phase with rho = 0 and the room in
                // the limit: near values as they are, large triangles or not.
                float blimit = limit, b0 = 0.0f, b1 = 0.0f, b2 = 0.0f;
                uint32_t largeBit = 0u;
                if (sv.boundQ > 0.0f) {
                    const float dl = rt_sqrt(rt_dot3(d, d));
                    const float q = sv.boundQ * dl;
                    const float rho = q < 0.5f ? ((q * h.t + sv.boundQ * sv.boundE) / (1.0f - q) + 9.5367431640625e-7f * (h.t + sv.boundE / dl)) * 1.0000153f : rt_inf();
                    const float rd = rho * dl * 1.0000153f;
                    b0 = rd * __builtin_fabsf(r.x) * 1.0000153f;
                    b1 = rd * __builtin_fabsf(r.y) * 1.0000153f;
                    b2 = rd * __builtin_fabsf(r.z) * 1.0000153f;
                    blimit = h.t;
                    largeBit = RT_REF_LARGE;
                }
                switch (oct) {
                case 0: node_phase_wide<0, COUNT, true, true>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, blimit, shareTrips, b0, b1, b2, largeBit); break;
                case 1: node_phase_wide<1, COUNT, true, true>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, blimit, shareTrips, b0, b1, b2, largeBit); break;
                case 2: node_phase_wide<2, COUNT, true, true>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, blimit, shareTrips, b0, b1, b2, largeBit); break;
                case 3: node_phase_wide<3, COUNT, true, true>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, blimit, shareTrips, b0, b1, b2, largeBit); break;
                case 4: node_phase_wide<4, COUNT, true, true>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, blimit, shareTrips, b0, b1, b2, largeBit); break;
                case 5: node_phase_wide<5, COUNT, true, true>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, blimit, shareTrips, b0, b1, b2, largeBit); break;
                case 6: node_phase_wide<6, COUNT, true, true>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, blimit, shareTrips, b0, b1, b2, largeBit); break;
                case 7: node_phase_wide<7, COUNT, true, true>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, blimit, shareTrips, b0, b1, b2, largeBit); break;
                default: node_phase_wide<8, COUNT, true, true>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, blimit, shareTrips, b0, b1, b2, largeBit); break;
                }
                }
            } else if (Src::FIRST_HIT_MAY_DECIDE && sv.wnodes != nullptr && ordered) {
                switch (oct) {
                case 0: node_phase_wide<0, COUNT, true>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, limit, shareTrips); break;
                case 1: node_phase_wide<1, COUNT, true>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, limit, shareTrips); break;
                case 2: node_phase_wide<2, COUNT, true>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, limit, shareTrips); break;
                case 3: node_phase_wide<3, COUNT, true>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, limit, shareTrips); break;
                case 4: node_phase_wide<4, COUNT, true>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, limit, shareTrips); break;
                case 5: node_phase_wide<5, COUNT, true>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, limit, shareTrips); break;
                case 6: node_phase_wide<6, COUNT, true>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, limit, shareTrips); break;
                case 7: node_phase_wide<7, COUNT, true>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, limit, shareTrips); break;
                default: node_phase_wide<8, COUNT, true>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, limit, shareTrips); break;
                }
            } else if (!COUNT && !Src::FIRST_HIT_MAY_DECIDE && sv.wnodes != nullptr) {
                switch (oct) {
                case 0: node_phase_wide<0, false, false>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, limit, shareTrips); break;
                case 1: node_phase_wide<1, false, false>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, limit, shareTrips); break;
                case 2: node_phase_wide<2, false, false>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, limit, shareTrips); break;
                case 3: node_phase_wide<3, false, false>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, limit, shareTrips); break;
                case 4: node_phase_wide<4, false, false>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, limit, shareTrips); break;
                case 5: node_phase_wide<5, false, false>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, limit, shareTrips); break;
                case 6: node_phase_wide<6, false, false>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, limit, shareTrips); break;
                case 7: node_phase_wide<7, false, false>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, limit, shareTrips); break;
                default: node_phase_wide<8, false, false>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, limit, shareTrips); break;
                }
            } else
            switch (oct) {
            case 0: node_phase<0, COUNT, NEAR>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, limit, shareTrips TRV_SHARE_STATS_ARG); break;
            case 1: node_phase<1, COUNT, NEAR>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, limit, shareTrips TRV_SHARE_STATS_ARG); break;
            case 2: node_phase<2, COUNT, NEAR>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, limit, shareTrips TRV_SHARE_STATS_ARG); break;
            case 3: node_phase<3, COUNT, NEAR>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, limit, shareTrips TRV_SHARE_STATS_ARG); break;
            case 4: node_phase<4, COUNT, NEAR>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, limit, shareTrips TRV_SHARE_STATS_ARG); break;
            case 5: node_phase<5, COUNT, NEAR>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, limit, shareTrips TRV_SHARE_STATS_ARG); break;
            case 6: node_phase<6, COUNT, NEAR>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, limit, shareTrips TRV_SHARE_STATS_ARG); break;
            case 7: node_phase<7, COUNT, NEAR>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, limit, shareTrips TRV_SHARE_STATS_ARG); break;
            default: node_phase<8, COUNT, NEAR>(sv, slab, w, stk, pend, spill, spillLanes, cnt, feed.exhausted, limit, shareTrips TRV_SHARE_STATS_ARG); break;
            }
        }

        // ---- rays that do not fit the fast walk (a direction component that is zero or not finite, a slab the
        //      half-float test cannot bound, a stack that ran out) take the reference's own walk, whole
        if (__builtin_expect(hasRay && w.node == TRV_DONE && w.tos == TRV_REDO, 0)) {
            h.t = rt_inf(); h.u = 0.0f; h.v = 0.0f; h.leaf = RT_INVALID;
            exact_walk<COUNT>(sv, o, d, r, h, cnt);
            firstHit = 2u;
            w.tos = TRV_DONE;
            w.npend = 0;
        }

        TRV_PHASE(1)
        TRV_SHARE_CLK(1)
        // ---- triangle phase: every lane tests its queued triangles in the order it met them
        const uint32_t npend = w.npend;
        if (COUNT) {
            for (uint32_t k = 0; __ballot(k < npend) != 0ull; ++k)
                if (k < npend) {
                    const bool boxed = tri_test<COUNT>(sv, pend[k * TRV_THREADS] & RT_REF_INDEX, o, d, r, h);   // (the two-level walk queues references as they are, flag bits and all)
                    cnt[1]++;
                    cnt[2] += boxed ? 1u : 0u;
                }
        } else {
            // Two triangles per trip: both records are requested before either is looked at, so a lane with
            // several queued triangles pays the memory round trip once per pair (the tests themselves are
            // short; the trip time is the latency of the 48-byte fetch).  Lanes with nothing queued fetch
            // record 0 (one cache line for the whole wave) and skip the test.
            // The phase ends when fewer than TRV_TRI_MIN_LANES lanes still have triangles queued (after one trip at
            // least): what is left -- the triangles those lanes met last, the farthest ones -- stays queued, moved to the
            // front of the list, and is tested in the next turn of the main loop together with what the node phase adds;
            // a ray retires when its walk is over AND its list is empty.  (No result depends on when a queued triangle
            // is tested.)
            uint32_t k = 0;
            for (;; k += 2u) {
                const uint32_t nLeft = lane_count(__ballot(k < npend));
                if (nLeft == 0u || (k != 0u && nLeft < TRV_TRI_MIN_LANES)) break;
#ifdef TRV_PROFILE
                pN[3]++;
                pN[4] += (unsigned long long)nLeft;
#endif
                const bool t0 = k < npend, t1 = k + 1u < npend;
                const uint32_t leaf0 = t0 ? pend[k * TRV_THREADS] & RT_REF_INDEX : 0u;
                const uint32_t leaf1 = t1 ? pend[(k + 1u) * TRV_THREADS] & RT_REF_INDEX : 0u;
                const float4 *p0 = (const float4 *)(sv.leafTri + leaf0), *p1 = (const float4 *)(sv.leafTri + leaf1);
                const float4 a0 = p0[0], b0 = p0[1], c0 = p0[2];
                const float4 a1 = p1[0], b1 = p1[1], c1 = p1[2];
                if (t0) {
                    RtHitRec cand = h;
                    if (tri_test_values(rt_mk3(a0.x, a0.y, a0.z), rt_mk3(a0.w, b0.x, b0.y), rt_mk3(b0.z, b0.w, c0.x), leaf0, o, d, cand, sv.leafRank)) {
                        const float4 e = p0[3];
                        if (box_hit(c0.z, c0.w, e.x, e.y, e.z, e.w, o, d, r)) h = cand;
                    }
                }
                if (t1) {
                    RtHitRec cand = h;
                    if (tri_test_values(rt_mk3(a1.x, a1.y, a1.z), rt_mk3(a1.w, b1.x, b1.y), rt_mk3(b1.z, b1.w, c1.x), leaf1, o, d, cand, sv.leafRank)) {
                        const float4 e = p1[3];
                        if (box_hit(c1.z, c1.w, e.x, e.y, e.z, e.w, o, d, r)) h = cand;
                    }
                }
            }
            // what stays queued moves to the front of its list (few lanes, few entries)
            const uint32_t left = npend > k ? npend - k : 0u;
            for (uint32_t m = 0; __ballot(m < left) != 0ull; ++m)
                if (m < left) pend[m * TRV_THREADS] = pend[(k + m) * TRV_THREADS];
            w.npend = left;
        }
        if (COUNT) w.npend = 0;
        if (Src::FIRST_HIT_MAY_DECIDE && hasRay && h.leaf != RT_INVALID && (w.node != TRV_DONE || w.npend != 0u) && firstHit != 2u) {
            if (firstHit == 0u) firstHit = src.first_hit_decides(o, d, r) ? 1u : 2u;
            if (firstHit == 1u) {                              // retired below
                w.node = TRV_DONE;
                w.npend = 0;
            }
        }
        TRV_PHASE(2)
        TRV_SHARE_CLK(2)

        // ---- subtrees that were taken over and are done: the hit goes back to the lane that gave the subtree away,
        //      which keeps it if it is closer than its own, or as close and earlier in the reference's order
        if (sharing && feed.exhausted) {
            for (;;) {
                const unsigned long long fin = __ballot(hasRay && w.node == TRV_DONE && w.npend == 0u && (share & 0x01007F00u) == 0x01000000u);
                if (fin == 0ull) break;
                const uint32_t T = (uint32_t)__ffsll((long long)fin) - 1u;                 // wave-uniform
                const uint32_t giver = (lane_value(share, T) >> 16) & 63u;
                const float ct = lane_value(h.t, T), cu = lane_value(h.u, T), cv = lane_value(h.v, T);
                const uint32_t cleaf = lane_value(h.leaf, T);
                if (lane == giver) {
                    if (cleaf != RT_INVALID && cleaf != h.leaf) {
                        bool closer = ct < h.t;
                        if (ct == h.t && h.leaf != RT_INVALID) closer = sv.leafRank[cleaf] < sv.leafRank[h.leaf];
                        if (closer) { h.t = ct; h.u = cu; h.v = cv; h.leaf = cleaf; }
                    }
                    share -= 1u << 8;
                }
                if (lane == T) {
                    hasRay = false;
                    share = 1u;
                }
            }
        }
        // ---- retire finished rays (one that gave subtrees away waits for them)
        if (hasRay && w.node == TRV_DONE && w.npend == 0u && (share & 0x01007F00u) == 0u) {
            src.store(rayIdx, h);
            if (COUNT && h.leaf != RT_INVALID) nHits++;
            hasRay = false;
        }
        TRV_PHASE(3)
    }
#ifdef TRV_SHARE_STATS
    if (lane == 0) {             // [28] hand-overs, [29] waves, [30] sum and [31] maximum of a wave's clocks after the feed ran dry
        const unsigned long long tail = shT0 ? clock64() - shT0 : 0ull;
        atomicAdd(&sv.stats[28], shGiven);
        atomicAdd(&sv.stats[29], 1ull);
        atomicAdd(&sv.stats[30], tail);
        atomicMax(&sv.stats[31], tail);
        // of the wave that went on longest: its node-phase trips, the busy lanes summed over them, main-loop trips and hand-overs
        atomicMax(&sv.stats[27], (tail >> 8) << 32 | (shStat[0] & 0xFFFFFFFFull));
        atomicMax(&sv.stats[26], (tail >> 8) << 32 | (shStat[1] & 0xFFFFFFFFull));
        atomicMax(&sv.stats[25], (tail >> 8) << 32 | (shIters & 0xFFFFull) << 16 | (shGiven & 0xFFFFull));
        for (int k = 0; k < 4; ++k) atomicMax(&sv.stats[21 + k], (tail >> 8) << 32 | (shClk[k] & 0xFFFFFFFFull));
    }
#endif
#ifdef TRV_PROFILE
    if (lane == 0) {
        for (int k = 0; k < 4; ++k) atomicAdd(&sv.stats[16 + k], pT[k]);
        for (int k = 0; k < 6; ++k) atomicAdd(&sv.stats[20 + k], pN[k]);
    }
#endif
    if (COUNT) flush_counters(sv, nRays, cnt, nHits);
}

// ---------------------------------------------------------------------------
// Packet traversal: 16 consecutive rays of the source (rays 16k .. 16k+15) walk
// the tree together in one quarter of a wave.  It is meant for rays that start
// close together and point the same way (the 16 soft-shadow rays of a pixel,
// camera rays of neighbouring pixels): the packet fetches a node once, with one
// coalesced 64-byte read spread over its 16 lanes, and hands the words round
// with row-broadcast DPP moves, instead of 16 lanes each fetching 64 bytes.
//
// Every ray still takes exactly the decisions of its own traversal: a stack
// entry carries the 16-bit set of rays whose own box test let them into that
// node, a ray tests a child box or a triangle only where it is in the set, and
// the packet walks the tree in the reference's depth-first order, so each ray
// meets its candidates in the order its own walk would (strict '<' keeps the
// first).  The packet only adds node visits that some other ray of the packet
// needed.
#define PKT_WIDTH 16u
#define PKT_STACK_WORDS (2 * RT_STACK_DEPTH)                       // (node, ray set) per level
#define PKT_LDS_WORDS (PKT_STACK_WORDS * (PKT_THREADS / PKT_WIDTH))

template <int J>
__device__ __forceinline__ uint32_t row_bcast(uint32_t x)           // lane J of each 16-lane row -> the whole row
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x150 + J, 0xF, 0xF, false);
}
template <int J>
__device__ __forceinline__ float row_bcastf(uint32_t x) { return __uint_as_float(row_bcast<J>(x)); }

// 16-bit ballot of this lane's row (all 16 lanes of the row must be active)
__device__ __forceinline__ uint32_t row_ballot(bool p, uint32_t rowShift)
{
    return (uint32_t)(__ballot(p) >> rowShift) & 0xFFFFu;
}

// the packet's rays in `want` test sorted leaf `leaf`
__device__ __forceinline__ void tri_test_packet(const RtSceneView &sv, uint32_t leaf, bool want, uint32_t sub, const rt_f3 o,
                                                const rt_f3 d, RtHitRec &h)
{
    const uint32_t w = ((const uint32_t *)(sv.leafTri + leaf))[sub < 9u ? sub : 8u];   // v0, e1, e2: 36 bytes
    const rt_f3 v0 = rt_mk3(row_bcastf<0>(w), row_bcastf<1>(w), row_bcastf<2>(w));
    const rt_f3 e1 = rt_mk3(row_bcastf<3>(w), row_bcastf<4>(w), row_bcastf<5>(w));
    const rt_f3 e2 = rt_mk3(row_bcastf<6>(w), row_bcastf<7>(w), row_bcastf<8>(w));
    if (want) tri_test_values(v0, e1, e2, leaf, o, d, h);
}

// lds: PKT_LDS_WORDS words per block.
template <bool COUNT, class Src>
__device__ __forceinline__ void traverse_packets(const RtSceneView &sv, Src &src, uint32_t *__restrict__ cursor, uint32_t *lds)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t sub = lane & 15u, rowShift = lane & 48u;
    uint32_t *stk = lds + (threadIdx.x >> 4) * PKT_STACK_WORDS;
    const uint32_t total = src.count();

    // packet state (the same value in all 16 lanes of a row)
    uint32_t node = TRV_DONE, set = 0, sp = 0;
    bool exhausted = false;
    // ray state
    bool hasRay = false;
    uint32_t rayIdx = 0;
    rt_f3 o = rt_splat3(0.0f), d = rt_splat3(0.0f), r = rt_splat3(0.0f);
    RtHitRec h;
    h.t = rt_inf(); h.u = 0.0f; h.v = 0.0f; h.leaf = RT_INVALID;
    uint32_t cnt[3] = {0, 0, 0};
    uint32_t nRays = 0, nHits = 0;

    for (;;) {
        if (node == TRV_DONE && !exhausted) {
            // ---- retire the finished packet, fetch the next one
            if (hasRay) {
                src.store(rayIdx, h);
                if (COUNT && h.leaf != RT_INVALID) nHits++;
                hasRay = false;
            }
            uint32_t base = 0;
            if (sub == 0) base = atomicAdd(cursor, PKT_WIDTH);
            base = row_bcast<0>(base);
            if (base >= total) {
                exhausted = true;
            } else {
                rayIdx = base + sub;
                hasRay = rayIdx < total;
                sp = 0;
                bool in = false;
                if (hasRay) {
                    src.load(rayIdx, o, d);
                    r = rt_mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                    h.t = rt_inf(); h.u = 0.0f; h.v = 0.0f; h.leaf = RT_INVALID;
                    if (COUNT) nRays++;
                    if (sv.nTriangles != 0) {
                        if (COUNT) cnt[0]++;
                        const RtAABB &rb = sv.rootBox;
                        in = sv.nTriangles >= 2 && box_hit(rb.min[0], rb.min[1], rb.min[2], rb.max[0], rb.max[1], rb.max[2], o, d, r);
                    }
                }
                set = row_ballot(in, rowShift);
                node = set ? 0u : TRV_DONE;
            }
        }
        if (__ballot(node != TRV_DONE || !exhausted) == 0ull) break;

        if (node != TRV_DONE) {
            // ---- one node for the whole packet: 16 lanes x 4 bytes, then row broadcasts
            const uint32_t w = ((const uint32_t *)(sv.nodes + node))[sub];
            const uint32_t l = row_bcast<12>(w), rr = row_bcast<13>(w);
            const bool lLeaf = (l & RT_LEAF_BIT) != 0, rLeaf = (rr & RT_LEAF_BIT) != 0;
            const bool act = hasRay && ((set >> sub) & 1u) != 0;
            bool hitL = false, hitR = false;
            {
                const float a0 = row_bcastf<0>(w), a1 = row_bcastf<1>(w), a2 = row_bcastf<2>(w);
                const float a3 = row_bcastf<3>(w), a4 = row_bcastf<4>(w), a5 = row_bcastf<5>(w);
                const float b0 = row_bcastf<6>(w), b1 = row_bcastf<7>(w), b2 = row_bcastf<8>(w);
                const float b3 = row_bcastf<9>(w), b4 = row_bcastf<10>(w), b5 = row_bcastf<11>(w);
                if (act) {
                    hitL = box_hit(a0, a1, a2, a3, a4, a5, o, d, r);
                    hitR = box_hit(b0, b1, b2, b3, b4, b5, o, d, r);
                }
            }
            if (COUNT && act) {
                cnt[0] += (lLeaf ? 0u : 1u) + (rLeaf ? 0u : 1u);
                cnt[1] += (lLeaf ? 1u : 0u) + (rLeaf ? 1u : 0u);
                cnt[2] += ((lLeaf && hitL) ? 1u : 0u) + ((rLeaf && hitR) ? 1u : 0u);
            }
            const uint32_t setL = row_ballot(hitL, rowShift), setR = row_ballot(hitR, rowShift);
            if (lLeaf && setL) tri_test_packet(sv, l & ~RT_LEAF_BIT, hitL, sub, o, d, h);
            if (rLeaf && setR) tri_test_packet(sv, rr & ~RT_LEAF_BIT, hitR, sub, o, d, h);
            const bool goL = !lLeaf && setL, goR = !rLeaf && setR;
            if (goR) {
                if (goL) {
                    if (sp >= RT_STACK_DEPTH) { *sv.errorFlag = 1u; node = TRV_DONE; }
                    else {
                        stk[2 * sp] = l;           // all 16 lanes store the same two words
                        stk[2 * sp + 1] = setL;
                        sp++;
                        node = rr;
                        set = setR;
                    }
                } else {
                    node = rr;
                    set = setR;
                }
            } else if (goL) {
                node = l;
                set = setL;
            } else if (sp == 0) {
                node = TRV_DONE;
            } else {
                sp--;
                node = stk[2 * sp];
                set = stk[2 * sp + 1];
            }
        }
    }
    if (COUNT) flush_counters(sv, nRays, cnt, nHits);
}

#endif
