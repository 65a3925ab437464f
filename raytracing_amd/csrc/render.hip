// render.hip -- BVH traversal and the wavefront (queue-based) path tracer.
//
// Reference path being replaced (same results on the same inputs and seeds):
//   Assets/Shaders/BVH/BVHRayTracing.cginc:27-133        traversal (rt_traverse)
//   Assets/Shaders/RayTracing/BruteForceRayTracing.cginc:166-204  Moller-Trumbore
//   Assets/Shaders/RayTracing/RayTracing.compute:44-91   CSMain  -> k_generate / k_traverse<QueueSrc> / k_shade
//   Assets/Shaders/RayTracing/RayTracing.compute:96-131  ShadowMap -> k_sm_generate / k_sm_setup / k_traverse<ShadowSrc> / k_sm_resolve
//   Assets/Shaders/RayTracing/AddShader.shader:47-79     accumulate -> k_accumulate
//
// The reference runs one thread per pixel through all 8 bounces.  Here a frame
// is a sequence of launches over compacted queues of live paths:
//   generate -> { extend (traverse) -> shade (+ wave64 ballot compaction) } x 8
// Path state lives in SoA arrays indexed by tile-local slot; queues hold slots.
#include <new>
#include <utility>

#include "rt_internal.h"
#include "rt_shade.h"

#include <algorithm>
#include "rt_traverse.h"
#include "rt_brute.h"

// Queue compaction.  A block works through a tile of Q_TILE entries and leaves in
// stage[] (LDS) the kept slot or RT_INVALID for every entry, in tile order; the
// kept ones are appended with ONE global atomic per block (wave64 ballots and
// prefix counts inside the block).
// (Atomics on one address are served at ~10 ns each whatever the number of CUs
// issuing them, so a launch of 16 M threads cannot afford one per wave.)
#define Q_ITEMS 8
#define Q_THREADS 256
#define Q_TILE (Q_ITEMS * Q_THREADS)
template <class Emit>
__device__ __forceinline__ void block_queue_flush(const uint32_t *stage, uint32_t *__restrict__ count, Emit emit)
{
    __shared__ uint32_t wsum[Q_THREADS / 64];
    __shared__ uint32_t gbase;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    __syncthreads();
    uint32_t v[Q_ITEMS], pre[Q_ITEMS], run = 0;
#pragma unroll
    for (int it = 0; it < Q_ITEMS; ++it) {
        v[it] = stage[it * Q_THREADS + tid];
        const unsigned long long m = __ballot(v[it] != RT_INVALID);
        pre[it] = run + (uint32_t)__popcll(m & lt);
        run += (uint32_t)__popcll(m);
    }
    if (lane == 0) wsum[wave] = run;
    __syncthreads();
    if (tid == 0) {
        uint32_t total = 0;
        for (int w = 0; w < Q_THREADS / 64; ++w) total += wsum[w];
        gbase = total ? atomicAdd(count, total) : 0u;
    }
    __syncthreads();
    uint32_t base = gbase;
    for (uint32_t w = 0; w < wave; ++w) base += wsum[w];
#pragma unroll 1
    for (int it = 0; it < Q_ITEMS; ++it)
        if (v[it] != RT_INVALID) emit(v[it], base + pre[it]);     // (what the entry stood for, its place in the queue)
}

// A queue entry IS the ray: 32 bytes in queue order, so that the walker's refill reads consecutive records
// instead of gathering origin and direction through a slot number out of per-path arrays tens of gigabytes long.
//   r0 = (origin.xyz, direction.x)   r1 = (direction.y, direction.z, path slot as bits, 0)
// Hit records are kept by queue position as well (the shading pass walks the queue in the same order).
__device__ __forceinline__ void ray_record_store(float4 *__restrict__ ray, uint32_t at, const rt_f3 o, const rt_f3 d, uint32_t slot)
{
    ray[2 * (size_t)at] = make_float4(o.x, o.y, o.z, d.x);
    ray[2 * (size_t)at + 1] = make_float4(d.y, d.z, __uint_as_float(slot), 0.0f);
}

// ------------------------------------------------------------------ ray sources for traverse_rays
struct ArraySrc {            // caller rays (rt_trace_rays)
    static constexpr bool FIRST_HIT_MAY_DECIDE = false;
    __device__ __forceinline__ bool first_hit_decides(const rt_f3, const rt_f3, const rt_f3) const { return false; }
    const float4 *org, *dir;
    uint32_t n;
    RtHitRec *hits;
    __device__ __forceinline__ uint32_t count() const { return n; }
    __device__ __forceinline__ void load(uint32_t i, rt_f3 &o, rt_f3 &d) const
    {
        const float4 o4 = org[i], d4 = dir[i];
        o = rt_mk3(o4.x, o4.y, o4.z);
        d = rt_mk3(d4.x, d4.y, d4.z);
    }
    __device__ __forceinline__ void store(uint32_t i, const RtHitRec &h) const { hits[i] = h; }
};

struct QueueSrc {            // live paths of one bounce
    static constexpr bool FIRST_HIT_MAY_DECIDE = false;
    __device__ __forceinline__ bool first_hit_decides(const rt_f3, const rt_f3, const rt_f3) const { return false; }
    const float4 *ray;         // 2 x float4 per queue entry
    const uint32_t *qcount;
    RtHitRec *hit;             // by queue position
    __device__ __forceinline__ uint32_t count() const { return *qcount; }
    __device__ __forceinline__ void load(uint32_t i, rt_f3 &o, rt_f3 &d) const
    {
        const float4 r0 = ray[2 * (size_t)i], r1 = ray[2 * (size_t)i + 1];
        o = rt_mk3(r0.x, r0.y, r0.z);
        d = rt_mk3(r0.w, r1.x, r1.y);
    }
    __device__ __forceinline__ void store(uint32_t i, const RtHitRec &h) const { hit[i] = h; }
};

// The 16 soft-shadow rays of every shaded pixel (Shadow.cginc:95-111): ray
// i = (pixel i/16, disk sample i%16); result = "closest hit casts a shadow".
__constant__ float c_blue_noise[16][2] = {   // first 16 entries of BLUE_NOISE_IN_DISK (Shadow.cginc:7-72)
    {0.478712f, 0.875764f},  {-0.337956f, -0.793959f}, {-0.955259f, -0.028164f}, {0.864527f, 0.325689f},
    {0.209342f, -0.395657f}, {-0.106779f, 0.672585f},  {0.156213f, 0.235113f},   {-0.413644f, -0.082856f},
    {-0.415667f, 0.323909f}, {0.141896f, -0.939980f},  {0.954932f, -0.182516f},  {-0.766184f, 0.410799f},
    {-0.434912f, -0.458845f}, {0.415242f, -0.078724f}, {0.728335f, -0.491777f},  {-0.058086f, -0.066401f}};

// Direction of soft-shadow ray k of a pixel (Shadow.cginc:95-104): the blue-noise disk sample k, turned by the
// pixel's random angle, scaled to the light's radius, laid round the light direction.
struct ShadowFrame {
    rt_f3 lightDir, lightTangent, lightBitangent;
    float lightRadius;
    // Box round the own (padded) boxes of the triangles that DO cast a shadow (min > max: none).  A ray that fails
    // the reference's slab test on it is accepted by none of them -- the test is monotone in the box -- so its
    // closest hit, if it has one, casts no shadow: the answer is "lit" without a walk.
    float casterMin[3], casterMax[3];
    __device__ __forceinline__ rt_f3 direction(uint32_t k, float cosTheta, float sinTheta) const
    {
        const float sx = c_blue_noise[k][0], sy = c_blue_noise[k][1];
        float dx = sx * cosTheta - sy * sinTheta;
        float dy = sx * sinTheta + sy * cosTheta;
        dx *= lightRadius;
        dy *= lightRadius;
        return rt_normalize3(rt_add3(rt_add3(lightDir, rt_scale3(lightTangent, dx)), rt_scale3(lightBitangent, dy)));
    }
    // A cheap sufficient test for "none of the 16 rays from o can meet a caster", in plain geometry: every ray
    // direction lies within the angle atan(coneTan) of the central direction, so a point of such a ray inside the
    // caster box, at most D from o (D = distance to the farthest corner), is within D * coneTan of the central ray; if
    // the central ray misses the box grown by that much on every side, no ray of the cone touches the box.  The
    // growth is taken 0.2 % of D larger than needed, thousands of times what rounding in the exact per-ray test
    // can move a graze.  coneTan < 0: no such test (degenerate light direction or radius).
    rt_f3 coneAxis;
    float coneTan;
    __device__ __forceinline__ bool cone_misses_the_casters(const rt_f3 o) const
    {
        if (!(coneTan >= 0.0f)) return false;
        if (casterMin[0] > casterMax[0]) return true;
        const float oc[3] = {o.x, o.y, o.z}, dc[3] = {coneAxis.x, coneAxis.y, coneAxis.z};
        float far2 = 0.0f;
        for (int a = 0; a < 3; ++a) {
            const float e = rt_max(__builtin_fabsf(casterMin[a] - oc[a]), __builtin_fabsf(casterMax[a] - oc[a]));
            far2 += e * e;
        }
        const float D = __builtin_sqrtf(far2);
        const float grow = D * coneTan + D * 2.0e-3f + 1.0e-6f;
        float tn = 0.0f, tf = rt_inf();
        for (int a = 0; a < 3; ++a) {
            const float lo = (casterMin[a] - grow) - oc[a], hi = (casterMax[a] + grow) - oc[a];
            if (__builtin_fabsf(dc[a]) < 1.0e-12f) {
                if (lo > 0.0f || hi < 0.0f) return true;
            } else {
                const float t1 = lo / dc[a], t2 = hi / dc[a];
                tn = rt_max(tn, rt_min(t1, t2));
                tf = rt_min(tf, rt_max(t1, t2));
            }
        }
        return tn > tf;           // any NaN on the way makes this false: not a sure miss, the 16 rays are tested
    }
    __device__ __forceinline__ bool may_meet_a_caster(const rt_f3 o, const rt_f3 d) const
    {
        if (casterMin[0] > casterMax[0]) return false;
        if (rt_ray_needs_exact_walk(o, d)) return true;
        const rt_f3 r = rt_mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);     // box_hit: the reference's boolean, divisions only on a graze
        return box_hit(casterMin[0], casterMin[1], casterMin[2], casterMax[0], casterMax[1], casterMax[2], o, d, r);
    }
};

struct ShadowSrc {
    const uint32_t *smqCount;
    const float4 *rec;         // per shaded pixel, in queue order: (origin.xyz, cosTheta), (sinTheta, alpha, path slot, .)
    uint8_t *flags;            // per shaded pixel x 16, in queue order: a ray's flag is flags[ray index]
    const RtLeafShade *leafShade;
    ShadowFrame frame;
    // All a shadow ray reports is whether its closest hit casts a shadow.  A ray that no NON-casting triangle
    // can accept only meets casters, so its first accepted hit settles the answer and the walk ends there.
    // Asked once per ray, when it first holds a hit: the ray is tested against the box round the own boxes
    // of the non-casting triangles (ncMin > ncMax: there are none) the way the reference tests a triangle's own
    // box before the triangle -- that test is monotone in the box, a ray that passes the box of such a
    // triangle passes this one -- and, where it passes and the non-casters are few (a ground quad), against
    // each of them with the acceptance test itself.
    static constexpr bool FIRST_HIT_MAY_DECIDE = true;
    float ncMin[3], ncMax[3];
    const float4 *ncTri;       // records of the non-casters, leafTri layout
    uint32_t nNc;              // how many; above RT_NONCASTER_LIST the box alone decides
    __device__ __forceinline__ bool first_hit_decides(const rt_f3 o, const rt_f3 d, const rt_f3 r) const
    {
        if (ncMin[0] > ncMax[0]) return true;
        if (!box_hit(ncMin[0], ncMin[1], ncMin[2], ncMax[0], ncMax[1], ncMax[2], o, d, r)) return true;
        if (nNc > RT_NONCASTER_LIST) return false;
        for (uint32_t k = 0; k < nNc; ++k) {
            const float4 a = ncTri[k * 4], b = ncTri[k * 4 + 1], c = ncTri[k * 4 + 2], e = ncTri[k * 4 + 3];
            RtHitRec cand;
            cand.t = rt_inf(); cand.u = 0.0f; cand.v = 0.0f; cand.leaf = RT_INVALID;
            if (tri_test_values(rt_mk3(a.x, a.y, a.z), rt_mk3(a.w, b.x, b.y), rt_mk3(b.z, b.w, c.x), 0u, o, d, cand) &&
                box_hit(c.z, c.w, e.x, e.y, e.z, e.w, o, d, r))
                return false;
        }
        return true;
    }
    __device__ __forceinline__ uint32_t count() const { return *smqCount * 16u; }
    __device__ __forceinline__ void load(uint32_t i, rt_f3 &o, rt_f3 &d) const
    {
        const uint32_t at = i >> 4, k = i & 15u;                  // the pixel's place in the queue: no look-up on the way to the record
        const float4 a = rec[(size_t)at * 2], b = rec[(size_t)at * 2 + 1];
        o = rt_mk3(a.x, a.y, a.z);
        d = frame.direction(k, a.w, b.x);
    }
    // (k_sm_setup leaves out the pixels none of whose 16 rays can meet a caster.  Asking the same of every ray of the
    //  other pixels when it is fetched cost more than the few walks it saved.)
    __device__ __forceinline__ void store(uint32_t i, const RtHitRec &h) const
    {
        uint8_t f = 0;
        if (h.leaf != RT_INVALID) f = __float_as_int(((const float4 *)(leafShade + h.leaf))[2].z) > 0 ? 1 : 0;
        flags[i] = f;
    }
};

// The same rays as a type of its own: launches of it share the chip with the bounce launches and two of
// them drain one queue (launch_shadow_overlapped), so their durations in a kernel trace say nothing about
// the walker by itself; they are listed apart from k_traverse<ShadowSrc>'s.
struct ShadowSrcOverlapped : ShadowSrc {};

// Six blocks per compute unit (six waves per SIMD): the walk is a chain of dependent fetches, so what it gains from a
// sixth wave to switch to (the image 99.5 -> 95.5 ms) is more than what it loses by keeping a few kernel constants
// (the spill-area pointer, source pointers) in scratch memory instead of registers: 80 registers per lane instead of
// 89, 28 bytes of scratch, none of it touched inside the node or triangle loops.  Seven (72 registers, 12 stack
// levels in LDS) is far slower: 118.7 ms.
template <bool COUNT, class Src>
__global__ __launch_bounds__(TRV_THREADS, TRV_BLOCKS_PER_CU) void k_traverse(RtSceneView sv, Src src, uint32_t *__restrict__ cursor)
{
    __shared__ uint32_t lds[TRV_LDS_WORDS];
    traverse_rays<COUNT>(sv, src, cursor, lds);
}

template <bool COUNT, class Src>
__global__ __launch_bounds__(PKT_THREADS) void k_traverse_packets(RtSceneView sv, Src src, uint32_t *__restrict__ cursor)
{
    __shared__ uint32_t lds[PKT_LDS_WORDS];
    traverse_packets<COUNT>(sv, src, cursor, lds);
}

// CSMain's extend step with the BVH keyword off (RayTracing.compute:75-77): one ray per lane,
// every lane walking the same primitive lists.
// stats: the context's counter block when counters are on (rays -> [0], hits -> [4]), else null.
template <class Src>
__global__ __launch_bounds__(256) void k_brute_extend(RtBruteView bv, Src src, unsigned long long *__restrict__ stats)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const bool live = i < src.count();
    bool hitSomething = false;
    if (live) {
        rt_f3 o, d;
        src.load(i, o, d);
        const RtHitRec h = brute_force_ray_trace(bv, o, d);
        hitSomething = h.leaf != RT_INVALID;
        src.store(i, h);
    }
    if (stats) {
        const uint32_t rays = (uint32_t)__popcll(__ballot(live)), hits = (uint32_t)__popcll(__ballot(hitSomething));
        if ((threadIdx.x & 63u) == 0 && rays) {
            atomicAdd(&stats[0], (unsigned long long)rays);
            if (hits) atomicAdd(&stats[4], (unsigned long long)hits);
        }
    }
}

__global__ void k_brute_unpack(RtBruteView bv, const float4 *__restrict__ org, const float4 *__restrict__ dir, const RtHitRec *__restrict__ hits,
                               uint32_t n, float *t, uint32_t *prim, float *normal3)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const RtHitRec h = hits[i];
    const bool hitSomething = h.leaf != RT_INVALID;
    t[i] = hitSomething ? h.t : rt_inf();
    prim[i] = h.leaf;
    rt_f3 nrm = rt_splat3(0.0f);
    if (hitSomething) {
        DHit hit;
        const float4 o4 = org[i], d4 = dir[i];
        brute_load_hit(bv, h, rt_mk3(o4.x, o4.y, o4.z), rt_mk3(d4.x, d4.y, d4.z), hit);
        nrm = hit.normal;
    }
    normal3[3 * i] = nrm.x;
    normal3[3 * i + 1] = nrm.y;
    normal3[3 * i + 2] = nrm.z;
}

__global__ void k_unpack_hits(RtSceneView sv, const RtHitRec *__restrict__ hits, uint32_t n, float *t, uint32_t *tri,
                              float *u, float *v)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const RtHitRec h = hits[i];
    if (t) t[i] = h.t;
    if (u) u[i] = h.u;
    if (v) v[i] = h.v;
    if (tri) tri[i] = h.leaf == RT_INVALID ? RT_INVALID : sv.leafTri[h.leaf].tri;
}

__global__ void k_pack_rays(const float *__restrict__ o3, const float *__restrict__ d3, uint32_t n, float4 *o4, float4 *d4)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    o4[i] = make_float4(o3[3 * i], o3[3 * i + 1], o3[3 * i + 2], 0.0f);
    d4[i] = make_float4(d3[3 * i], d3[3 * i + 1], d3[3 * i + 2], 0.0f);
}

// ------------------------------------------------------------------ wavefront kernels
// CSMain prologue (RayTracing.compute:48-65): seed, jitter, camera ray.
// A launch covers `nFrames` consecutive frames of the same view: path slot
// g = frame * nSlots + tile-local pixel slot, frame f uses seeds[f].  USE_DOF
// selects which flag turns the lens camera on (CSMain: the RT_FLAG_DOF_PRIMARY
// extension; ShadowMap: RT_FLAG_DEPTH_OF_FIELD as in the reference).
template <uint32_t DOF_FLAG, bool CLEAR_RT>
__global__ __launch_bounds__(Q_THREADS) void k_generate(RtTileMap tm, RtParams P, const float *__restrict__ seeds, uint32_t nFrames,
                                                         float4 *__restrict__ ray, float4 *__restrict__ pe,
                                                         float4 *__restrict__ img, uint2 *__restrict__ rng, uint32_t *__restrict__ qcount,
                                                         uint32_t *__restrict__ qpos)
{
    __shared__ uint32_t stage[Q_TILE];
    const uint32_t total = tm.nSlots * nFrames;
#pragma unroll 1
    for (int it = 0; it < Q_ITEMS; ++it) {
        // Thread q makes path q: the camera rays are queued frame by frame, a frame's pixel blocks in tile order (a wave
        // of the walker holds one 8x8 block; nSlots is a multiple of 64).  The ShadowMap pass wants another order -- a
        // pixel block's rays of all frames next to each other, k_sm_setup -- and finds a path's ray and hit through
        // qpos[path] = the path's place in this queue (RT_INVALID: a padding slot, never queued).
        const uint32_t q = blockIdx.x * Q_TILE + it * Q_THREADS + threadIdx.x;
        uint32_t frame;
        const uint32_t slot = tm.local_slot(q, &frame);
        const uint32_t g = q < total ? q : RT_INVALID;
        uint32_t px = 0, py = 0;
        const bool active = q < total && tm.slot_to_pixel(slot, &px, &py);
        // padding slots of the image read 0.  A live CSMain path gets its radiance sum and its energy from the first shading
        // pass, which knows both start values (0 and 1) without reading them; a live ShadowMap pixel is written by its pass
        if (q < total && !active) {
            img[g] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            qpos[g] = RT_INVALID;
        }
        stage[it * Q_THREADS + threadIdx.x] = active ? g : RT_INVALID;
    }
    // the ray of an active path is made once its place in the queue is known, and written there
    block_queue_flush(stage, qcount, [&](uint32_t g, uint32_t at) {
        uint32_t frame, px = 0, py = 0;
        tm.slot_to_pixel(tm.local_slot(g, &frame), &px, &py);
        rt_rng r;
        rt_rng_init(&r, px, py, seeds[frame]);
        const float r0 = rt_rand(&r), r1 = rt_rand(&r);
        const float uvx = (((float)px + r0) - 0.5f) / (float)P.width * 2 - 1;
        const float uvy = (((float)py + r1) - 0.5f) / (float)P.height * 2 - 1;
        rt_f3 o, d;
        if (P.flags & DOF_FLAG) d_dof_camera_ray(P, r, uvx, uvy, o, d);
        else d_camera_ray(P, uvx, uvy, o, d);
        ray_record_store(ray, at, o, d, g);
        rng[g] = make_uint2(r.state, __float_as_uint(r.seed));
        qpos[g] = at;
    });
}

// Gathers what CheckTriangle copies into RayHit on an accepted hit
// (BVHRayTracing.cginc:54-73) for the final (closest) triangle only.
__device__ __forceinline__ void load_hit(const RtSceneView &sv, const RtHitRec &h, const rt_f3 o, const rt_f3 d, DHit &hit,
                                         int &castShadow)
{
    hit.distance = h.t;
    hit.position = rt_add3(o, rt_scale3(d, h.t));
    const float4 *p = (const float4 *)(sv.leafShade + h.leaf);
    const float4 a = p[0], b = p[1], c = p[2];
    const rt_f3 n0 = rt_mk3(a.x, a.y, a.z), n1 = rt_mk3(a.w, b.x, b.y), n2 = rt_mk3(b.z, b.w, c.x);
    const rt_f3 nsum = rt_add3(rt_add3(rt_scale3(n0, (1 - h.u) - h.v), rt_scale3(n1, h.u)), rt_scale3(n2, h.v));
    hit.normal = rt_normalize3(nsum);
    const uint32_t mi = __float_as_uint(c.y);
    castShadow = __float_as_int(c.z);
    if (mi < sv.nMaterials) {
        const RtMaterialData &m = sv.materials[mi];
        hit.albedo = ld3(m.albedo);
        hit.metallic = m.metallic;
        hit.smoothness = m.smoothness;
        hit.transparent = m.transparent;
        hit.emissionColor = ld3(m.emissionColor);
        hit.materialType = m.materialType;
        hit.ior = m.ior;
    } else {   // out-of-range StructuredBuffer read = 0
        hit.albedo = rt_splat3(0.0f);
        hit.metallic = hit.smoothness = hit.transparent = hit.ior = 0.0f;
        hit.emissionColor = rt_splat3(0.0f);
        hit.materialType = 0;
    }
}

// One bounce of `col += ray.energy * Shade(hit, ray)` (RayTracing.compute:83-87).
// An emissive hit leaves ray and energy untouched in the reference, so the same
// ray hits the same emitter on every remaining bounce; those identical
// iterations are added here in one go (same float additions, no re-tracing).
// SORT: the block first groups the 2048 paths of its tile by what their shading will do (miss,
// emitter, matte mask, opaque, glass, clear coat) with a counting sort in LDS, so that a wave
// mostly runs one of those branches instead of all of them.  Paths are independent and carry
// their own random stream, so the order they are shaded in changes no value.
// BRUTE: the hit records come from k_brute_extend and name a primitive instead of a BVH leaf.
template <bool SORT, bool BRUTE>
__global__ __launch_bounds__(Q_THREADS) void k_shade(RtSceneView sv, RtBruteView bv, RtTileMap tm, RtParams P, int bounce, float4 *__restrict__ rayIn,
                                                const uint32_t *__restrict__ countIn, float4 *__restrict__ rayOut,
                                                uint32_t *__restrict__ countOut, float4 *__restrict__ pe, float4 *__restrict__ rt,
                                                uint2 *__restrict__ rng, const RtHitRec *__restrict__ hitRec)
{
    __shared__ uint32_t stage[Q_TILE];
    __shared__ uint32_t order[SORT ? Q_TILE : 1];
    __shared__ uint32_t bin[8];
    const uint32_t n = *countIn;
    if (blockIdx.x * Q_TILE >= n) return;
    if (SORT) {
        if (threadIdx.x < 8) bin[threadIdx.x] = 0;
        __syncthreads();
        uint32_t entry[Q_ITEMS], where[Q_ITEMS];
#pragma unroll
        for (int it = 0; it < Q_ITEMS; ++it) {
            const uint32_t i = blockIdx.x * Q_TILE + it * Q_THREADS + threadIdx.x;
            uint32_t key = 7u;                                   // past the end of the queue
            entry[it] = RT_INVALID;
            if (i < n) {
                entry[it] = i;
                const uint32_t leaf = hitRec[i].leaf;
                key = 0u;                                        // miss
                if (leaf != RT_INVALID) {
                    const uint32_t mi = __float_as_uint(((const float4 *)(sv.leafShade + leaf))[2].y);
                    const uint32_t type = mi < sv.nMaterials ? sv.materials[mi].materialType : 0u;
                    key = type == 2u ? 1u : type == 4u ? 2u : type == 1u ? 4u : type == 3u ? 5u : 3u;
                }
            }
            where[it] = (key << 16) | atomicAdd(&bin[key], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t run = 0;
            for (int k = 0; k < 8; ++k) { const uint32_t c = bin[k]; bin[k] = run; run += c; }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < Q_ITEMS; ++it) order[bin[where[it] >> 16] + (where[it] & 0xFFFFu)] = entry[it];
        __syncthreads();
    }
#pragma unroll 1
    for (int it = 0; it < Q_ITEMS; ++it) {
    const uint32_t own = blockIdx.x * Q_TILE + it * Q_THREADS + threadIdx.x;
    bool alive = false;
    const uint32_t i = SORT ? order[it * Q_THREADS + threadIdx.x] : (own < n ? own : RT_INVALID);   // queue entry this thread shades
    if (i != RT_INVALID) {
        const float4 r0 = rayIn[2 * (size_t)i], r1 = rayIn[2 * (size_t)i + 1];
        const uint32_t slot = __float_as_uint(r1.z);
        // the first segment starts with energy 1 and no radiance (RayTracing.compute:67-68): nothing to read
        const float4 e4 = bounce == 0 ? make_float4(1.0f, 1.0f, 1.0f, 0.0f) : pe[slot];
        const float4 c4 = bounce == 0 ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : rt[slot];
        const RtHitRec h = hitRec[i];
        DRay ray;
        ray.origin = rt_mk3(r0.x, r0.y, r0.z);
        ray.direction = rt_mk3(r0.w, r1.x, r1.y);
        ray.energy = rt_mk3(e4.x, e4.y, e4.z);
        rt_f3 col = rt_mk3(c4.x, c4.y, c4.z);
        if (h.leaf == RT_INVALID) {
            const rt_f3 sky = d_shade_miss(sv, P, ray.direction);
            col = rt_add3(col, rt_mul3(ray.energy, sky));
            // energy := 0 -> path ends
        } else {
            DHit hit;
            int castShadow;
            if (BRUTE) brute_load_hit(bv, h, ray.origin, ray.direction, hit);
            else load_hit(sv, h, ray.origin, ray.direction, hit, castShadow);
            if (hit.materialType == 2) {
                const rt_f3 add = rt_mul3(ray.energy, hit.emissionColor);
                for (int k = bounce; k < RT_MAX_BOUNCE; ++k) col = rt_add3(col, add);
            } else if (hit.materialType == 4) {
                ray.origin = rt_sub3(hit.position, rt_scale3(hit.normal, RT_NORMAL_BIAS));
                col = rt_add3(col, rt_mul3(ray.energy, rt_splat3(0.0f)));
                alive = rt_any3(ray.energy);
            } else {
                DShadeCtx c;
                c.sv = &sv;
                c.P = &P;
                const uint2 rs = rng[slot];
                uint32_t px, py;
                uint32_t frameOfSlot;
                tm.slot_to_pixel(tm.local_slot(slot, &frameOfSlot), &px, &py);   // global pixel coordinates feed Rand()
                c.rng.state = rs.x;
                c.rng.seed = __uint_as_float(rs.y);
                c.rng.px = (float)px;
                c.rng.py = (float)py;
                const rt_f3 energyBefore = ray.energy;
                const rt_f3 f = d_function_bsdf(c, hit, ray);
                ray.energy = rt_mul3(ray.energy, f);
                col = rt_add3(col, rt_mul3(energyBefore, rt_splat3(0.0f)));
                rng[slot] = make_uint2(c.rng.state, __float_as_uint(c.rng.seed));
                alive = rt_any3(ray.energy);
            }
        }
        rt[slot] = make_float4(col.x, col.y, col.z, 0.0f);
        alive = alive && (bounce + 1 < RT_MAX_BOUNCE);
        if (alive) {
            ray_record_store(rayIn, i, ray.origin, ray.direction, slot);   // in place; copied to its place in the next queue below
            pe[slot] = make_float4(ray.energy.x, ray.energy.y, ray.energy.z, 0.0f);
        }
    }
    stage[it * Q_THREADS + threadIdx.x] = alive ? i : RT_INVALID;
    }
    block_queue_flush(stage, countOut, [&](uint32_t i, uint32_t at) {
        rayOut[2 * (size_t)at] = rayIn[2 * (size_t)i];
        rayOut[2 * (size_t)at + 1] = rayIn[2 * (size_t)i + 1];
    });
}

// ------------------------------------------------------------------ the later bounces of a few paths in one launch
// A bounce launch lasts as long as its longest ray, and from the third bounce on a frame or two hold so few paths that
// the launches are nothing but that: five of them, 0.2 ms each, with a shading launch in between.  Here a wave takes 64
// paths of the queue of bounce `firstBounce` and brings them to their ends by itself -- walk (the walker, drawing from
// a list of the wave's own live paths in LDS), shade (k_shade's arithmetic for one path, its state kept in registers),
// walk again -- so a path waits for the longest ray among 64 per bounce, not among all, and nothing waits for a launch.
// Paths are independent and carry their own random stream: which wave shades them when changes no value.
struct WaveSrc {             // the live paths of one wave, in LDS
    static constexpr bool FIRST_HIT_MAY_DECIDE = false;
    __device__ __forceinline__ bool first_hit_decides(const rt_f3, const rt_f3, const rt_f3) const { return false; }
    float *rays;               // component c of the path on lane e: rays[c * 64 + e], c = 0..5 (origin, direction)
    uint32_t *hits;            // word c of its hit record: hits[c * 64 + e]
    const uint32_t *list;      // list[j]: the lane of the j-th live path
    uint32_t n;
    __device__ __forceinline__ uint32_t count() const { return n; }
    __device__ __forceinline__ void load(uint32_t j, rt_f3 &o, rt_f3 &d) const
    {
        const uint32_t e = list[j];
        o = rt_mk3(rays[e], rays[64 + e], rays[128 + e]);
        d = rt_mk3(rays[192 + e], rays[256 + e], rays[320 + e]);
    }
    __device__ __forceinline__ void store(uint32_t j, const RtHitRec &h) const
    {
        const uint32_t e = list[j];
        hits[e] = __float_as_uint(h.t);
        hits[64 + e] = __float_as_uint(h.u);
        hits[128 + e] = __float_as_uint(h.v);
        hits[192 + e] = h.leaf;
    }
};
#define CHAIN_WAVE_WORDS (11 * 64)    // 6 ray components, 4 hit words, the list

__global__ __launch_bounds__(TRV_THREADS, 2) void k_traverse_chain(RtSceneView sv, RtTileMap tm, RtParams P, int firstBounce,
                                                                      const float4 *__restrict__ rayIn, const uint32_t *__restrict__ countIn,
                                                                      uint32_t *__restrict__ cursor, const float4 *__restrict__ pe,
                                                                      float4 *__restrict__ rt, const uint2 *__restrict__ rng)
{
    __shared__ uint32_t lds[TRV_LDS_WORDS];
    __shared__ uint32_t own[(TRV_THREADS / 64) * CHAIN_WAVE_WORDS];
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t *mine = own + (threadIdx.x >> 6) * CHAIN_WAVE_WORDS;
    WaveSrc src;
    src.rays = (float *)mine;
    src.hits = mine + 6 * 64;
    src.list = mine + 10 * 64;
    uint32_t *list = mine + 10 * 64;
    const uint32_t n = *countIn;
    for (;;) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(cursor, 64u);         // (fewer paths to a wave, 8 .. 32, change nothing: a chain is as long as its path's rays)
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (base >= n) break;
        const uint32_t i = base + lane;
        bool alive = i < n;
        DRay ray;
        ray.origin = ray.direction = ray.energy = rt_splat3(0.0f);
        rt_f3 col = rt_splat3(0.0f);
        uint32_t slot = 0;
        rt_rng rs;
        rs.state = 0; rs.seed = 0.0f; rs.px = rs.py = 0.0f;
        if (alive) {
            const float4 r0 = rayIn[2 * (size_t)i], r1 = rayIn[2 * (size_t)i + 1];
            slot = __float_as_uint(r1.z);
            const float4 e4 = pe[slot], c4 = rt[slot];
            const uint2 s2 = rng[slot];
            uint32_t px, py, frameOfSlot;
            tm.slot_to_pixel(tm.local_slot(slot, &frameOfSlot), &px, &py);   // global pixel coordinates feed Rand()
            ray.origin = rt_mk3(r0.x, r0.y, r0.z);
            ray.direction = rt_mk3(r0.w, r1.x, r1.y);
            ray.energy = rt_mk3(e4.x, e4.y, e4.z);
            col = rt_mk3(c4.x, c4.y, c4.z);
            rs.state = s2.x;
            rs.seed = __uint_as_float(s2.y);
            rs.px = (float)px;
            rs.py = (float)py;
        }
        const bool mineToWrite = alive;
#pragma unroll 1
        for (int bounce = firstBounce; bounce < RT_MAX_BOUNCE; ++bounce) {
            const unsigned long long live = __ballot(alive);
            if (live == 0ull) break;
            if (alive) {
                list[__popcll(live & ((1ull << lane) - 1ull))] = lane;
                src.rays[lane] = ray.origin.x; src.rays[64 + lane] = ray.origin.y; src.rays[128 + lane] = ray.origin.z;
                src.rays[192 + lane] = ray.direction.x; src.rays[256 + lane] = ray.direction.y; src.rays[320 + lane] = ray.direction.z;
            }
            src.n = (uint32_t)__popcll(live);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            traverse_rays<false>(sv, src, nullptr, lds);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (alive) {
                // (k_shade's arithmetic for one path; the two are held to the same bits by the tests that compare the
                //  per-frame calls with the batch call)
                RtHitRec h;
                h.t = __uint_as_float(src.hits[lane]); h.u = __uint_as_float(src.hits[64 + lane]); h.v = __uint_as_float(src.hits[128 + lane]);
                h.leaf = src.hits[192 + lane];
                alive = false;
                if (h.leaf == RT_INVALID) {
                    const rt_f3 sky = d_shade_miss(sv, P, ray.direction);
                    col = rt_add3(col, rt_mul3(ray.energy, sky));
                } else {
                    DHit hit;
                    int castShadow;
                    load_hit(sv, h, ray.origin, ray.direction, hit, castShadow);
                    if (hit.materialType == 2) {
                        const rt_f3 add = rt_mul3(ray.energy, hit.emissionColor);
                        for (int k = bounce; k < RT_MAX_BOUNCE; ++k) col = rt_add3(col, add);
                    } else if (hit.materialType == 4) {
                        ray.origin = rt_sub3(hit.position, rt_scale3(hit.normal, RT_NORMAL_BIAS));
                        col = rt_add3(col, rt_mul3(ray.energy, rt_splat3(0.0f)));
                        alive = rt_any3(ray.energy);
                    } else {
                        DShadeCtx c;
                        c.sv = &sv;
                        c.P = &P;
                        c.rng = rs;
                        const rt_f3 energyBefore = ray.energy;
                        const rt_f3 f = d_function_bsdf(c, hit, ray);
                        ray.energy = rt_mul3(ray.energy, f);
                        col = rt_add3(col, rt_mul3(energyBefore, rt_splat3(0.0f)));
                        rs = c.rng;
                        alive = rt_any3(ray.energy);
                    }
                }
            }
        }
        if (mineToWrite) rt[slot] = make_float4(col.x, col.y, col.z, 0.0f);
    }
}

// ------------------------------------------------------------------ ShadowMap pass (RayTracing.compute:96-131)
// generate (primary ray, DoF-capable) -> traverse -> setup (hit point, disk
// rotation) -> traverse 16 shadow rays per shaded pixel -> resolve.
__global__ __launch_bounds__(Q_THREADS) void k_sm_setup(RtSceneView sv, RtTileMap tm, RtParams P, const float4 *__restrict__ ray,
                                                   const uint2 *__restrict__ rng,
                                                   const RtHitRec *__restrict__ hitRec, float4 *__restrict__ rec, float4 *__restrict__ recQ,
                                                   float4 *__restrict__ shadowMap, uint32_t *__restrict__ smq,
                                                   uint32_t *__restrict__ smqCount, ShadowFrame frame, int skipUnshadowable,
                                                   unsigned long long *__restrict__ shadowRayCounter, const uint32_t *__restrict__ qpos,
                                                   uint32_t nFrames)
{
    __shared__ uint32_t stage[Q_TILE];
    const uint32_t total = tm.nSlots * nFrames;
#pragma unroll 1
    for (int it = 0; it < Q_ITEMS; ++it) {
    // Thread j takes the path (8x8-pixel block, frame, lane) read off j in that order: the shadow queue holds a pixel
    // block's rays of ALL frames in flight next to each other, and the waves that walk them at about the same time
    // fetch the same nodes and triangles (frame order meets a pixel block again only a whole frame of rays later).
    // The camera and bounce rays are the other way round: queued by pixel block they walk slower than queued by frame
    // (1080p, 64 frames: all eight launches 23.0 against 21.9 ms, the shadow launch 48.7 against 49.2; both queues
    // in their better order: 83.3 ms per image against 84.4 by block and 85.2 by frame; an eighth of the image in 8x8
    // tiles: 12.6 against 14.4 and 13.0).
    const uint32_t j = blockIdx.x * Q_TILE + it * Q_THREADS + threadIdx.x;
    const uint32_t wave = j >> 6;
    const uint32_t path = (wave % nFrames) * tm.nSlots + (wave / nFrames) * 64u + (j & 63u);
    const uint32_t i = j < total ? qpos[path] : RT_INVALID;
    bool shaded = false, skipped = false;
    uint32_t slot = 0;
    if (i != RT_INVALID) {
        const float4 r0 = ray[2 * (size_t)i], r1 = ray[2 * (size_t)i + 1];
        slot = path;
        const RtHitRec h = hitRec[i];
        if (h.leaf == RT_INVALID) {
            shadowMap[slot] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
        } else {
            DHit hit;
            int castShadow;
            load_hit(sv, h, rt_mk3(r0.x, r0.y, r0.z), rt_mk3(r0.w, r1.x, r1.y), hit, castShadow);
            const rt_f3 origin = rt_add3(hit.position, rt_scale3(hit.normal, RT_NORMAL_BIAS));
            const float alpha = hit.materialType == 4 ? 0.0f : 1.0f;
            // a pixel none of whose 16 rays can meet a shadow caster is lit: what k_sm_resolve makes of 16 misses.  The
            // cone test needs the origin only, so it comes before the pixel's random angle is drawn (the draw is local
            // to this kernel: nothing else sees whether it was made)
            bool any = !(skipUnshadowable && frame.cone_misses_the_casters(origin));
            if (any) {
                rt_rng r;
                uint32_t px, py;
                uint32_t frameOfSlot;
                tm.slot_to_pixel(tm.local_slot(slot, &frameOfSlot), &px, &py);
                const uint2 rs = rng[slot];
                r.state = rs.x;
                r.seed = __uint_as_float(rs.y);
                r.px = (float)px;
                r.py = (float)py;
                const float blueNoise = rt_rand(&r);
                const float theta = blueNoise * 2.0f * RT_PI;
                const float cosTheta = rt_cos(theta);
                const float sinTheta = rt_sin(theta);
                if (skipUnshadowable) {
                    any = false;
#pragma unroll 1
                    for (uint32_t k = 0; k < 16u && !any; ++k) any = frame.may_meet_a_caster(origin, frame.direction(k, cosTheta, sinTheta));
                }
                if (any) {
                    rec[slot * 2 + 0] = make_float4(origin.x, origin.y, origin.z, cosTheta);
                    rec[slot * 2 + 1] = make_float4(sinTheta, alpha, 0.0f, 0.0f);
                    shaded = true;
                }
            }
            if (!any) {
                const rt_f3 lit = rt_lerp3(rt_splat3(1.0f), rt_mk3(P.shadowParameter[0], P.shadowParameter[1], P.shadowParameter[2]),
                                           P.shadowParameter[3] * 0.0f);
                shadowMap[slot] = make_float4(lit.x, lit.y, lit.z, alpha);
                skipped = true;
            }
        }
    }
    stage[it * Q_THREADS + threadIdx.x] = shaded ? slot : RT_INVALID;
    if (shadowRayCounter) {     // counters on: the 16 rays of a pixel that got none are rays answered (not walked) all the same
        const unsigned long long m = __ballot(skipped);
        if ((threadIdx.x & 63u) == 0 && m) atomicAdd(shadowRayCounter, 16ull * (unsigned long long)__popcll(m));
    }
    }
    // the record of a shaded pixel goes to its place in the queue (the thread that made it is the one that places it), with
    // the path slot in a spare word: the shadow rays and k_sm_resolve read records and flags in queue order
    block_queue_flush(stage, smqCount, [&](uint32_t shadedSlot, uint32_t at) {
        smq[at] = shadedSlot;
        const float4 b = rec[(size_t)shadedSlot * 2 + 1];
        recQ[(size_t)at * 2] = rec[(size_t)shadedSlot * 2];
        recQ[(size_t)at * 2 + 1] = make_float4(b.x, b.y, __uint_as_float(shadedSlot), 0.0f);
    });
}

// SoftShadow's running mean over the 16 rays, in ray order (Shadow.cginc:109-112).
__global__ __launch_bounds__(256) void k_sm_resolve(RtParams P, const uint32_t *__restrict__ smq, const uint32_t *__restrict__ smqCount,
                                                     const float4 *__restrict__ rec, const uint8_t *__restrict__ flags,
                                                     float4 *__restrict__ shadowMap)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *smqCount) return;
    const uint32_t slot = smq[i];
    const uint4 f4 = *(const uint4 *)(flags + (size_t)i * 16);
    const uint32_t w[4] = {f4.x, f4.y, f4.z, f4.w};
    float shadowTerm = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if ((w[k >> 2] >> ((k & 3) * 8)) & 0xFFu) shadowTerm = rt_lerp(shadowTerm, 1.0f, 1.0f / (float)(k + 1));
    const rt_f3 s = rt_lerp3(rt_splat3(1.0f), rt_mk3(P.shadowParameter[0], P.shadowParameter[1], P.shadowParameter[2]),
                             P.shadowParameter[3] * shadowTerm);
    shadowMap[slot] = make_float4(s.x, s.y, s.z, rec[(size_t)i * 2 + 1].y);
}

// ------------------------------------------------------------------ accumulate
// AddShader pass 0/1 under Blend SrcAlpha OneMinusSrcAlpha: dst = src*a + dst*(1-a),
// a = 1/(n+1); the alpha channel blends src.a (= a) the same way.  The frames of
// a batch are folded in frame order, so the result is the one n successive
// blits produce.
__global__ __launch_bounds__(256) void k_accumulate(uint32_t nSlots, uint32_t nFrames, const float4 *__restrict__ rt,
                                                     const float4 *__restrict__ shadowMap, float4 *__restrict__ conv,
                                                     uint32_t samplePrePixel, int pass)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nSlots) return;
    float4 dst = conv[i];
    for (uint32_t f = 0; f < nFrames; ++f) {
        float4 c = rt[(size_t)f * nSlots + i];
        if (pass == 1) {
            const float4 sm = shadowMap[(size_t)f * nSlots + i];
            c.x *= sm.x; c.y *= sm.y; c.z *= sm.z;
        }
        const float a = 1.0f / ((float)(samplePrePixel + f) + 1.0f);
        dst = make_float4(c.x * a + dst.x * (1.0f - a), c.y * a + dst.y * (1.0f - a), c.z * a + dst.z * (1.0f - a),
                          a * a + dst.w * (1.0f - a));
    }
    conv[i] = dst;
}

__global__ void k_expand_local(RtTileMap tm, const float4 *__restrict__ local, float4 *__restrict__ full)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t px, py;
    if (slot < tm.nSlots && tm.slot_to_pixel(slot, &px, &py)) full[(size_t)py * tm.width + px] = local[slot];
}

// ------------------------------------------------------------------ numeric self-check kernel
// Evaluates the shared numeric rules (include/rt_math.h) on the device so that
// a test can compare them bit for bit with a host build of the same header.
__global__ void k_debug_eval(int which, const float *__restrict__ x, const float *__restrict__ y, uint32_t n, float *__restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (which == 7) {          // Rand() stream of pixel (x[0], x[1]) with seed y[0]
        if (i != 0) return;
        rt_rng r;
        rt_rng_init(&r, (uint32_t)x[0], (uint32_t)x[1], y[0]);
        for (uint32_t k = 0; k < n; ++k) out[k] = rt_rand(&r);
        return;
    }
    if (which == 8) {          // DiscLightImportanceSampling: x = pixel(2) hit.position(3) hit.normal(3) hit.albedo(3) light.position(3) radius
        if (i != 0) return;    // y[0] = seed; out = ray.origin(3) ray.direction(3) func(3) pdf
        DShadeCtx c;
        c.sv = nullptr;
        c.P = nullptr;
        rt_rng_init(&c.rng, (uint32_t)x[0], (uint32_t)x[1], y[0]);
        DHit hit;
        hit.position = rt_mk3(x[2], x[3], x[4]);
        hit.normal = rt_mk3(x[5], x[6], x[7]);
        hit.albedo = rt_mk3(x[8], x[9], x[10]);
        DRay ray;
        ray.origin = ray.direction = rt_splat3(0.0f);
        ray.energy = rt_splat3(1.0f);
        rt_f3 func;
        float pdf;
        d_disc_light(c, hit, ray, rt_mk3(x[11], x[12], x[13]), x[14], func, pdf);
        const float o[10] = {ray.origin.x, ray.origin.y, ray.origin.z, ray.direction.x, ray.direction.y, ray.direction.z, func.x, func.y, func.z, pdf};
        for (uint32_t k = 0; k < n && k < 10; ++k) out[k] = o[k];
        return;
    }
    if (i >= n) return;
    switch (which) {
    case 0: out[i] = rt_sin(x[i]); break;
    case 1: out[i] = rt_cos(x[i]); break;
    case 2: out[i] = rt_acos(x[i]); break;
    case 3: out[i] = rt_pow(x[i], y[i]); break;
    case 4: out[i] = rt_srgb_to_linear1(x[i]); break;
    case 5: out[i] = rt_linear_to_srgb1(x[i]); break;
    default: out[i] = rt_u2f_bits(rt_ftou(x[i])); break;
    }
}

int rt_debug_eval_impl(RtContext *ctx, int which, const float *dx, const float *dy, uint32_t n, float *dout)
{
    hipLaunchKernelGGL(k_debug_eval, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, which, dx, dy, n, dout);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

// ------------------------------------------------------------------ host side
RtSceneView rt_scene_view(RtContext *ctx)
{
    RtSceneView sv;
    memset(&sv, 0, sizeof sv);
    sv.nodes = ctx->dNodes;
    sv.qnodes = ctx->dQNodes;
    sv.wnodes = ctx->optWideNodes ? ctx->dWNodes : nullptr;
    sv.grid = ctx->grid;
    sv.leafTri = ctx->dLeafTri;
    sv.leafShade = ctx->dLeafShade;
    sv.leafRank = ctx->dLeafRank;
    sv.materials = ctx->dMats;
    sv.nMaterials = ctx->nMats;
    sv.nTriangles = ctx->N;
    sv.rootBox = ctx->rootBox;
    sv.sphereLights = ctx->dSphere;
    sv.areaLights = ctx->dArea;
    sv.nSphere = ctx->nSphere;
    sv.nArea = ctx->nArea;
    sv.cube = ctx->dCube;
    sv.cubeSize = ctx->cubeSize;
    sv.stats = ctx->dStats;
    // walker launches that may be on the chip together spill into different areas
    sv.stackSpill = ctx->dStackSpill + (size_t)ctx->spillArea * ctx->numCU * RT_SPILL_WORDS_PER_CU;
    sv.errorFlag = ctx->dErrorFlag;
    // RT_OPT_PRUNE_BY_DISTANCE: 0 no box is left out.  1 (the default) camera, caller and bounce rays leave out what the
    // bounded rule allows (rt_traverse.h: the same hits as with 0, by the bound; it needs the two-level records, which carry
    // the flags of the large triangles, and a scene with small triangles), shadow rays leave out nothing (they stop at a
    // first hit where that settles the answer, which is exact as well).  2 every ray leaves out what begins more than 1.5 %
    // beyond its closest hit: faster, and not the reference's result where Moller-Trumbore's distance is far off (rtcore.h).
    const int opt = ctx->optPruneByDistance;
    sv.pruneScale = opt == 2 ? RT_PRUNE_SCALE : rt_inf();
    sv.boundQ = opt == 1 && ctx->smallEdgeProduct > 0.0f && ctx->optWideNodes ? RT_BOUND_C * ctx->smallEdgeProduct : 0.0f;
    sv.boundE = ctx->smallEdgeSum;
    sv.shareWalks = ctx->optShareLongWalks ? 1u : 0u;
    return sv;
}

// Layout of the counter block (dQCount), zeroed at the start of a batch.  Every counter has a 128-byte line
// to itself: a line that thousands of waves update with atomics answers plain reads very slowly (a ray
// source that read its queue count from the cursors' line on every refill walked its rays 15 times
// slower), and launches that run at the same time would otherwise queue at one line for different cursors.
//   lines 0..8  live-path counts of bounce 0..8     9 shaded pixels of the ShadowMap pass, 10 its primary rays
//   lines 12..19 ray cursors of the 8 extend launches, 20 / 21 ShadowMap primary / shadow rays, 22 rt_trace_rays
#define CNT_LINE 32
#define CNT_Q(b) ((b) * CNT_LINE)
#define CNT_SMQ (9 * CNT_LINE)
#define CNT_SMPRIM (10 * CNT_LINE)
#define CNT_CURSOR(b) ((12 + (b)) * CNT_LINE)
#define CNT_FETCH_SM0 (20 * CNT_LINE)
#define CNT_FETCH_SM1 (21 * CNT_LINE)
#define CNT_FETCH_TRACE (22 * CNT_LINE)
#define CNT_WORDS (24 * CNT_LINE)
static_assert(RT_FRAME_LANES * CNT_WORDS <= RT_COUNTER_BLOCK_WORDS, "counter block too small");   // one per lane
#define RT_MAX_BATCH 64u
#define RT_BATCH_PATH_SLOTS (1u << 27)   // frames in flight are chosen so that a launch covers up to this many paths (224 B each: 30 GB)

void rt_free_render(RtContext *ctx)
{
    void **ptrs[] = {(void **)&ctx->dRt, (void **)&ctx->dConv, (void **)&ctx->dShadowMap,
                     (void **)&ctx->dPe, (void **)&ctx->dRng, (void **)&ctx->dHit,
                     (void **)&ctx->dRay[0], (void **)&ctx->dRay[1], (void **)&ctx->dSmQueue, (void **)&ctx->dSmRec, (void **)&ctx->dSmRecQ,
                     (void **)&ctx->dSmFlags, (void **)&ctx->dSeeds, (void **)&ctx->dQPos};
    for (void **p : ptrs) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    ctx->allocSlots = 0;
    ctx->batchCap = 0;
}

// Path-state buffers for `frames` frames in flight; the accumulator stays one frame.
static int alloc_batch(RtContext *ctx, uint32_t frames)
{
    const uint32_t n = ctx->tile.nSlots;
    if (frames <= ctx->batchCap) return RT_OK;
    // a ShadowMap pass that rt_dispatch started may still be walking its rays on the masked stream, out of the buffers that go now
    int rc = rt_eager_cancel(ctx);
    if (rc) return rc;
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    void **ptrs[] = {(void **)&ctx->dRt, (void **)&ctx->dShadowMap, (void **)&ctx->dPe,
                     (void **)&ctx->dRng, (void **)&ctx->dHit, (void **)&ctx->dRay[0], (void **)&ctx->dRay[1],
                     (void **)&ctx->dSmQueue, (void **)&ctx->dSmRec, (void **)&ctx->dSmRecQ, (void **)&ctx->dSmFlags, (void **)&ctx->dSeeds, (void **)&ctx->dQPos};
    for (void **p : ptrs) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    ctx->batchCap = 0;
    const size_t m = (size_t)(n ? n : 1) * frames;
    RT_HIP(ctx, hipMalloc((void **)&ctx->dRt, m * sizeof(float4)));
    RT_HIP(ctx, hipMalloc((void **)&ctx->dShadowMap, m * sizeof(float4)));
    RT_HIP(ctx, hipMalloc((void **)&ctx->dPe, m * sizeof(float4)));
    RT_HIP(ctx, hipMalloc((void **)&ctx->dRng, m * sizeof(uint2)));
    RT_HIP(ctx, hipMalloc((void **)&ctx->dHit, m * sizeof(RtHitRec)));
    RT_HIP(ctx, hipMalloc((void **)&ctx->dRay[0], m * 2 * sizeof(float4)));
    RT_HIP(ctx, hipMalloc((void **)&ctx->dRay[1], m * 2 * sizeof(float4)));
    RT_HIP(ctx, hipMalloc((void **)&ctx->dSmQueue, m * sizeof(uint32_t)));
    RT_HIP(ctx, hipMalloc((void **)&ctx->dQPos, m * sizeof(uint32_t)));
    RT_HIP(ctx, hipMalloc((void **)&ctx->dSmRec, m * 2 * sizeof(float4)));
    RT_HIP(ctx, hipMalloc((void **)&ctx->dSmRecQ, m * 2 * sizeof(float4)));
    RT_HIP(ctx, hipMalloc((void **)&ctx->dSmFlags, m * 16));
    RT_HIP(ctx, hipMalloc((void **)&ctx->dSeeds, RT_MAX_BATCH * sizeof(float)));
    RT_HIP(ctx, hipMemsetAsync(ctx->dRt, 0, m * sizeof(float4), ctx->stream));
    RT_HIP(ctx, hipMemsetAsync(ctx->dShadowMap, 0, m * sizeof(float4), ctx->stream));
    ctx->batchCap = frames;
    ctx->lastFrame = 0;
    return RT_OK;
}

int rt_alloc_render(RtContext *ctx)
{
    const uint32_t n = ctx->tile.nSlots;
    if (n == ctx->allocSlots && ctx->dConv) return RT_OK;
    rt_free_render(ctx);
    if (ctx->dConv) (void)hipFree(ctx->dConv);
    ctx->dConv = nullptr;
    const size_t m = n ? n : 1;
    RT_HIP(ctx, hipMalloc((void **)&ctx->dConv, m * sizeof(float4)));
    RT_HIP(ctx, hipMemsetAsync(ctx->dConv, 0, m * sizeof(float4), ctx->stream));
    ctx->allocSlots = n;
    ctx->sampleCount = 0;
    return alloc_batch(ctx, RT_FRAME_LANES);     // a frame of path state for each lane of the per-frame call sequence
}

// Frames rt_render_frames keeps in flight for a tile share of nSlots path slots: `option` = RT_OPT_BATCH_FRAMES
// (0: by path count), `wanted` = frames asked for.
uint32_t rt_batch_frames_rule(uint32_t nSlots, uint32_t option, uint32_t wanted)
{
    // As many frames in flight as 128 M paths allow, 64 at most: 64 frames of a whole 1080p image, 16 of a
    // 4K image.  The bounce launches of a frame are latency-bound (each lasts as long as its longest ray,
    // and deep bounces hold few rays), so the more frames share them the better: 8 / 16 / 32 / 64 frames in
    // flight gave 5.1 / 4.4 / 4.0 / 3.8 ms per 1080p frame.
    // An explicit RT_OPT_BATCH_FRAMES is held to the same path count: ray indices (16 shadow rays per path)
    // and ray cursors are 32-bit, so nSlots * frames * 16 must stay below 2^32.
    const uint32_t n = nSlots ? nSlots : 1;
    const uint32_t byPaths = RT_BATCH_PATH_SLOTS / n;
    uint32_t cap = option && option < byPaths ? option : byPaths;
    cap = cap < 1u ? 1u : (cap > RT_MAX_BATCH ? RT_MAX_BATCH : cap);
    return wanted < cap ? (wanted ? wanted : 1u) : cap;
}

uint32_t rt_batch_frames(RtContext *ctx, uint32_t wanted) { return rt_batch_frames_rule(ctx->tile.nSlots, ctx->optBatchFrames, wanted); }

// Persistent launch: enough resident waves to fill the chip; every wave pulls rays from the
// shared cursor until none are left.
template <class Src>
static void launch_traverse(RtContext *ctx, const RtSceneView &sv, const Src &src, uint32_t *cursor, uint32_t maxRays, const char *name)
{
    uint32_t blocks = (uint32_t)ctx->numCU * TRV_BLOCKS_PER_CU;
    const uint32_t need = (maxRays + TRV_THREADS - 1) / TRV_THREADS;
    if (need < blocks) blocks = need ? need : 1;
    rt_timer_begin(ctx, name);
    if (ctx->countersOn) hipLaunchKernelGGL((k_traverse<true, Src>), dim3(blocks), dim3(TRV_THREADS), 0, ctx->stream, sv, src, cursor);
    else hipLaunchKernelGGL((k_traverse<false, Src>), dim3(blocks), dim3(TRV_THREADS), 0, ctx->stream, sv, src, cursor);
    rt_timer_end(ctx);
}

// Same contract, rays walked 16 at a time (coherent sources only: correct for any rays, fast for coherent ones).
template <class Src>
static void launch_traverse_packets(RtContext *ctx, const RtSceneView &sv, const Src &src, uint32_t *cursor, uint32_t maxRays, const char *name)
{
    static int perCU[2] = {0, 0};
    const int ci = ctx->countersOn ? 1 : 0;
    if (perCU[ci] == 0) {
        int n = 0;
        hipError_t e = ctx->countersOn
            ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_traverse_packets<true, Src>, PKT_THREADS, 0)
            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_traverse_packets<false, Src>, PKT_THREADS, 0);
        perCU[ci] = (e == hipSuccess && n > 0) ? n : 4;
    }
    uint32_t blocks = (uint32_t)ctx->numCU * (uint32_t)perCU[ci];
    const uint32_t need = (maxRays + PKT_THREADS - 1) / PKT_THREADS;
    if (need < blocks) blocks = need ? need : 1;
    rt_timer_begin(ctx, name);
    if (ctx->countersOn) hipLaunchKernelGGL((k_traverse_packets<true, Src>), dim3(blocks), dim3(PKT_THREADS), 0, ctx->stream, sv, src, cursor);
    else hipLaunchKernelGGL((k_traverse_packets<false, Src>), dim3(blocks), dim3(PKT_THREADS), 0, ctx->stream, sv, src, cursor);
    rt_timer_end(ctx);
}

static bool shadow_shares_primary(const RtParams &P)
{
    if (P.flags & RT_FLAG_BRUTE_FORCE) return false;   // CSMain's hits are primitives then; the ShadowMap kernel walks the BVH
    // CSMain and ShadowMap draw the same two jitter numbers from the same seed; they trace the
    // same primary ray whenever both use the same camera model (RayTracing.compute:58-65 vs 103-110).
    return ((P.flags & RT_FLAG_DOF_PRIMARY) != 0) == ((P.flags & RT_FLAG_DEPTH_OF_FIELD) != 0);
}

static ShadowFrame shadow_frame(RtContext *ctx);

// (the primary rays and their hits are found through dQPos, which the k_generate launch of the same frames filled)
static void launch_sm_setup(RtContext *ctx, const RtSceneView &sv, uint32_t frames, const float4 *ray)
{
    const uint32_t total = ctx->tile.nSlots * frames;
    rt_timer_begin(ctx, "k_sm_setup");
    hipLaunchKernelGGL(k_sm_setup, dim3((total + Q_TILE - 1) / Q_TILE), dim3(Q_THREADS), 0, ctx->stream, sv, ctx->tile, ctx->params, ray,
                       ctx->dRng, ctx->dHit, ctx->dSmRec, ctx->dSmRecQ, ctx->dShadowMap, ctx->dSmQueue, ctx->dQCount + CNT_SMQ,
                       shadow_frame(ctx), ctx->optCasterBox ? 1 : 0, ctx->countersOn ? ctx->dStats + 37 : nullptr, ctx->dQPos, frames);
    rt_timer_end(ctx);
}

static ShadowFrame shadow_frame(RtContext *ctx)
{
    const RtParams &P = ctx->params;
    ShadowFrame f;
    // Shadow.cginc:92-93, evaluated once on the host with the same IEEE float operations
    f.lightDir = rt_mk3(-P.lightParameter[0], -P.lightParameter[1], -P.lightParameter[2]);
    f.lightTangent = rt_normalize3(rt_cross3(f.lightDir, rt_mk3(0.0f, 1.0f, 0.0f)));
    f.lightBitangent = rt_normalize3(rt_cross3(f.lightTangent, f.lightDir));
    f.lightRadius = P.lightParameter[3];
    for (int a = 0; a < 3; ++a) {
        f.casterMin[a] = ctx->optCasterBox ? ctx->casterBox.min[a] : -rt_inf();
        f.casterMax[a] = ctx->optCasterBox ? ctx->casterBox.max[a] : rt_inf();
    }
    // the cone of the 16 directions: lightDir + (tangent, bitangent) * (disk sample * radius), the samples inside the
    // unit disk, tangent and bitangent unit vectors at right angles to lightDir
    const float len = sqrtf(rt_dot3(f.lightDir, f.lightDir));
    f.coneAxis = rt_scale3(f.lightDir, 1.0f / len);
    f.coneTan = fabsf(f.lightRadius) / len * 1.001f;
    const float ortho = fabsf(rt_dot3(f.lightTangent, f.lightDir)) + fabsf(rt_dot3(f.lightBitangent, f.lightDir)) +
                        fabsf(rt_dot3(f.lightTangent, f.lightBitangent));
    const bool unit = fabsf(rt_dot3(f.lightTangent, f.lightTangent) - 1.0f) < 1e-3f && fabsf(rt_dot3(f.lightBitangent, f.lightBitangent) - 1.0f) < 1e-3f;
    if (!(len > 1e-6f && len < 1e6f && f.coneTan < 10.0f && unit && ortho < 1e-3f * (1.0f + len))) f.coneTan = -1.0f;
    return f;
}

static ShadowSrc shadow_source(RtContext *ctx)
{
    ShadowSrc sh;
    sh.smqCount = ctx->dQCount + CNT_SMQ;
    sh.rec = ctx->dSmRecQ;
    sh.flags = ctx->dSmFlags;
    sh.leafShade = ctx->dLeafShade;
    sh.frame = shadow_frame(ctx);
    sh.ncTri = ctx->dNonCasters;
    sh.nNc = ctx->nNonCasters;
    for (int a = 0; a < 3; ++a) {
        sh.ncMin[a] = ctx->optFirstHitShadow ? ctx->nonCasterBox.min[a] : -rt_inf();
        sh.ncMax[a] = ctx->optFirstHitShadow ? ctx->nonCasterBox.max[a] : rt_inf();
    }
    return sh;
}

static void launch_sm_resolve(RtContext *ctx, uint32_t total)
{
    rt_timer_begin(ctx, "k_sm_resolve");
    hipLaunchKernelGGL(k_sm_resolve, dim3((total + 255) / 256), dim3(256), 0, ctx->stream, ctx->params, ctx->dSmQueue, ctx->dQCount + CNT_SMQ,
                       ctx->dSmRecQ, ctx->dSmFlags, ctx->dShadowMap);
    rt_timer_end(ctx);
}

// 16 shadow rays per shaded pixel + the running mean (needs dSmQueue / dSmRec filled by k_sm_setup).
static void launch_shadow_rays(RtContext *ctx, const RtSceneView &sceneView, uint32_t total)
{
    RtSceneView sv = sceneView;
    sv.stats = ctx->dStats + 32;                 // counted apart from the other traversals
    const ShadowSrc sh = shadow_source(ctx);
    if (ctx->optPacketShadow) launch_traverse_packets(ctx, sv, sh, ctx->dQCount + CNT_FETCH_SM1, total * 16u, "k_shadow_rays");
    else launch_traverse(ctx, sv, sh, ctx->dQCount + CNT_FETCH_SM1, total * 16u, "k_shadow_rays");
    launch_sm_resolve(ctx, total);
}

// ---- shadow rays walked while the later bounces run
// A bounce launch lasts at least as long as its longest ray, a chain of dependent node fetches (0.5 ms on
// the car scene), and from the third bounce on it holds few rays: most of the chip idles through it.  The
// shadow rays depend on the primary hits only.  So, once the second bounce's rays are walked:
//   - a first shadow launch goes to a stream whose queue may use all compute units but a few, with
//     exactly as many blocks as fit there (the walker's blocks stay until the queue is empty);
//   - the remaining bounce and shading launches go to a stream that may use only those few units;
//   - after the last bounce a second shadow launch on the main stream drains the same queue through the
//     same cursor from the units the bounces had.  Which launch walks a ray does not matter to its result.
// The units are split by queue masks, not by leaving slots free: blocks of kernels with different register
// and LDS footprints that come and go among the walker's long-lived blocks fragment a unit's register
// file (a 128-register hole takes one 88-register wave and wastes the rest), which cost the walker a
// fifth of its waves and starved the shading kernel outright when tried.
static uint32_t walker_blocks_per_cu()
{
    static int perCU = 0;
    if (perCU == 0) {
        int n = 0;
        const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_traverse<false, ShadowSrcOverlapped>, TRV_THREADS, 0);
        perCU = (e == hipSuccess && n > 0) ? n : 4;
    }
    return (uint32_t)perCU;
}

// The masked queues of a device -- a set per unit split that was asked for, a pair per frame lane in each -- are shared by
// the contexts of the process that render on it; a set goes when the last context that used it is destroyed.
struct RtMaskedStreams {
    uint32_t users = 0;
    hipStream_t shadow[RT_FRAME_LANES] = {}, chain[RT_FRAME_LANES] = {};   // a pair per frame lane, same masks
};
static std::mutex g_maskedMutex;
static std::map<std::pair<int, uint32_t>, RtMaskedStreams> g_masked;       // by (device, units of the bounce chain)

// paths: what the call that asks has in flight.  Units for the bounce chain: 64 of 256 where the shadow launch is what the
// step waits for (the whole 1080p image with 64 frames in flight: 64 / 96 / 128 units give 86.3 / 87.0 / 87.7 ms; a quarter
// of it: 24.8 / 25.5 / 26.1).  A small batch -- the frames of the per-frame calls above all -- waits for the chain instead,
// whose launches last as long as their longest rays however few there are: it gets 96 (four frames of the per-frame
// calls in flight: 2.42 / 2.16 / 2.27 ms per frame with 64 / 96 / 112; an eighth of the 64-frame batch: no difference).
// A context changes sets only between frames (its callers see to that: no ShadowMap pass is under way), and the work it
// queued on the old set's streams has been ordered into its own streams by then.
static bool overlap_streams_ready(RtContext *ctx, size_t paths)
{
    const uint32_t n = (uint32_t)ctx->numCU;
    const uint32_t few = ctx->optOverlapCUs ? ctx->optOverlapCUs : (paths < ((size_t)24 << 20) && n >= 192u ? 96u : (n >= 128u ? 64u : n / 4u));
    if (few == ctx->overlapCUs) return true;
    if (few < 1 || few >= n || ctx->overlapFailed) return false;
    std::lock_guard<std::mutex> lock(g_maskedMutex);
    RtMaskedStreams &m = g_masked[std::make_pair(ctx->device, few)];
    const bool mine = std::find(ctx->overlapSets.begin(), ctx->overlapSets.end(), few) != ctx->overlapSets.end();
    if (m.users == 0) {
        const uint32_t words = (n + 31u) / 32u;
        std::vector<uint32_t> most(words, 0u), rest(words, 0u);
        for (uint32_t cu = 0; cu < n; ++cu) (cu < n - few ? most : rest)[cu >> 5] |= 1u << (cu & 31u);
        bool ok = true;
        for (int lane = 0; lane < RT_FRAME_LANES && ok; ++lane)
            ok = hipExtStreamCreateWithCUMask(&m.shadow[lane], words, most.data()) == hipSuccess &&
                 hipExtStreamCreateWithCUMask(&m.chain[lane], words, rest.data()) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();             // no masked queues here: the plain order is used
            for (int lane = 0; lane < RT_FRAME_LANES; ++lane) {
                if (m.shadow[lane]) (void)hipStreamDestroy(m.shadow[lane]);
                if (m.chain[lane]) (void)hipStreamDestroy(m.chain[lane]);
                m.shadow[lane] = m.chain[lane] = nullptr;
            }
            ctx->overlapFailed = true;
            return false;
        }
    }
    if (!mine) {
        m.users += 1;
        ctx->overlapSets.push_back(few);
    }
    // (a context may ask from inside the scope of frame lane k >= 1, where its lane-0 fields are swapped with entry k: every
    //  lane's pair is filled whichever way round)
    const uint32_t k = ctx->scopeLane;
    for (uint32_t lane = 0; lane < RT_FRAME_LANES; ++lane) {
        hipStream_t &sh = lane == k ? ctx->streamShadow : (lane == 0 ? ctx->laneShadow[k] : ctx->laneShadow[lane]);
        hipStream_t &ch = lane == k ? ctx->streamChain : (lane == 0 ? ctx->laneChain[k] : ctx->laneChain[lane]);
        sh = m.shadow[lane];
        ch = m.chain[lane];
    }
    ctx->overlapCUs = few;
    return true;
}

void rt_release_overlap_streams(RtContext *ctx)
{
    std::lock_guard<std::mutex> lock(g_maskedMutex);
    for (const uint32_t few : ctx->overlapSets) {
        RtMaskedStreams &m = g_masked[std::make_pair(ctx->device, few)];
        for (int lane = 0; lane < RT_FRAME_LANES; ++lane) {
            (void)hipStreamSynchronize(m.shadow[lane]);
            (void)hipStreamSynchronize(m.chain[lane]);
        }
        if (--m.users == 0) {
            for (int lane = 0; lane < RT_FRAME_LANES; ++lane) {
                (void)hipStreamDestroy(m.shadow[lane]);
                (void)hipStreamDestroy(m.chain[lane]);
            }
            m = RtMaskedStreams();
        }
    }
    ctx->overlapSets.clear();
    ctx->streamShadow = ctx->streamChain = nullptr;
    for (int lane = 0; lane < RT_FRAME_LANES; ++lane) ctx->laneShadow[lane] = ctx->laneChain[lane] = nullptr;
    ctx->overlapCUs = 0;
}

// Called right after the extend launch of the bounce the overlap starts at; from here on ctx->stream is the
// chain stream (overlap_join puts the main stream back).
static int overlap_fork(RtContext *ctx, const RtSceneView &sceneView, uint32_t total)
{
    RtSceneView sv = sceneView;
    sv.stats = ctx->dStats + 32;
    sv.stackSpill = ctx->dStackSpill + (size_t)(ctx->spillArea + 1u) * ctx->numCU * RT_SPILL_WORDS_PER_CU;   // the area next to the one the lane's own stream uses
    ShadowSrcOverlapped sh;
    static_cast<ShadowSrc &>(sh) = shadow_source(ctx);
    RT_HIP(ctx, hipEventRecord(ctx->evFork, ctx->stream));
    RT_HIP(ctx, hipStreamWaitEvent(ctx->streamShadow, ctx->evFork, 0));
    RT_HIP(ctx, hipStreamWaitEvent(ctx->streamChain, ctx->evFork, 0));
    uint32_t blocks = ((uint32_t)ctx->numCU - ctx->overlapCUs) * walker_blocks_per_cu();
    const uint32_t need = (total * 16u + TRV_THREADS - 1) / TRV_THREADS;
    if (need < blocks) blocks = need ? need : 1;
    hipLaunchKernelGGL((k_traverse<false, ShadowSrcOverlapped>), dim3(blocks), dim3(TRV_THREADS), 0, ctx->streamShadow, sv, sh, ctx->dQCount + CNT_FETCH_SM1);
    RT_HIP(ctx, hipEventRecord(ctx->evJoin, ctx->streamShadow));
    ctx->streamMain = ctx->stream;
    ctx->stream = ctx->streamChain;
    return RT_OK;
}

// After the last bounce: back to the main stream, which waits for the chain stream's work.
static int overlap_chain_done(RtContext *ctx)
{
    if (!ctx->streamMain) return RT_OK;
    hipStream_t chain = ctx->stream;
    ctx->stream = ctx->streamMain;
    ctx->streamMain = nullptr;
    RT_HIP(ctx, hipEventRecord(ctx->evFork, chain));
    RT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->evFork, 0));
    return RT_OK;
}

// The main stream helps with the shadow rays from the units the bounces had, then folds the results.
static int overlap_finish(RtContext *ctx, const RtSceneView &sceneView, uint32_t total)
{
    RtSceneView sv = sceneView;
    sv.stats = ctx->dStats + 32;
    ShadowSrcOverlapped sh;
    static_cast<ShadowSrc &>(sh) = shadow_source(ctx);
    uint32_t blocks = ctx->overlapCUs * walker_blocks_per_cu();
    const uint32_t need = (total * 16u + TRV_THREADS - 1) / TRV_THREADS;
    if (need < blocks) blocks = need ? need : 1;
    hipLaunchKernelGGL((k_traverse<false, ShadowSrcOverlapped>), dim3(blocks), dim3(TRV_THREADS), 0, ctx->stream, sv, sh, ctx->dQCount + CNT_FETCH_SM1);
    RT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->evJoin, 0));
    launch_sm_resolve(ctx, total);
    return RT_OK;
}

static int overlap_join(RtContext *ctx, const RtSceneView &sceneView, uint32_t total)
{
    const int rc = overlap_chain_done(ctx);
    return rc ? rc : overlap_finish(ctx, sceneView, total);
}

// CSMain for `frames` frames of the current view (seeds already in dSeeds).
// withShadowSetup: also prepare the ShadowMap pass from CSMain's own primary hits.
// overlapAfter >= 0 (needs withShadowSetup): the shadow rays start after the extend launch of that bounce
// (overlap_fork); the caller then ends with overlap_join instead of launch_shadow_rays.
static int dispatch_frames(RtContext *ctx, uint32_t frames, bool withShadowSetup, int overlapAfter = -1)
{
    const uint32_t n = ctx->tile.nSlots;
    if (n == 0) return RT_OK;
    const uint32_t total = n * frames;
    hipStream_t st = ctx->stream;
    const RtSceneView sv = rt_scene_view(ctx);
    const bool brute = (ctx->params.flags & RT_FLAG_BRUTE_FORCE) != 0;
    const dim3 blk(Q_THREADS), grid((total + Q_TILE - 1) / Q_TILE);
    RT_HIP(ctx, hipMemsetAsync(ctx->dQCount, 0, CNT_WORDS * sizeof(uint32_t), st));
    rt_timer_begin(ctx, "k_generate");
    hipLaunchKernelGGL((k_generate<RT_FLAG_DOF_PRIMARY, true>), grid, blk, 0, st, ctx->tile, ctx->params, ctx->dSeeds, frames,
                       ctx->dRay[0], ctx->dPe, ctx->dRt, ctx->dRng, ctx->dQCount, ctx->dQPos);
    rt_timer_end(ctx);
    // (RT_OPT_CHAIN_BOUNCES) few paths in flight: the bounces from this one on are one launch, k_traverse_chain
    const int chainFrom = (!brute && !ctx->countersOn && ctx->optChainBounces >= 1 && ctx->optChainBounces > overlapAfter && total <= (1u << 23))
                              ? ctx->optChainBounces : -1;
    for (int b = 0; b < RT_MAX_BOUNCE; ++b) {
        float4 *qin = ctx->dRay[b & 1], *qout = ctx->dRay[(b + 1) & 1];
        if (b == chainFrom) {
            uint32_t blocks = (uint32_t)ctx->numCU * 3u;        // (three blocks of its 150 registers a lane fit a unit)
            const uint32_t need = (total + TRV_THREADS - 1) / TRV_THREADS;
            if (need < blocks) blocks = need ? need : 1;
            rt_timer_begin(ctx, "k_chain");
            hipLaunchKernelGGL(k_traverse_chain, dim3(blocks), dim3(TRV_THREADS), 0, st, sv, ctx->tile, ctx->params, b, qin, ctx->dQCount + CNT_Q(b),
                               ctx->dQCount + CNT_CURSOR(b), ctx->dPe, ctx->dRt, ctx->dRng);
            rt_timer_end(ctx);
            break;
        }
        QueueSrc src;
        src.ray = qin;
        src.qcount = ctx->dQCount + CNT_Q(b);
        src.hit = ctx->dHit;
        if (brute) {
            rt_timer_begin(ctx, "k_brute_extend");
            hipLaunchKernelGGL(k_brute_extend<QueueSrc>, dim3((total + 255) / 256), dim3(256), 0, st, ctx->brute, src, ctx->countersOn ? ctx->dStats : nullptr);
            rt_timer_end(ctx);
        } else if (b == 0 && ctx->optPacketPrimary) launch_traverse_packets(ctx, sv, src, ctx->dQCount + CNT_CURSOR(b), total, "k_extend");
        else launch_traverse(ctx, sv, src, ctx->dQCount + CNT_CURSOR(b), total, "k_extend");
        if (b == 0 && withShadowSetup) launch_sm_setup(ctx, sv, frames, qin);   // before shade moves the rays on
        if (b == overlapAfter) {
            const int rc = overlap_fork(ctx, sv, total);
            if (rc) return rc;
            st = ctx->stream;
        }
        rt_timer_begin(ctx, "k_shade");
        if (brute)
            hipLaunchKernelGGL((k_shade<false, true>), grid, blk, 0, st, sv, ctx->brute, ctx->tile, ctx->params, b, qin, ctx->dQCount + CNT_Q(b), qout,
                               ctx->dQCount + CNT_Q(b + 1), ctx->dPe, ctx->dRt, ctx->dRng, ctx->dHit);
        else if (ctx->optMaterialSort)
            hipLaunchKernelGGL((k_shade<true, false>), grid, blk, 0, st, sv, ctx->brute, ctx->tile, ctx->params, b, qin, ctx->dQCount + CNT_Q(b), qout,
                               ctx->dQCount + CNT_Q(b + 1), ctx->dPe, ctx->dRt, ctx->dRng, ctx->dHit);
        else
            hipLaunchKernelGGL((k_shade<false, false>), grid, blk, 0, st, sv, ctx->brute, ctx->tile, ctx->params, b, qin, ctx->dQCount + CNT_Q(b), qout,
                               ctx->dQCount + CNT_Q(b + 1), ctx->dPe, ctx->dRt, ctx->dRng, ctx->dHit);
        rt_timer_end(ctx);
    }
    ctx->lastFrame = frames - 1;
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

// The ShadowMap kernel's own primary rays (used when its camera differs from CSMain's, or stand-alone).
static int shadowmap_frames(RtContext *ctx, uint32_t frames)
{
    const uint32_t n = ctx->tile.nSlots;
    if (n == 0) return RT_OK;
    const uint32_t total = n * frames;
    hipStream_t st = ctx->stream;
    const RtSceneView sv = rt_scene_view(ctx);
    uint32_t *cnt = ctx->dQCount;
    // the path-state buffers of CSMain are free again at this point and are reused
    RT_HIP(ctx, hipMemsetAsync(cnt + CNT_SMQ, 0, 2 * CNT_LINE * sizeof(uint32_t), st));          // CNT_SMQ and CNT_SMPRIM
    RT_HIP(ctx, hipMemsetAsync(cnt + CNT_FETCH_SM0, 0, 2 * CNT_LINE * sizeof(uint32_t), st));    // both ShadowMap cursors
    rt_timer_begin(ctx, "k_sm_generate");
    hipLaunchKernelGGL((k_generate<RT_FLAG_DEPTH_OF_FIELD, false>), dim3((total + Q_TILE - 1) / Q_TILE), dim3(Q_THREADS), 0, st, ctx->tile, ctx->params,
                       ctx->dSeeds, frames, ctx->dRay[0], ctx->dPe, ctx->dShadowMap, ctx->dRng, cnt + CNT_SMPRIM, ctx->dQPos);
    rt_timer_end(ctx);
    QueueSrc prim;
    prim.ray = ctx->dRay[0];
    prim.qcount = cnt + CNT_SMPRIM;
    prim.hit = ctx->dHit;
    if (ctx->optPacketPrimary) launch_traverse_packets(ctx, sv, prim, cnt + CNT_FETCH_SM0, total, "k_extend");
    else launch_traverse(ctx, rt_scene_view(ctx), prim, cnt + CNT_FETCH_SM0, total, "k_extend");
    launch_sm_setup(ctx, sv, frames, ctx->dRay[0]);
    launch_shadow_rays(ctx, sv, total);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

static int upload_seeds(RtContext *ctx, const float *seeds, uint32_t n)
{
    RT_HIP(ctx, hipMemcpyAsync(ctx->dSeeds, seeds, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    return RT_OK;
}

static int overlap_choice(RtContext *ctx, bool shared, uint32_t frames)
{
    // Per-launch counters and event timers describe launches of one stream, one after the other: both keep the plain order.
    if (shared && ctx->optOverlapShadow && ctx->optTwoStreams != 1 && !ctx->countersOn && !ctx->timingOn && !ctx->optPacketShadow &&
        !(ctx->params.flags & RT_FLAG_BRUTE_FORCE) && ctx->tile.nSlots != 0 && ctx->optOverlapAfter + 1 < RT_MAX_BOUNCE &&
        overlap_streams_ready(ctx, (size_t)ctx->tile.nSlots * frames))
        return (int)ctx->optOverlapAfter;
    return -1;
}

// While alive, every per-path buffer of the context points `firstFrame` frames further on, work goes to
// `stream`, the queue counters to `counters` and deep stack levels to spill area `area`: the launch helpers then work
// on a sub-range of the frames in flight without knowing it.  frameLane k >= 1: the scope is lane k of the per-frame
// call sequence, which also has events and masked streams of its own (swapped with lane 0's while the scope lives).
struct RtLaneScope {
    RtContext *ctx;
    hipStream_t stream;
    float4 *rt, *sm, *pe, *smRec, *smRecQ, *ray0, *ray1;
    uint2 *rng;
    RtHitRec *hit;
    uint32_t *smq, *cnt, *qpos;
    uint8_t *smFlags;
    float *seeds;
    uint32_t frameLane;
    RtLaneScope(RtContext *c, hipStream_t s, uint32_t firstFrame, uint32_t *counters, uint32_t area = 1, uint32_t frameLaneIndex = 0)
        : ctx(c), stream(c->stream), rt(c->dRt), sm(c->dShadowMap), pe(c->dPe), smRec(c->dSmRec), smRecQ(c->dSmRecQ), ray0(c->dRay[0]), ray1(c->dRay[1]), rng(c->dRng),
          hit(c->dHit), smq(c->dSmQueue), cnt(c->dQCount), qpos(c->dQPos), smFlags(c->dSmFlags), seeds(c->dSeeds), frameLane(frameLaneIndex)
    {
        const size_t off = (size_t)firstFrame * c->tile.nSlots;
        c->stream = s;
        c->dRt += off; c->dShadowMap += off; c->dPe += off; c->dRng += off; c->dHit += off;
        c->dRay[0] += 2 * off; c->dRay[1] += 2 * off; c->dSmQueue += off; c->dQPos += off; c->dSmRec += 2 * off; c->dSmRecQ += 2 * off; c->dSmFlags += 16 * off;
        c->dSeeds += firstFrame;
        c->dQCount = counters;
        c->spillArea = area;     // (two streams of one batch: area 1 -- the overlapped shadow launch, its other user, is off then)
        if (frameLane) swap_lane_fields();
        c->scopeLane = frameLane;
    }
    void swap_lane_fields()
    {
        RtContext *c = ctx;
        std::swap(c->evFork, c->laneFork[frameLane]);
        std::swap(c->evJoin, c->laneJoin[frameLane]);
        std::swap(c->streamShadow, c->laneShadow[frameLane]);
        std::swap(c->streamChain, c->laneChain[frameLane]);
    }
    ~RtLaneScope()
    {
        RtContext *c = ctx;
        if (frameLane) swap_lane_fields();
        c->scopeLane = 0;
        c->stream = stream;
        c->dRt = rt; c->dShadowMap = sm; c->dPe = pe; c->dSmRec = smRec; c->dSmRecQ = smRecQ; c->dRng = rng; c->dHit = hit;
        c->dRay[0] = ray0; c->dRay[1] = ray1; c->dSmQueue = smq; c->dQPos = qpos; c->dQCount = cnt; c->dSmFlags = smFlags; c->dSeeds = seeds;
        c->spillArea = 0;
    }
};

// ---- several frames of the per-frame call sequence in flight (RT_OPT_PIPELINE_FRAMES)
// A host that keeps the reference's loop calls rt_set_params (new seed), rt_dispatch, rt_dispatch_shadowmap, rt_accumulate
// once per frame (RayTracing.cs:118-164).  One frame by itself is a chain of two dozen launches most of which are as
// long as their longest ray and leave the chip nearly idle; so consecutive frames go round
// RT_FRAME_LANES lanes -- each with a stream, a frame of path state, a counter block, masked streams and spill areas of
// its own -- and the launches of the frames after frame k fill the idle time of frame k's.  What orders them is data: a lane's stream keeps its own frames in order, and the
// fold of a frame into the running mean waits for the fold of the frame before it (evAccumulated), so the mean is
// built in frame order with the same operations -- same bits as one frame after the other.
static bool lanes_active(RtContext *ctx)
{
    return ctx->optPipelineFrames >= 2 && !ctx->countersOn && !ctx->timingOn && ctx->optTwoStreams <= 0 && ctx->stream2 &&
           ctx->batchCap >= (uint32_t)ctx->optPipelineFrames &&
           !(ctx->params.flags & RT_FLAG_BRUTE_FORCE);
}

struct RtFrameLane {
    char room[sizeof(RtLaneScope)];
    RtLaneScope *scope = nullptr;
    RtFrameLane(RtContext *ctx, uint32_t lane)
    {
        if (lane >= 1u) {
            if (!ctx->laneBusy[lane]) {
                // the lane starts up (again): after everything lane 0's stream holds so far -- a batch of rt_render_frames
                // may still be at work in the frame of path state that is this lane's
                (void)hipEventRecord(ctx->evLaneIdle[lane], ctx->stream);
                (void)hipStreamWaitEvent(ctx->laneStream[lane], ctx->evLaneIdle[lane], 0);
            }
            scope = new (room) RtLaneScope(ctx, ctx->laneStream[lane], lane, ctx->dQCount + lane * CNT_WORDS, 2u * lane, lane);
            ctx->laneBusy[lane] = true;
        }
    }
    RtFrameLane(const RtFrameLane &) = delete;
    RtFrameLane &operator=(const RtFrameLane &) = delete;
    ~RtFrameLane()
    {
        if (scope) scope->~RtLaneScope();
    }
};

int rt_join_lanes(RtContext *ctx)
{
    ctx->nextLane = 0;
    for (uint32_t lane = 1; lane < RT_FRAME_LANES; ++lane) {
        if (!ctx->laneBusy[lane]) continue;
        RT_HIP(ctx, hipEventRecord(ctx->evLaneIdle[lane], ctx->laneStream[lane]));
        RT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->evLaneIdle[lane], 0));
        ctx->laneBusy[lane] = false;
    }
    return RT_OK;
}

// A ShadowMap pass that rt_dispatch started and nobody asked for yet is dropped: the main stream waits for
// its launch (the buffers it works on are about to be reused) and the next rt_dispatch_shadowmap starts over.
int rt_eager_cancel(RtContext *ctx)
{
    const int state = ctx->eagerShadow;
    ctx->eagerShadow = RT_EAGER_NONE;
    if (state == RT_EAGER_FORKED) {
        RtFrameLane lane(ctx, ctx->frameLane);
        RT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->evJoin, 0));
    }
    return RT_OK;
}

// CSMain of one frame.  When the parameters say that the ShadowMap kernel follows (RT_FLAG_MATTE_MASK:
// RayTracing.cs:134-135 dispatches it right after CSMain, same uniforms) and it traces the same primary rays,
// its pass is prepared from CSMain's primary hits here and its 16 rays per pixel are walked while the later
// bounces run; rt_dispatch_shadowmap then only drains and folds.  Work counters keep the two kernels apart.
int rt_dispatch_impl(RtContext *ctx)
{
    int rc = rt_eager_cancel(ctx);
    if (rc) return rc;
    const uint32_t laneOfFrame = lanes_active(ctx) ? ctx->nextLane : 0u;
    if (!lanes_active(ctx) && (rc = rt_join_lanes(ctx))) return rc;
    ctx->nextLane = lanes_active(ctx) ? (laneOfFrame + 1u) % (uint32_t)ctx->optPipelineFrames : 0u;
    ctx->frameLane = laneOfFrame;
    {
        RtFrameLane lane(ctx, laneOfFrame);
        if ((rc = upload_seeds(ctx, &ctx->params.seed, 1))) return rc;
        const bool eager = ctx->optEagerShadowMap && (ctx->params.flags & RT_FLAG_MATTE_MASK) && shadow_shares_primary(ctx->params) &&
                           ctx->built && !ctx->countersOn && ctx->tile.nSlots != 0;
        if (!eager) {
            rc = dispatch_frames(ctx, 1, false);
        } else {
            const int overlapAfter = overlap_choice(ctx, true, 1);
            rc = dispatch_frames(ctx, 1, true, overlapAfter);
            const int rc2 = overlap_chain_done(ctx);
            if (!rc) rc = rc2;
            if (!rc) ctx->eagerShadow = overlapAfter >= 0 ? RT_EAGER_FORKED : RT_EAGER_SETUP;
        }
    }
    ctx->lastFrame = laneOfFrame;
    return rc;
}

int rt_shadowmap_impl(RtContext *ctx)
{
    const int state = ctx->eagerShadow;
    ctx->eagerShadow = RT_EAGER_NONE;
    const uint32_t laneOfFrame = ctx->frameLane;
    ctx->lastFrame = laneOfFrame;
    RtFrameLane lane(ctx, laneOfFrame);
    if (state == RT_EAGER_FORKED) return overlap_finish(ctx, rt_scene_view(ctx), ctx->tile.nSlots);
    if (state == RT_EAGER_SETUP) {
        launch_shadow_rays(ctx, rt_scene_view(ctx), ctx->tile.nSlots);
        RT_HIP(ctx, hipGetLastError());
        return RT_OK;
    }
    int rc = upload_seeds(ctx, &ctx->params.seed, 1);
    if (rc) return rc;
    return shadowmap_frames(ctx, 1);
}

static int accumulate_frames(RtContext *ctx, uint32_t frames, uint32_t samplePerPixel, int useShadowMap, size_t frameOffset)
{
    const uint32_t n = ctx->tile.nSlots;
    if (n == 0) return RT_OK;
    rt_timer_begin(ctx, "k_accumulate");
    hipLaunchKernelGGL(k_accumulate, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, frames, ctx->dRt + frameOffset * n,
                       ctx->dShadowMap + frameOffset * n, ctx->dConv, samplePerPixel, useShadowMap ? 1 : 0);
    rt_timer_end(ctx);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

int rt_accumulate_impl(RtContext *ctx, uint32_t samplePerPixel, int useShadowMap)
{
    const uint32_t laneOfFrame = ctx->frameLane;
    const size_t frame = laneOfFrame >= 1u ? 0u : ctx->lastFrame;     // (the scope of a lane >= 1 has moved the frame pointers on already)
    RtFrameLane lane(ctx, laneOfFrame);
    // the running mean is built in frame order: this fold after the fold of the frame before, which another lane made
    const uint32_t before = ctx->lastFoldLane;
    if (before != laneOfFrame && ctx->accumulatedPending[before]) RT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->evAccumulated[before], 0));
    const int rc = accumulate_frames(ctx, 1, samplePerPixel, useShadowMap, frame);
    if (rc) return rc;
    RT_HIP(ctx, hipEventRecord(ctx->evAccumulated[laneOfFrame], ctx->stream));
    ctx->accumulatedPending[laneOfFrame] = true;
    ctx->lastFoldLane = laneOfFrame;
    return RT_OK;
}

static int render_lane(RtContext *ctx, uint32_t frames, bool matte, bool shared)
{
    const int overlapAfter = overlap_choice(ctx, shared, frames);
    int rc = dispatch_frames(ctx, frames, shared, overlapAfter);
    if (rc || !matte) {
        if (ctx->streamMain) {                   // left the bounce loop on the chain stream
            ctx->stream = ctx->streamMain;
            ctx->streamMain = nullptr;
        }
        return rc;
    }
    if (overlapAfter >= 0) return overlap_join(ctx, rt_scene_view(ctx), ctx->tile.nSlots * frames);
    if (shared) launch_shadow_rays(ctx, rt_scene_view(ctx), ctx->tile.nSlots * frames);
    else rc = shadowmap_frames(ctx, frames);
    return rc;
}

// RT_OPT_TWO_STREAMS for a batch of f frames: 0 one stream, 1 two plain streams, 2 .. RT_FRAME_LANES that many frame lanes.
// Left to the library (-1): two lanes for a batch of 8 M to 20 M paths -- a share of the image -- whose launches' ends
// are worth filling with the other half's work (64 frames of a twelfth / an eighth of the 1080p bench image: 9.4 -> 8.7 /
// 12.7 -> 12.1 ms; a sixth and a sixteenth: no difference; a fifth 18.7 -> 19.3, a quarter 22.7 -> 23.4, the whole image
// 83.3 -> 86.5: there the halves only take units from each other; a thirty-second 4.3 -> 4.4).  Three and four lanes
// lose to two (an eighth: 12.8 and 12.6).
static int two_streams_mode(RtContext *ctx, uint32_t f)
{
    int mode = ctx->optTwoStreams;
    if (mode < 0)
        mode = (f >= 2 && (size_t)ctx->tile.nSlots * f <= ((size_t)20 << 20) && (size_t)ctx->tile.nSlots * f > ((size_t)1 << 23) &&
                !ctx->countersOn && !ctx->timingOn && !(ctx->params.flags & RT_FLAG_BRUTE_FORCE)) ? 2 : 0;
    if (mode == 0 || f < 2 || !ctx->stream2) return 0;
    return mode > RT_FRAME_LANES ? RT_FRAME_LANES : mode;
}

// n x CustomRender, `batch` frames in flight at a time.  With RT_OPT_TWO_STREAMS the frames of a batch go
// down two streams (first half / second half) so that one half's launch tails overlap the other half's work.
int rt_render_frames_impl(RtContext *ctx, const float *seeds, uint32_t nFrames)
{
    const bool matte = (ctx->params.flags & RT_FLAG_MATTE_MASK) != 0;
    const bool shared = matte && shadow_shares_primary(ctx->params);
    uint32_t batch = rt_batch_frames(ctx, nFrames);
    int rc = alloc_batch(ctx, batch);
    while (rc == RT_E_NOMEM && batch > 1u) {      // not enough device memory for that many frames in flight: halve
        batch >>= 1;
        rc = alloc_batch(ctx, batch);
    }
    if (rc) return rc;
    if ((rc = rt_eager_cancel(ctx))) return rc;
    for (uint32_t base = 0; base < nFrames; base += batch) {
        const uint32_t f = nFrames - base < batch ? nFrames - base : batch;
        if ((rc = upload_seeds(ctx, seeds + base, f))) return rc;
        // parts: how many streams the batch goes down (1 plain).  Part k >= 1 gets frames of its own and either lane k's
        // resources -- stream, counter block, masked streams, events, spill areas (RT_OPT_TWO_STREAMS 2 .. 4, and the
        // automatic choice) -- or, with value 1, just a second stream
        const int mode = two_streams_mode(ctx, f);
        const uint32_t parts = mode <= 0 ? 1u : (mode == 1 ? 2u : (uint32_t)mode);
        uint32_t first[RT_FRAME_LANES + 1];
        for (uint32_t k = 0; k <= parts; ++k) first[k] = (uint32_t)(((uint64_t)f * k + parts - 1) / parts);   // part 0 is the largest
        for (uint32_t k = 1; k < parts; ++k) {
            const uint32_t fk = first[k + 1] - first[k];
            if (!fk) continue;
            // the part's stream starts after the seeds are up (and, by stream order, after the last batch was folded in)
            RT_HIP(ctx, hipEventRecord(ctx->evLaneIdle[k], ctx->stream));
            RT_HIP(ctx, hipStreamWaitEvent(ctx->laneStream[k], ctx->evLaneIdle[k], 0));
            {
                const bool ownLane = mode >= 2;
                RtLaneScope lane(ctx, ctx->laneStream[k], first[k], ctx->dQCount + k * CNT_WORDS, ownLane ? 2u * k : 1u, ownLane ? k : 0u);
                rc = render_lane(ctx, fk, matte, shared);
            }
            if (rc) return rc;
        }
        if ((rc = render_lane(ctx, first[1], matte, shared))) return rc;
        for (uint32_t k = 1; k < parts; ++k) {
            if (first[k + 1] == first[k]) continue;
            RT_HIP(ctx, hipEventRecord(ctx->evLaneIdle[k], ctx->laneStream[k]));
            RT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->evLaneIdle[k], 0));
        }
        ctx->lastFrame = f - 1;
        if ((rc = accumulate_frames(ctx, f, ctx->sampleCount, matte ? 1 : 0, 0))) return rc;
        ctx->sampleCount += f;
        ctx->params.seed = seeds[base + f - 1];
    }
    return RT_OK;
}

// rt_readback_async: on the stream of the frame that was dispatched last (its lane), behind that frame's fold, the image
// is expanded into the staging frame; the copy stream takes it from there.  The next frame's fold -- the other lane's --
// waits for this lane's evAccumulated, which is moved behind the expansion: the running mean is not folded further
// before it has been staged.
int rt_readback_async_impl(RtContext *ctx, const float4 *src, float *pinnedDst)
{
    const uint32_t laneOfFrame = ctx->frameLane;
    const size_t px = (size_t)ctx->params.width * ctx->params.height;
    RtFrameLane lane(ctx, laneOfFrame);
    const uint32_t slot = ctx->readbackSlot;
    ctx->readbackSlot ^= 1u;
    float4 *staging = ctx->dReadbackFull + (size_t)slot * (px ? px : 1);
    if (ctx->copyPending[slot]) RT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->evCopied[slot], 0));      // this staging frame is still being read
    if (ctx->tile.count > 1) RT_HIP(ctx, hipMemsetAsync(staging, 0, px * sizeof(float4), ctx->stream));   // other ranks' pixels read 0
    const int rc = rt_expand_local_impl(ctx, src, staging);
    if (rc) return rc;
    RT_HIP(ctx, hipEventRecord(ctx->evExpanded, ctx->stream));
    if (ctx->accumulatedPending[laneOfFrame]) RT_HIP(ctx, hipEventRecord(ctx->evAccumulated[laneOfFrame], ctx->stream));
    RT_HIP(ctx, hipStreamWaitEvent(ctx->streamCopy, ctx->evExpanded, 0));
    RT_HIP(ctx, hipMemcpyAsync(pinnedDst, staging, px * sizeof(float4), hipMemcpyDeviceToHost, ctx->streamCopy));
    RT_HIP(ctx, hipEventRecord(ctx->evCopied[slot], ctx->streamCopy));
    ctx->copyPending[slot] = true;
    return RT_OK;
}

int rt_trace_device_impl(RtContext *ctx, const float4 *o, const float4 *d, uint32_t n, RtHitRec *hits)
{
    if (n == 0) return RT_OK;
    const RtSceneView sv = rt_scene_view(ctx);
    RT_HIP(ctx, hipMemsetAsync(ctx->dQCount + CNT_FETCH_TRACE, 0, sizeof(uint32_t), ctx->stream));
    ArraySrc src;
    src.org = o;
    src.dir = d;
    src.n = n;
    src.hits = hits;
    if (ctx->optPacketTrace) launch_traverse_packets(ctx, sv, src, ctx->dQCount + CNT_FETCH_TRACE, n, "k_trace");
    else launch_traverse(ctx, sv, src, ctx->dQCount + CNT_FETCH_TRACE, n, "k_trace");
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

// ---- first-hit queries that allocate nothing
// The walker's scratch -- the ray cursor, the overflow flag and the per-lane spill levels of the stack -- in a
// file-scope array instead of buffers the context allocates, for callers that may not allocate device memory
// (tight serving loops, harnesses that forbid hipMalloc).  Every GPU the library runs on holds one copy of the
// array until the process ends; all calls on a GPU share it, so calls that overlap on one GPU race.
//   words 0..31   the ray cursor's line          (cleared by every call)
//   words 32..63  the overflow flag's line       (cleared by every call)
//   from word 64  RT_STATIC_BLOCKS x TRV_THREADS lanes x (RT_STACK_DEPTH - TRV_STACK_LDS) stack levels (written before read)
#define RT_STATIC_BLOCKS 1280u                                     // five blocks on each of 256 compute units, one fewer than are resident at a time: the array stays at 20 MB
#define RT_STATIC_HEAD_WORDS 64u
#define RT_STATIC_SPILL_WORDS (RT_STATIC_BLOCKS * TRV_THREADS * (RT_STACK_DEPTH - TRV_STACK_LDS))
static __device__ __align__(256) uint32_t scratch[RT_STATIC_HEAD_WORDS + RT_STATIC_SPILL_WORDS];   // 20 MB

// In host code `scratch` is only a handle for the runtime; the memory of each GPU's copy is found once per
// device, with that device current, and only that pointer is used from then on.
#define RT_STATIC_MAX_DEVICES 64
static std::mutex g_scratchMutex;
static uint32_t *g_scratchPtr[RT_STATIC_MAX_DEVICES];

int rt_static_scratch_init(RtContext *ctx)
{
    if (ctx->device < 0 || ctx->device >= RT_STATIC_MAX_DEVICES) return RT_OK;    // such a device gets no allocation-free entry point
    std::lock_guard<std::mutex> lock(g_scratchMutex);
    if (g_scratchPtr[ctx->device]) return RT_OK;
    uint32_t *ptr = nullptr;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) return rt_fail(ctx, RT_E_DEVICE, "hipGetSymbolAddress gave no address for the static scratch array");
    g_scratchPtr[ctx->device] = ptr;
    return RT_OK;
}

int rt_trace_device_static_impl(RtContext *ctx, const float4 *o, const float4 *d, uint32_t n, RtHitRec *hits)
{
    uint32_t *ptr = nullptr;
    if (ctx->device >= 0 && ctx->device < RT_STATIC_MAX_DEVICES) {
        std::lock_guard<std::mutex> lock(g_scratchMutex);
        ptr = g_scratchPtr[ctx->device];
    }
    if (!ptr) return rt_fail(ctx, RT_E_STATE, "no static scratch array on device %d", ctx->device);
    if (n == 0) return RT_OK;
    RtSceneView sv = rt_scene_view(ctx);
    sv.stats = nullptr;
    sv.errorFlag = ptr + 32;
    sv.stackSpill = ptr + RT_STATIC_HEAD_WORDS;
    RT_HIP(ctx, hipMemsetAsync(ptr, 0, RT_STATIC_HEAD_WORDS * sizeof(uint32_t), ctx->stream));
    ArraySrc src;
    src.org = o;
    src.dir = d;
    src.n = n;
    src.hits = hits;
    uint32_t blocks = (n + TRV_THREADS - 1) / TRV_THREADS;
    if (blocks > RT_STATIC_BLOCKS) blocks = RT_STATIC_BLOCKS;      // the spill area holds that many blocks' lanes
    hipLaunchKernelGGL((k_traverse<false, ArraySrc>), dim3(blocks), dim3(TRV_THREADS), 0, ctx->stream, sv, src, ptr);
    RT_HIP(ctx, hipGetLastError());
    uint32_t flag = 0;
    RT_HIP(ctx, hipMemcpyAsync(&flag, ptr + 32, sizeof flag, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (flag) return rt_fail(ctx, RT_E_OVERFLOW, "traversal stack overflow (depth > %d)", RT_STACK_DEPTH);
    return RT_OK;
}

int rt_trace_primitives_impl(RtContext *ctx, const float4 *o, const float4 *d, uint32_t n, float *t, uint32_t *prim, float *normal3)
{
    RtHitRec *hits = nullptr;
    RtArena arena(ctx->stream);
    RT_HIP(ctx, arena.alloc(&hits, (size_t)n));
    ArraySrc src;
    src.org = o;
    src.dir = d;
    src.n = n;
    src.hits = hits;
    rt_timer_begin(ctx, "k_brute_extend");
    hipLaunchKernelGGL(k_brute_extend<ArraySrc>, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->brute, src, ctx->countersOn ? ctx->dStats : nullptr);
    rt_timer_end(ctx);
    hipLaunchKernelGGL(k_brute_unpack, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->brute, o, d, hits, n, t, prim, normal3);
    RT_HIP(ctx, hipGetLastError());
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_expand_local_on(RtContext *ctx, hipStream_t stream, const float4 *local, float4 *full)
{
    const uint32_t n = ctx->tile.nSlots;
    if (n == 0) return RT_OK;
    hipLaunchKernelGGL(k_expand_local, dim3((n + 255) / 256), dim3(256), 0, stream, ctx->tile, local, full);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

int rt_expand_local_impl(RtContext *ctx, const float4 *local, float4 *full) { return rt_expand_local_on(ctx, ctx->stream, local, full); }

int rt_hits_to_host_impl(RtContext *ctx, const RtHitRec *hits, uint32_t n, float *t, uint32_t *tri, float *u, float *v)
{
    // t/tri/u/v are DEVICE scratch arrays here
    const RtSceneView sv = rt_scene_view(ctx);
    hipLaunchKernelGGL(k_unpack_hits, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, sv, hits, n, t, tri, u, v);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

int rt_pack_rays_impl(RtContext *ctx, const float *o3, const float *d3, uint32_t n, float4 *o4, float4 *d4)
{
    hipLaunchKernelGGL(k_pack_rays, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, o3, d3, n, o4, d4);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}
