// render.hip -- BVH traversal and the wavefront (queue-based) path tracer.
//
// Reference path being replaced (same results on the same inputs and seeds):
//   Assets/Shaders/BVH/BVHRayTracing.cginc:27-133        traversal (rt_traverse)
//   Assets/Shaders/RayTracing/BruteForceRayTracing.cginc:166-204  Moller-Trumbore
//   Assets/Shaders/RayTracing/RayTracing.compute:44-91   CSMain  -> k_generate / k_extend / k_shade
//   Assets/Shaders/RayTracing/RayTracing.compute:96-131  ShadowMap -> k_shadowmap
//   Assets/Shaders/RayTracing/AddShader.shader:47-79     accumulate -> k_accumulate
//
// The reference runs one thread per pixel through all 8 bounces.  Here a frame
// is a sequence of launches over compacted queues of live paths:
//   generate -> { extend (traverse) -> shade (+ wave64 ballot compaction) } x 8
// Path state lives in SoA arrays indexed by tile-local slot; queues hold slots.
#include "rt_internal.h"
#include "rt_shade.h"

#define TRV_THREADS 256

// ------------------------------------------------------------------ traversal
// Slab test exactly as RayBoxIntersection: true divisions, min/max returning
// the non-NaN operand (v_min_f32 / v_max_f32 in IEEE mode).
__device__ __forceinline__ bool box_hit(float bminx, float bminy, float bminz, float bmaxx, float bmaxy, float bmaxz,
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

// Depth-first walk in the reference's visiting order: at a node, the left
// then the right LEAF child are tested at once; then the right subtree is
// walked before the left one (the reference pushes left, then right, and pops
// right first).  No distance pruning: the closest hit wins with strict '<', so
// on equal distances the first-visited triangle is kept.
// stk = this lane's column of the LDS stack, entries `stride` words apart.
template <bool COUNT>
__device__ __forceinline__ RtHitRec rt_traverse(const RtSceneView &sv, const rt_f3 o, const rt_f3 d, uint32_t *stk,
                                                const uint32_t stride, uint32_t cnt[3])
{
    RtHitRec h;
    h.t = rt_inf();
    h.u = 0.0f;
    h.v = 0.0f;
    h.leaf = RT_INVALID;
    if (sv.nTriangles == 0) return h;
    if (COUNT) cnt[0]++;
    const RtAABB &rb = sv.rootBox;
    if (sv.nTriangles < 2 || !box_hit(rb.min[0], rb.min[1], rb.min[2], rb.max[0], rb.max[1], rb.max[2], o, d)) return h;
    uint32_t node = 0;
    uint32_t sp = 0;
    for (;;) {
        const float4 *np = (const float4 *)(sv.nodes + node);
        const float4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
        const uint32_t l = __float_as_uint(q3.x), r = __float_as_uint(q3.y);
        const bool lLeaf = (l & RT_LEAF_BIT) != 0, rLeaf = (r & RT_LEAF_BIT) != 0;
        const bool hitL = box_hit(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, o, d);
        const bool hitR = box_hit(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, o, d);
        if (COUNT) {
            cnt[0] += (lLeaf ? 0u : 1u) + (rLeaf ? 0u : 1u);
            cnt[1] += (lLeaf ? 1u : 0u) + (rLeaf ? 1u : 0u);
            cnt[2] += ((lLeaf && hitL) ? 1u : 0u) + ((rLeaf && hitR) ? 1u : 0u);
        }
        if (lLeaf && hitL) tri_test(sv, l & ~RT_LEAF_BIT, o, d, h);
        if (rLeaf && hitR) tri_test(sv, r & ~RT_LEAF_BIT, o, d, h);
        const bool goL = !lLeaf && hitL, goR = !rLeaf && hitR;
        if (goR) {
            if (goL) {
                if (sp >= RT_STACK_DEPTH) { *sv.errorFlag = 1u; break; }
                stk[sp * stride] = l;
                sp++;
            }
            node = r;
        } else if (goL) {
            node = l;
        } else {
            if (sp == 0) break;
            sp--;
            node = stk[sp * stride];
        }
    }
    return h;
}

__device__ __forceinline__ void flush_counters(const RtSceneView &sv, uint32_t rays, const uint32_t cnt[3], uint32_t hits)
{
    // wave-level reduction, then one atomic per wave per counter
    uint32_t v[5] = {rays, cnt[0], cnt[1], cnt[2], hits};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        uint32_t x = v[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if ((threadIdx.x & 63u) == 0 && x) atomicAdd(&sv.stats[k], (unsigned long long)x);
    }
}

// wave64 ballot + prefix-count compaction: appends `slot` of every lane with
// keep == true to queue[], one atomic per wave.
__device__ __forceinline__ void queue_push(bool keep, uint32_t slot, uint32_t *__restrict__ queue, uint32_t *__restrict__ count)
{
    const unsigned long long mask = __ballot(keep);
    if (mask == 0ull) return;
    const uint32_t lane = threadIdx.x & 63u;
    const int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(count, (uint32_t)__popcll(mask));
    base = __shfl(base, leader, 64);
    if (keep) queue[base + __popcll(mask & ((1ull << lane) - 1ull))] = slot;
}

// ------------------------------------------------------------------ first-hit query kernel
template <bool COUNT>
__global__ __launch_bounds__(TRV_THREADS) void k_trace(RtSceneView sv, const float4 *__restrict__ org,
                                                        const float4 *__restrict__ dir, uint32_t n, RtHitRec *__restrict__ hits)
{
    __shared__ uint32_t stack[RT_STACK_DEPTH * TRV_THREADS];
    const uint32_t i = blockIdx.x * TRV_THREADS + threadIdx.x;
    uint32_t cnt[3] = {0, 0, 0};
    uint32_t rays = 0, nh = 0;
    if (i < n) {
        const float4 o4 = org[i], d4 = dir[i];
        const RtHitRec h = rt_traverse<COUNT>(sv, rt_mk3(o4.x, o4.y, o4.z), rt_mk3(d4.x, d4.y, d4.z), stack + threadIdx.x,
                                              TRV_THREADS, cnt);
        hits[i] = h;
        rays = 1;
        nh = h.leaf != RT_INVALID;
    }
    if (COUNT) flush_counters(sv, rays, cnt, nh);
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
__global__ __launch_bounds__(256) void k_generate(RtTileMap tm, RtParams P, float4 *__restrict__ po, float4 *__restrict__ pd,
                                                   float4 *__restrict__ pe, float4 *__restrict__ rt, uint2 *__restrict__ rng,
                                                   uint32_t *__restrict__ queue, uint32_t *__restrict__ qcount)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t px = 0, py = 0;
    const bool active = slot < tm.nSlots && tm.slot_to_pixel(slot, &px, &py);
    if (slot < tm.nSlots) rt[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (active) {
        rt_rng r;
        rt_rng_init(&r, px, py, P.seed);
        const float r0 = rt_rand(&r), r1 = rt_rand(&r);
        const float uvx = (((float)px + r0) - 0.5f) / (float)P.width * 2 - 1;
        const float uvy = (((float)py + r1) - 0.5f) / (float)P.height * 2 - 1;
        rt_f3 o, d;
        if (P.flags & RT_FLAG_DOF_PRIMARY) d_dof_camera_ray(P, r, uvx, uvy, o, d);
        else d_camera_ray(P, uvx, uvy, o, d);
        po[slot] = make_float4(o.x, o.y, o.z, 0.0f);
        pd[slot] = make_float4(d.x, d.y, d.z, 0.0f);
        pe[slot] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
        rng[slot] = make_uint2(r.state, __float_as_uint(r.seed));
    }
    queue_push(active, slot, queue, qcount);
}

// BVHRayTrace for every live path of this bounce.
template <bool COUNT>
__global__ __launch_bounds__(TRV_THREADS) void k_extend(RtSceneView sv, const uint32_t *__restrict__ queue,
                                                         const uint32_t *__restrict__ qcount, const float4 *__restrict__ po,
                                                         const float4 *__restrict__ pd, RtHitRec *__restrict__ hit)
{
    __shared__ uint32_t stack[RT_STACK_DEPTH * TRV_THREADS];
    const uint32_t i = blockIdx.x * TRV_THREADS + threadIdx.x;
    const uint32_t n = *qcount;
    if (blockIdx.x * TRV_THREADS >= n) return;
    uint32_t cnt[3] = {0, 0, 0};
    uint32_t rays = 0, nh = 0;
    if (i < n) {
        const uint32_t slot = queue[i];
        const float4 o4 = po[slot], d4 = pd[slot];
        const RtHitRec h = rt_traverse<COUNT>(sv, rt_mk3(o4.x, o4.y, o4.z), rt_mk3(d4.x, d4.y, d4.z), stack + threadIdx.x,
                                              TRV_THREADS, cnt);
        hit[slot] = h;
        rays = 1;
        nh = h.leaf != RT_INVALID;
    }
    if (COUNT) flush_counters(sv, rays, cnt, nh);
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
__global__ __launch_bounds__(256) void k_shade(RtSceneView sv, RtTileMap tm, RtParams P, int bounce, const uint32_t *__restrict__ queueIn,
                                                const uint32_t *__restrict__ countIn, uint32_t *__restrict__ queueOut,
                                                uint32_t *__restrict__ countOut, float4 *__restrict__ po,
                                                float4 *__restrict__ pd, float4 *__restrict__ pe, float4 *__restrict__ rt,
                                                uint2 *__restrict__ rng, const RtHitRec *__restrict__ hitRec)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = *countIn;
    if (blockIdx.x * blockDim.x >= n) return;
    bool alive = false;
    uint32_t slot = 0;
    if (i < n) {
        slot = queueIn[i];
        const float4 o4 = po[slot], d4 = pd[slot], e4 = pe[slot];
        float4 c4 = rt[slot];
        const RtHitRec h = hitRec[slot];
        DRay ray;
        ray.origin = rt_mk3(o4.x, o4.y, o4.z);
        ray.direction = rt_mk3(d4.x, d4.y, d4.z);
        ray.energy = rt_mk3(e4.x, e4.y, e4.z);
        rt_f3 col = rt_mk3(c4.x, c4.y, c4.z);
        if (h.leaf == RT_INVALID) {
            const rt_f3 sky = d_shade_miss(sv, P, ray.direction);
            col = rt_add3(col, rt_mul3(ray.energy, sky));
            // energy := 0 -> path ends
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
                const uint2 rs = rng[slot];
                uint32_t px, py;
                tm.slot_to_pixel(slot, &px, &py);   // global pixel coordinates feed Rand()
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
            po[slot] = make_float4(ray.origin.x, ray.origin.y, ray.origin.z, 0.0f);
            pd[slot] = make_float4(ray.direction.x, ray.direction.y, ray.direction.z, 0.0f);
            pe[slot] = make_float4(ray.energy.x, ray.energy.y, ray.energy.z, 0.0f);
        }
    }
    queue_push(alive, slot, queueOut, countOut);
}

// ------------------------------------------------------------------ ShadowMap kernel
__constant__ float c_blue_noise[16][2] = {   // first 16 entries of BLUE_NOISE_IN_DISK (Shadow.cginc:7-72)
    {0.478712f, 0.875764f},  {-0.337956f, -0.793959f}, {-0.955259f, -0.028164f}, {0.864527f, 0.325689f},
    {0.209342f, -0.395657f}, {-0.106779f, 0.672585f},  {0.156213f, 0.235113f},   {-0.413644f, -0.082856f},
    {-0.415667f, 0.323909f}, {0.141896f, -0.939980f},  {0.954932f, -0.182516f},  {-0.766184f, 0.410799f},
    {-0.434912f, -0.458845f}, {0.415242f, -0.078724f}, {0.728335f, -0.491777f},  {-0.058086f, -0.066401f}};

template <bool COUNT>
__global__ __launch_bounds__(TRV_THREADS) void k_shadowmap(RtSceneView sv, RtTileMap tm, RtParams P, float4 *__restrict__ out)
{
    __shared__ uint32_t stack[RT_STACK_DEPTH * TRV_THREADS];
    const uint32_t slot = blockIdx.x * TRV_THREADS + threadIdx.x;
    uint32_t px = 0, py = 0;
    const bool active = slot < tm.nSlots && tm.slot_to_pixel(slot, &px, &py);
    uint32_t cnt[3] = {0, 0, 0};
    uint32_t rays = 0, nh = 0;
    if (active) {
        rt_rng r;
        rt_rng_init(&r, px, py, P.seed);
        const float r0 = rt_rand(&r), r1 = rt_rand(&r);
        const float uvx = (((float)px + r0) - 0.5f) / (float)P.width * 2 - 1;
        const float uvy = (((float)py + r1) - 0.5f) / (float)P.height * 2 - 1;
        rt_f3 o, d;
        if (P.flags & RT_FLAG_DEPTH_OF_FIELD) d_dof_camera_ray(P, r, uvx, uvy, o, d);
        else d_camera_ray(P, uvx, uvy, o, d);
        const RtHitRec h = rt_traverse<COUNT>(sv, o, d, stack + threadIdx.x, TRV_THREADS, cnt);
        rays++;
        float4 res = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
        if (h.leaf != RT_INVALID) {
            nh++;
            DHit hit;
            int castShadow;
            load_hit(sv, h, o, d, hit, castShadow);
            // SoftShadow (Shadow.cginc:84-113)
            const rt_f3 origin = rt_add3(hit.position, rt_scale3(hit.normal, RT_NORMAL_BIAS));
            const rt_f3 lightDir = rt_mk3(-P.lightParameter[0], -P.lightParameter[1], -P.lightParameter[2]);
            const float lightRadius = P.lightParameter[3];
            const float blueNoise = rt_rand(&r);
            const float theta = blueNoise * 2.0f * RT_PI;
            const float cosTheta = rt_cos(theta);
            const float sinTheta = rt_sin(theta);
            float shadowTerm = 0.0f;
            const rt_f3 lightTangent = rt_normalize3(rt_cross3(lightDir, rt_mk3(0.0f, 1.0f, 0.0f)));
            const rt_f3 lightBitangent = rt_normalize3(rt_cross3(lightTangent, lightDir));
            for (int k = 0; k < 16; ++k) {
                const float sx = c_blue_noise[k][0], sy = c_blue_noise[k][1];
                float dx = sx * cosTheta - sy * sinTheta;
                float dy = sx * sinTheta + sy * cosTheta;
                dx *= lightRadius;
                dy *= lightRadius;
                const rt_f3 sd = rt_normalize3(rt_add3(rt_add3(lightDir, rt_scale3(lightTangent, dx)), rt_scale3(lightBitangent, dy)));
                const RtHitRec sh = rt_traverse<COUNT>(sv, origin, sd, stack + threadIdx.x, TRV_THREADS, cnt);
                rays++;
                if (sh.leaf != RT_INVALID) {
                    nh++;
                    const int cast = __float_as_int(((const float4 *)(sv.leafShade + sh.leaf))[2].z);
                    if (cast > 0) shadowTerm = rt_lerp(shadowTerm, 1.0f, 1.0f / (float)(k + 1));
                }
            }
            const rt_f3 s = rt_lerp3(rt_splat3(1.0f), rt_mk3(P.shadowParameter[0], P.shadowParameter[1], P.shadowParameter[2]),
                                     P.shadowParameter[3] * shadowTerm);
            res = make_float4(s.x, s.y, s.z, hit.materialType == 4 ? 0.0f : 1.0f);
        }
        out[slot] = res;
    } else if (slot < tm.nSlots) {
        out[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    if (COUNT) flush_counters(sv, rays, cnt, nh);
}

// ------------------------------------------------------------------ accumulate
// AddShader pass 0/1 under Blend SrcAlpha OneMinusSrcAlpha: dst = src*a + dst*(1-a),
// a = 1/(n+1); the alpha channel blends src.a (= a) the same way.
__global__ __launch_bounds__(256) void k_accumulate(uint32_t n, const float4 *__restrict__ rt, const float4 *__restrict__ shadowMap,
                                                     float4 *__restrict__ conv, float samplePrePixel, int pass)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 c = rt[i];
    if (pass == 1) {
        const float4 s = shadowMap[i];
        c.x *= s.x; c.y *= s.y; c.z *= s.z;
    }
    const float a = 1.0f / (samplePrePixel + 1.0f);
    const float4 dst = conv[i];
    conv[i] = make_float4(c.x * a + dst.x * (1.0f - a), c.y * a + dst.y * (1.0f - a), c.z * a + dst.z * (1.0f - a),
                          a * a + dst.w * (1.0f - a));
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
    sv.leafTri = ctx->dLeafTri;
    sv.leafShade = ctx->dLeafShade;
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
    sv.errorFlag = ctx->dErrorFlag;
    return sv;
}

void rt_free_render(RtContext *ctx)
{
    void **ptrs[] = {(void **)&ctx->dRt, (void **)&ctx->dConv, (void **)&ctx->dShadowMap, (void **)&ctx->dPo,
                     (void **)&ctx->dPd, (void **)&ctx->dPe, (void **)&ctx->dRng, (void **)&ctx->dHit,
                     (void **)&ctx->dQueue[0], (void **)&ctx->dQueue[1], (void **)&ctx->dQCount};
    for (void **p : ptrs) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    ctx->allocSlots = 0;
}

int rt_alloc_render(RtContext *ctx)
{
    const uint32_t n = ctx->tile.nSlots;
    if (n == ctx->allocSlots && ctx->dRt) return RT_OK;
    rt_free_render(ctx);
    const size_t m = n ? n : 1;
    RT_HIP(ctx, hipMalloc((void **)&ctx->dRt, m * sizeof(float4)));
    RT_HIP(ctx, hipMalloc((void **)&ctx->dConv, m * sizeof(float4)));
    RT_HIP(ctx, hipMalloc((void **)&ctx->dShadowMap, m * sizeof(float4)));
    RT_HIP(ctx, hipMalloc((void **)&ctx->dPo, m * sizeof(float4)));
    RT_HIP(ctx, hipMalloc((void **)&ctx->dPd, m * sizeof(float4)));
    RT_HIP(ctx, hipMalloc((void **)&ctx->dPe, m * sizeof(float4)));
    RT_HIP(ctx, hipMalloc((void **)&ctx->dRng, m * sizeof(uint2)));
    RT_HIP(ctx, hipMalloc((void **)&ctx->dHit, m * sizeof(RtHitRec)));
    RT_HIP(ctx, hipMalloc((void **)&ctx->dQueue[0], m * sizeof(uint32_t)));
    RT_HIP(ctx, hipMalloc((void **)&ctx->dQueue[1], m * sizeof(uint32_t)));
    RT_HIP(ctx, hipMalloc((void **)&ctx->dQCount, (RT_MAX_BOUNCE + 2) * sizeof(uint32_t)));
    RT_HIP(ctx, hipMemsetAsync(ctx->dRt, 0, m * sizeof(float4), ctx->stream));
    RT_HIP(ctx, hipMemsetAsync(ctx->dConv, 0, m * sizeof(float4), ctx->stream));
    RT_HIP(ctx, hipMemsetAsync(ctx->dShadowMap, 0, m * sizeof(float4), ctx->stream));
    ctx->allocSlots = n;
    ctx->sampleCount = 0;
    return RT_OK;
}

int rt_dispatch_impl(RtContext *ctx)
{
    const uint32_t n = ctx->tile.nSlots;
    if (n == 0) return RT_OK;
    hipStream_t st = ctx->stream;
    const RtSceneView sv = rt_scene_view(ctx);
    const dim3 blk(256), grid((n + 255) / 256);
    RT_HIP(ctx, hipMemsetAsync(ctx->dQCount, 0, (RT_MAX_BOUNCE + 2) * sizeof(uint32_t), st));
    rt_timer_begin(ctx, "k_generate");
    hipLaunchKernelGGL(k_generate, grid, blk, 0, st, ctx->tile, ctx->params, ctx->dPo, ctx->dPd, ctx->dPe, ctx->dRt,
                       ctx->dRng, ctx->dQueue[0], ctx->dQCount);
    rt_timer_end(ctx);
    for (int b = 0; b < RT_MAX_BOUNCE; ++b) {
        uint32_t *qin = ctx->dQueue[b & 1], *qout = ctx->dQueue[(b + 1) & 1];
        rt_timer_begin(ctx, "k_extend");
        if (ctx->countersOn)
            hipLaunchKernelGGL((k_extend<true>), grid, dim3(TRV_THREADS), 0, st, sv, qin, ctx->dQCount + b, ctx->dPo, ctx->dPd, ctx->dHit);
        else
            hipLaunchKernelGGL((k_extend<false>), grid, dim3(TRV_THREADS), 0, st, sv, qin, ctx->dQCount + b, ctx->dPo, ctx->dPd, ctx->dHit);
        rt_timer_end(ctx);
        rt_timer_begin(ctx, "k_shade");
        hipLaunchKernelGGL(k_shade, grid, blk, 0, st, sv, ctx->tile, ctx->params, b, qin, ctx->dQCount + b, qout, ctx->dQCount + b + 1,
                           ctx->dPo, ctx->dPd, ctx->dPe, ctx->dRt, ctx->dRng, ctx->dHit);
        rt_timer_end(ctx);
    }
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

int rt_shadowmap_impl(RtContext *ctx)
{
    const uint32_t n = ctx->tile.nSlots;
    if (n == 0) return RT_OK;
    const RtSceneView sv = rt_scene_view(ctx);
    const dim3 grid((n + TRV_THREADS - 1) / TRV_THREADS);
    rt_timer_begin(ctx, "k_shadowmap");
    if (ctx->countersOn)
        hipLaunchKernelGGL((k_shadowmap<true>), grid, dim3(TRV_THREADS), 0, ctx->stream, sv, ctx->tile, ctx->params, ctx->dShadowMap);
    else
        hipLaunchKernelGGL((k_shadowmap<false>), grid, dim3(TRV_THREADS), 0, ctx->stream, sv, ctx->tile, ctx->params, ctx->dShadowMap);
    rt_timer_end(ctx);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

int rt_accumulate_impl(RtContext *ctx, uint32_t samplePerPixel, int useShadowMap)
{
    const uint32_t n = ctx->tile.nSlots;
    if (n == 0) return RT_OK;
    rt_timer_begin(ctx, "k_accumulate");
    hipLaunchKernelGGL(k_accumulate, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, ctx->dRt, ctx->dShadowMap, ctx->dConv,
                       (float)samplePerPixel, useShadowMap ? 1 : 0);
    rt_timer_end(ctx);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

int rt_trace_device_impl(RtContext *ctx, const float4 *o, const float4 *d, uint32_t n, RtHitRec *hits)
{
    if (n == 0) return RT_OK;
    const RtSceneView sv = rt_scene_view(ctx);
    const dim3 grid((n + TRV_THREADS - 1) / TRV_THREADS);
    rt_timer_begin(ctx, "k_trace");
    if (ctx->countersOn) hipLaunchKernelGGL((k_trace<true>), grid, dim3(TRV_THREADS), 0, ctx->stream, sv, o, d, n, hits);
    else hipLaunchKernelGGL((k_trace<false>), grid, dim3(TRV_THREADS), 0, ctx->stream, sv, o, d, n, hits);
    rt_timer_end(ctx);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

int rt_expand_local_impl(RtContext *ctx, const float4 *local, float4 *full)
{
    const uint32_t n = ctx->tile.nSlots;
    if (n == 0) return RT_OK;
    hipLaunchKernelGGL(k_expand_local, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->tile, local, full);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

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
