// rt_brute.h -- BruteForceRayTrace (BruteForceRayTracing.cginc:68-279): every ray
// against every plane, sphere, cube and custom-mesh triangle, in list order.
//
// All lanes of a wave walk the same lists, so every primitive fetch is wave-uniform
// (scalar loads through the constant cache); the per-ray state is the 16-byte hit
// record.  Custom-mesh triangles are brought to world space once per upload
// (k_brute_prepare: the same mul(localToWorldMatrix, float4(v, 1)) and the same edge
// subtractions the reference repeats for every ray) and stored in test order.
//
// What the reference's loop does is kept as is: planes and triangles replace the hit
// only when closer (and in front); spheres and cubes overwrite it unconditionally
// (IntersectSphere / IntersectCube have no distance test), a cube behind the origin
// yields a negative distance, and only IntersectPlane fills materialType / ior /
// clearCoatColor / shadow flags -- the others leave them as they were.  In this path
// materialType is therefore always 0 and ior / transparent / emissionColor are never
// read by Shade, so the hit record keeps just (distance, u, v, last writer).
#ifndef RT_BRUTE_H_
#define RT_BRUTE_H_

#include "rt_internal.h"
#include "rt_shade.h"

// IntersectPlane (BruteForceRayTracing.cginc:68-99)
__device__ __forceinline__ void brute_plane(const RtPlane &pl, uint32_t index, const rt_f3 o, const rt_f3 d, RtHitRec &h)
{
    const rt_f3 n = ld3(pl.normal), pos = ld3(pl.position), size = ld3(pl.size);
    const float t = rt_dot3(n, rt_sub3(pos, o)) / rt_dot3(n, d);
    if (t > 0.0f && t < h.t) {
        const rt_f3 half = rt_divs3(size, 2.0f);
        const rt_f3 mn = rt_sub3(pos, half), mx = rt_add3(pos, half);
        const rt_f3 p = rt_add3(o, rt_scale3(d, t));
        if (p.x > mn.x && p.x < mx.x && p.y > mn.y && p.y < mx.y && p.z > mn.z && p.z < mx.z) {
            h.t = t;
            h.leaf = (RT_PRIM_PLANE << RT_PRIM_SHIFT) | index;
        }
    }
}

// IntersectSphere (:101-130)
__device__ __forceinline__ void brute_sphere(const RtSphere &sp, uint32_t index, const rt_f3 o, const rt_f3 d, RtHitRec &h)
{
    const rt_f3 dir = rt_sub3(ld3(sp.center), o);
    const float distance = rt_length3(dir);
    if (distance < sp.radius) return;
    const float l = rt_dot3(dir, rt_normalize3(d));
    if (l < 0.0f) return;
    const float m = rt_sqrt(distance * distance - l * l);
    if (m > sp.radius) return;
    const float q = rt_sqrt(sp.radius * sp.radius - m * m);
    h.t = l - q;
    h.leaf = (RT_PRIM_SPHERE << RT_PRIM_SHIFT) | index;
}

// IntersectCube (:132-163); the face the ray enters through goes into h.u (0, 1, 2 = x, y, z)
__device__ __forceinline__ void brute_cube(const RtCube &cu, uint32_t index, const rt_f3 o, const rt_f3 d, RtHitRec &h)
{
    const rt_f3 invDir = rt_mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const rt_f3 tMin = rt_mul3(rt_sub3(ld3(cu.min), o), invDir);
    const rt_f3 tMax = rt_mul3(rt_sub3(ld3(cu.max), o), invDir);
    const rt_f3 t1 = rt_mk3(rt_min(tMin.x, tMax.x), rt_min(tMin.y, tMax.y), rt_min(tMin.z, tMax.z));
    const rt_f3 t2 = rt_mk3(rt_max(tMin.x, tMax.x), rt_max(tMin.y, tMax.y), rt_max(tMin.z, tMax.z));
    const float tNear = rt_max(rt_max(t1.x, t1.y), t1.z);
    const float tFar = rt_min(rt_min(t2.x, t2.y), t2.z);
    if (tNear > tFar) return;
    h.t = tNear;
    h.u = tNear == t1.y ? 1.0f : (tNear == t1.z ? 2.0f : 0.0f);
    h.leaf = (RT_PRIM_CUBE << RT_PRIM_SHIFT) | index;
}

// BruteForceRayTrace (:246-279)
__device__ __forceinline__ RtHitRec brute_force_ray_trace(const RtBruteView &bv, const rt_f3 o, const rt_f3 d)
{
    RtHitRec h;
    h.t = rt_inf();
    h.u = h.v = 0.0f;
    h.leaf = RT_INVALID;
    for (uint32_t i = 0; i < bv.nPlanes; ++i) brute_plane(bv.planes[i], i, o, d, h);
    for (uint32_t i = 0; i < bv.nSpheres; ++i) brute_sphere(bv.spheres[i], i, o, d, h);
    for (uint32_t i = 0; i < bv.nCubes; ++i) brute_cube(bv.cubes[i], i, o, d, h);
    // IntersectCustomMesh (:206-243) over the triangles of every mesh, in list order
    for (uint32_t i = 0; i < bv.nTris; ++i) {
        const float4 a = bv.triPos[3 * i], b = bv.triPos[3 * i + 1], c = bv.triPos[3 * i + 2];
        tri_test_values(rt_mk3(a.x, a.y, a.z), rt_mk3(b.x, b.y, b.z), rt_mk3(c.x, c.y, c.z), (RT_PRIM_TRIANGLE << RT_PRIM_SHIFT) | i, o, d, h);
    }
    if (!(h.t < rt_inf())) h.leaf = RT_INVALID;   // what Shade's `hit.distance < 1.#INF` sees (a NaN distance is a miss)
    return h;
}

__device__ __forceinline__ void brute_surface(const float *albedo, float metallic, float smoothness, float transparent, const float *emission, DHit &hit)
{
    hit.albedo = ld3(albedo);
    hit.metallic = metallic;
    hit.smoothness = smoothness;
    hit.transparent = transparent;
    hit.emissionColor = ld3(emission);
    hit.materialType = 0;
    hit.ior = 0.0f;
}

// The RayHit fields Shade reads, for the primitive that wrote the hit last.
__device__ __forceinline__ void brute_load_hit(const RtBruteView &bv, const RtHitRec &h, const rt_f3 o, const rt_f3 d, DHit &hit)
{
    const uint32_t kind = h.leaf >> RT_PRIM_SHIFT, i = h.leaf & RT_PRIM_INDEX_MASK;
    hit.distance = h.t;
    hit.position = rt_add3(o, rt_scale3(d, h.t));
    if (kind == RT_PRIM_PLANE) {
        const RtPlane &pl = bv.planes[i];
        hit.normal = ld3(pl.normal);
        brute_surface(pl.albedo, pl.metallic, pl.smoothness, pl.transparent, pl.emissionColor, hit);
    } else if (kind == RT_PRIM_SPHERE) {
        const RtSphere &sp = bv.spheres[i];
        hit.normal = rt_normalize3(rt_sub3(hit.position, ld3(sp.center)));
        brute_surface(sp.albedo, sp.metallic, sp.smoothness, sp.transparent, sp.emissionColor, hit);
    } else if (kind == RT_PRIM_CUBE) {
        const RtCube &cu = bv.cubes[i];
        const rt_f3 axis = h.u == 1.0f ? rt_mk3(0.0f, 1.0f, 0.0f) : (h.u == 2.0f ? rt_mk3(0.0f, 0.0f, 1.0f) : rt_mk3(1.0f, 0.0f, 0.0f));
        const rt_f3 centre = rt_scale3(rt_add3(ld3(cu.min), ld3(cu.max)), 0.5f);
        const rt_f3 side = rt_sub3(o, centre);
        hit.normal = rt_mul3(axis, rt_mk3(rt_sign(side.x), rt_sign(side.y), rt_sign(side.z)));
        brute_surface(cu.albedo, cu.metallic, cu.smoothness, cu.transparent, cu.emissionColor, hit);
    } else {
        const float4 a = bv.triNrm[3 * i], b = bv.triNrm[3 * i + 1], c = bv.triNrm[3 * i + 2];
        const rt_f3 n0 = rt_mk3(a.x, a.y, a.z), n1 = rt_mk3(b.x, b.y, b.z), n2 = rt_mk3(c.x, c.y, c.z);
        hit.normal = rt_normalize3(rt_add3(rt_add3(rt_scale3(n0, (1 - h.u) - h.v), rt_scale3(n1, h.u)), rt_scale3(n2, h.v)));
        const RtCustomMesh &m = bv.meshes[bv.triMesh[i]];
        brute_surface(m.albedo, m.metallic, m.smoothness, m.transparent, m.emissionColor, hit);
    }
}

#endif
