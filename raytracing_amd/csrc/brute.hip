// brute.hip -- scene upload of the brute-force primitive path (MeshCollector.Init,
// MeshCollector.cs:170-200) and the first-hit query rt_trace_primitives.
#include <vector>

#include "rt_internal.h"
#include "rt_shade.h"

// World-space copy of one custom-mesh triangle: what IntersectCustomMesh computes for
// every ray (BruteForceRayTracing.cginc:214-220) and the two edge subtractions of
// IntersectTriangle_MT97 (:170-171), done once.  Output: triPos = (v0, e1, e2),
// triNrm = (n0, n1, n2), three float4 per triangle.
__global__ void k_brute_prepare(const RtCustomMesh *__restrict__ meshes, const uint32_t *__restrict__ triMesh, const uint32_t *__restrict__ triFirstIndex,
                                const float *__restrict__ vertices, const float *__restrict__ normals, const int32_t *__restrict__ indices,
                                uint32_t nTris, float4 *__restrict__ triPos, float4 *__restrict__ triNrm)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nTris) return;
    const RtCustomMesh &m = meshes[triMesh[t]];
    const uint32_t first = triFirstIndex[t];
    rt_f3 v[3], n[3];
    for (int k = 0; k < 3; ++k) {
        const int32_t vi = indices[first + k];
        float w[4];
        d_mul44(m.localToWorldMatrix, vertices[3 * vi], vertices[3 * vi + 1], vertices[3 * vi + 2], 1.0f, w);
        v[k] = rt_mk3(w[0], w[1], w[2]);
        d_mul44(m.localToWorldMatrix, normals[3 * vi], normals[3 * vi + 1], normals[3 * vi + 2], 0.0f, w);
        n[k] = rt_mk3(w[0], w[1], w[2]);
    }
    const rt_f3 e1 = rt_sub3(v[1], v[0]), e2 = rt_sub3(v[2], v[0]);
    triPos[3 * t + 0] = make_float4(v[0].x, v[0].y, v[0].z, 0.0f);
    triPos[3 * t + 1] = make_float4(e1.x, e1.y, e1.z, 0.0f);
    triPos[3 * t + 2] = make_float4(e2.x, e2.y, e2.z, 0.0f);
    for (int k = 0; k < 3; ++k) triNrm[3 * t + k] = make_float4(n[k].x, n[k].y, n[k].z, 0.0f);
}

void rt_free_brute(RtContext *ctx)
{
    void **ptrs[] = {(void **)&ctx->dPlanes, (void **)&ctx->dSpheres, (void **)&ctx->dCubes, (void **)&ctx->dMeshes,
                     (void **)&ctx->dTriPos, (void **)&ctx->dTriNrm, (void **)&ctx->dTriMesh};
    for (void **p : ptrs) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    memset(&ctx->brute, 0, sizeof ctx->brute);
    ctx->bruteSet = false;
}

template <typename T>
static int to_device(RtContext *ctx, T **dst, const T *src, size_t count)
{
    RT_HIP(ctx, hipMalloc((void **)dst, (count ? count : 1) * sizeof(T)));
    if (count) RT_HIP(ctx, hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
    return RT_OK;
}

extern "C" int rt_primitives_upload(RtContext *ctx, const RtPlane *planes, uint32_t n_planes, const RtSphere *spheres, uint32_t n_spheres,
                                    const RtCube *cubes, uint32_t n_cubes, const RtCustomMesh *meshes, uint32_t n_meshes,
                                    const float *vertices3, const float *normals3, uint32_t n_vertices, const int32_t *indices,
                                    uint32_t n_indices)
{
    if (!ctx) return RT_E_INVALID;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if ((n_planes && !planes) || (n_spheres && !spheres) || (n_cubes && !cubes) || (n_meshes && !meshes))
        return rt_fail(ctx, RT_E_INVALID, "null primitive array");
    if ((n_vertices && (!vertices3 || !normals3)) || (n_indices && !indices)) return rt_fail(ctx, RT_E_INVALID, "null mesh buffers");
    // triangles in the order IntersectCustomMesh meets them: mesh by mesh, index triple by index triple
    std::vector<uint32_t> triMesh, triFirst;
    for (uint32_t m = 0; m < n_meshes; ++m) {
        const int64_t off = meshes[m].indicesOffset, cnt = meshes[m].indicesCount;
        if (off < 0 || cnt < 0 || cnt % 3 != 0 || off + cnt > (int64_t)n_indices)
            return rt_fail(ctx, RT_E_INVALID, "custom mesh %u: indices [%lld, %lld) do not fit the index buffer of %u (count must be a multiple of 3)",
                           m, (long long)off, (long long)(off + cnt), n_indices);
        for (int64_t i = off; i < off + cnt; i += 3) {
            for (int k = 0; k < 3; ++k)
                if (indices[i + k] < 0 || (uint32_t)indices[i + k] >= n_vertices)
                    return rt_fail(ctx, RT_E_INVALID, "custom mesh %u: index %d outside the %u vertices", m, indices[i + k], n_vertices);
            triMesh.push_back(m);
            triFirst.push_back((uint32_t)i);
        }
    }
    if (triMesh.size() > RT_PRIM_INDEX_MASK || n_planes > RT_PRIM_INDEX_MASK || n_spheres > RT_PRIM_INDEX_MASK || n_cubes > RT_PRIM_INDEX_MASK)
        return rt_fail(ctx, RT_E_INVALID, "too many primitives");
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    rt_free_brute(ctx);
    const uint32_t nTris = (uint32_t)triMesh.size();
    int rc;
    if ((rc = to_device(ctx, &ctx->dPlanes, planes, n_planes))) return rc;
    if ((rc = to_device(ctx, &ctx->dSpheres, spheres, n_spheres))) return rc;
    if ((rc = to_device(ctx, &ctx->dCubes, cubes, n_cubes))) return rc;
    if ((rc = to_device(ctx, &ctx->dMeshes, meshes, n_meshes))) return rc;
    if ((rc = to_device(ctx, &ctx->dTriMesh, triMesh.data(), nTris))) return rc;
    RT_HIP(ctx, hipMalloc((void **)&ctx->dTriPos, (size_t)(nTris ? nTris : 1) * 3 * sizeof(float4)));
    RT_HIP(ctx, hipMalloc((void **)&ctx->dTriNrm, (size_t)(nTris ? nTris : 1) * 3 * sizeof(float4)));
    if (nTris) {
        uint32_t *dFirst = nullptr;
        float *dV = nullptr, *dN = nullptr;
        int32_t *dI = nullptr;
        if ((rc = to_device(ctx, &dFirst, triFirst.data(), nTris))) return rc;
        if ((rc = to_device(ctx, &dV, vertices3, (size_t)3 * n_vertices))) return rc;
        if ((rc = to_device(ctx, &dN, normals3, (size_t)3 * n_vertices))) return rc;
        if ((rc = to_device(ctx, &dI, indices, n_indices))) return rc;
        rt_timer_begin(ctx, "k_brute_prepare");
        hipLaunchKernelGGL(k_brute_prepare, dim3((nTris + 255) / 256), dim3(256), 0, ctx->stream, ctx->dMeshes, ctx->dTriMesh, dFirst, dV, dN, dI, nTris,
                           ctx->dTriPos, ctx->dTriNrm);
        rt_timer_end(ctx);
        RT_HIP(ctx, hipGetLastError());
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(dFirst);
        (void)hipFree(dV);
        (void)hipFree(dN);
        (void)hipFree(dI);
    }
    RtBruteView &bv = ctx->brute;
    bv.planes = ctx->dPlanes;
    bv.spheres = ctx->dSpheres;
    bv.cubes = ctx->dCubes;
    bv.meshes = ctx->dMeshes;
    bv.triPos = ctx->dTriPos;
    bv.triNrm = ctx->dTriNrm;
    bv.triMesh = ctx->dTriMesh;
    bv.nPlanes = n_planes;
    bv.nSpheres = n_spheres;
    bv.nCubes = n_cubes;
    bv.nMeshes = n_meshes;
    bv.nTris = nTris;
    ctx->bruteSet = true;
    ctx->sampleCount = 0;
    return RT_OK;
}

extern "C" int rt_trace_primitives(RtContext *ctx, const float *origins, const float *directions, uint32_t n, float *out_t, uint32_t *out_primitive,
                                   float *out_normal)
{
    if (!ctx) return RT_E_INVALID;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->bruteSet) return rt_fail(ctx, RT_E_STATE, "rt_primitives_upload has not been called");
    if (n == 0) return RT_OK;
    if (!origins || !directions) return rt_fail(ctx, RT_E_INVALID, "null rays");
    float *d3 = nullptr, *dout = nullptr;   // dout: [t | primitive | normal(3)] x n
    float4 *d4 = nullptr;
    RtArena arena(ctx->stream);
    RT_HIP(ctx, arena.alloc(&d3, (size_t)n * 6));
    RT_HIP(ctx, arena.alloc(&d4, (size_t)n * 2));
    RT_HIP(ctx, arena.alloc(&dout, (size_t)n * 5));
    RT_HIP(ctx, hipMemcpyAsync(d3, origins, (size_t)n * 12, hipMemcpyHostToDevice, ctx->stream));
    RT_HIP(ctx, hipMemcpyAsync(d3 + (size_t)3 * n, directions, (size_t)n * 12, hipMemcpyHostToDevice, ctx->stream));
    int rc = rt_pack_rays_impl(ctx, d3, d3 + (size_t)3 * n, n, d4, d4 + n);
    if (!rc) rc = rt_trace_primitives_impl(ctx, d4, d4 + n, n, dout, (uint32_t *)(dout + n), dout + (size_t)2 * n);
    if (rc) return rc;
    if (out_t) RT_HIP(ctx, hipMemcpyAsync(out_t, dout, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (out_primitive) RT_HIP(ctx, hipMemcpyAsync(out_primitive, dout + n, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (out_normal) RT_HIP(ctx, hipMemcpyAsync(out_normal, dout + (size_t)2 * n, (size_t)n * 12, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    rt_timer_resolve(ctx);
    return RT_OK;
}
