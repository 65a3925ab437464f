// api.hip -- the extern "C" surface of librtcore.so (include/rtcore.h).
#include <vector>
#include <algorithm>
#include <thread>
#include <stdarg.h>
#include <stdio.h>

#include <new>

#include "rt_internal.h"

int rt_fail(RtContext *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

// ---- per-kernel device timing with HIP events on the library's stream --------
static hipEvent_t take_event(RtContext *ctx)
{
    if (!ctx->eventPool.empty()) {
        hipEvent_t e = ctx->eventPool.back();
        ctx->eventPool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

void rt_timer_begin(RtContext *ctx, const char *name)
{
    if (!ctx->timingOn) return;
    RtPendingEvent p;
    p.name = name;
    p.start = take_event(ctx);
    p.stop = take_event(ctx);
    (void)hipEventRecord(p.start, ctx->stream);
    ctx->pending.push_back(p);
}

void rt_timer_end(RtContext *ctx)
{
    if (!ctx->timingOn || ctx->pending.empty()) return;
    (void)hipEventRecord(ctx->pending.back().stop, ctx->stream);
}

void rt_timer_resolve(RtContext *ctx)
{
    if (ctx->pending.empty()) return;
    (void)hipStreamSynchronize(ctx->stream);
    for (RtPendingEvent &p : ctx->pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
            RtKernelTimer &t = ctx->timers[p.name];
            t.totalMs += ms;
            t.launches += 1;
        }
        ctx->eventPool.push_back(p.start);
        ctx->eventPool.push_back(p.stop);
    }
    ctx->pending.clear();
}

static int use_device(RtContext *ctx)
{
    RT_HIP(ctx, hipSetDevice(ctx->device));
    return RT_OK;
}

// The calls a host makes once per frame (rt_set_params, rt_dispatch, rt_dispatch_shadowmap, rt_accumulate, rt_set_dirty)
// leave the two frame lanes running side by side; every other call first makes lane 0's stream wait for lane 1's, so
// that what it then does on, or waits for with, ctx->stream covers all work in flight.
#define RT_ENTER_FRAME(ctx)                             \
    if (!(ctx)) return RT_E_INVALID;                    \
    {                                                   \
        int rc__ = use_device(ctx);                     \
        if (rc__) return rc__;                          \
    }

#define RT_ENTER(ctx)                                   \
    RT_ENTER_FRAME(ctx);                                \
    {                                                   \
        int rc__ = rt_join_lanes(ctx);                  \
        if (rc__) return rc__;                          \
    }

// Entry points that change what a ShadowMap pass started by rt_dispatch works on (or reuse its buffers) drop that pass first.
#define RT_ENTER_CHANGING(ctx)                                   \
    RT_ENTER(ctx);                                               \
    do {                                                         \
        const int eagerRc_ = rt_eager_cancel(ctx);               \
        if (eagerRc_) return eagerRc_;                           \
    } while (0)

// had: how many elements *dst was made for (it is kept when that is `count`)
template <typename T>
static int upload(RtContext *ctx, T **dst, const T *src, size_t count, size_t had = ~(size_t)0)
{
    if (*dst && had != count) {
        (void)hipFree(*dst);
        *dst = nullptr;
    }
    if (!*dst) RT_HIP(ctx, hipMalloc((void **)dst, (count ? count : 1) * sizeof(T)));
    if (count) RT_HIP(ctx, hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
    return RT_OK;
}

static int check_error_flag(RtContext *ctx)
{
    uint32_t flag = 0;
    RT_HIP(ctx, hipMemcpy(&flag, ctx->dErrorFlag, 4, hipMemcpyDeviceToHost));
    if (flag) {
        (void)hipMemset(ctx->dErrorFlag, 0, 4);
        if (flag == 2u) return rt_fail(ctx, RT_E_STATE, "BVH build: a box does not fit the 16-bit grid of the root box");
        return rt_fail(ctx, RT_E_OVERFLOW, "traversal stack overflow (depth > %d)", RT_STACK_DEPTH);
    }
    return RT_OK;
}

// streams and events of the frame lanes 1 .. RT_FRAME_LANES - 1 (lane 1's stream is stream2, made by the caller)
static bool rt_create_lanes(RtContext *ctx)
{
    ctx->laneStream[1] = ctx->stream2;
    for (int k = 0; k < RT_FRAME_LANES; ++k) {
        if (k >= 2 && hipStreamCreateWithFlags(&ctx->laneStream[k], hipStreamNonBlocking) != hipSuccess) return false;
        if (k >= 1 && (hipEventCreateWithFlags(&ctx->laneFork[k], hipEventDisableTiming) != hipSuccess ||
                       hipEventCreateWithFlags(&ctx->laneJoin[k], hipEventDisableTiming) != hipSuccess ||
                       hipEventCreateWithFlags(&ctx->evLaneIdle[k], hipEventDisableTiming) != hipSuccess))
            return false;
        if (hipEventCreateWithFlags(&ctx->evAccumulated[k], hipEventDisableTiming) != hipSuccess) return false;
    }
    return true;
}

extern "C" {

const char *rt_version(void) { return "rtcore 0.1 (gfx950)"; }

int rt_create(int device_id, RtContext **out_ctx)
{
    if (!out_ctx) return RT_E_INVALID;
    *out_ctx = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return RT_E_DEVICE;
    if (device_id < 0 || device_id >= n) return RT_E_INVALID;
    RtContext *ctx = new (std::nothrow) RtContext();
    if (!ctx) return RT_E_NOMEM;
    ctx->device = device_id;
    memset(&ctx->params, 0, sizeof ctx->params);
    memset(&ctx->tile, 0, sizeof ctx->tile);
    memset(&ctx->rootBox, 0, sizeof ctx->rootBox);
    memset(&ctx->brute, 0, sizeof ctx->brute);
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0) ctx->numCU = cus;
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->evFork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->evJoin, hipEventDisableTiming) != hipSuccess ||
        !rt_create_lanes(ctx) ||
        hipMalloc((void **)&ctx->dStats, 64 * sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc((void **)&ctx->dNonCasters, RT_NONCASTER_LIST * 64) != hipSuccess ||
        hipMalloc((void **)&ctx->dErrorFlag, 4) != hipSuccess || hipMalloc((void **)&ctx->dQCount, RT_COUNTER_BLOCK_WORDS * 4) != hipSuccess ||
        hipMalloc((void **)&ctx->dStackSpill, (size_t)RT_SPILL_AREAS * ctx->numCU * RT_SPILL_WORDS_PER_CU * 4) != hipSuccess) {
        delete ctx;
        return RT_E_DEVICE;
    }
    (void)hipMemset(ctx->dStats, 0, 64 * sizeof(unsigned long long));
    (void)hipMemset(ctx->dErrorFlag, 0, 4);
    (void)hipMemset(ctx->dQCount, 0, RT_COUNTER_BLOCK_WORDS * 4);
    if (rt_static_scratch_init(ctx) != RT_OK) {    // the address of this device's static scratch array (rt_trace_rays_device_static)
        (void)rt_destroy(ctx);
        return RT_E_DEVICE;
    }
    *out_ctx = ctx;
    return RT_OK;
}

int rt_destroy(RtContext *ctx)
{
    if (!ctx) return RT_OK;
    (void)hipSetDevice(ctx->device);
    // nothing may still run when the buffers go: a ShadowMap pass that rt_dispatch started walks its rays on the masked
    // stream shared with other contexts; the main stream is made to wait for it, then drained
    if (ctx->stream) {
        (void)rt_eager_cancel(ctx);
        (void)rt_join_lanes(ctx);
        (void)hipStreamSynchronize(ctx->stream);
    }
    for (int k = 1; k < RT_FRAME_LANES; ++k)
        if (ctx->laneStream[k]) (void)hipStreamSynchronize(ctx->laneStream[k]);
    for (hipStream_t st : {ctx->streamShadow, ctx->streamChain})
        if (st) (void)hipStreamSynchronize(st);
    for (int k = 1; k < RT_FRAME_LANES; ++k)
        for (hipStream_t st : {ctx->laneShadow[k], ctx->laneChain[k]})
            if (st) (void)hipStreamSynchronize(st);
    rt_timer_resolve(ctx);
    for (hipEvent_t e : ctx->eventPool) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->buildEvents)
        if (e) (void)hipEventDestroy(e);
    rt_free_build(ctx);
    rt_free_render(ctx);
    rt_free_post(ctx);
    rt_free_brute(ctx);
    void *ptrs[] = {ctx->dVerts, ctx->dIdx, ctx->dMatIdx, ctx->dShadow, ctx->dMats, ctx->dSphere, ctx->dArea,
                    ctx->dDisc, ctx->dCube, ctx->dStats, ctx->dErrorFlag, ctx->dQCount, ctx->dStackSpill, ctx->dNonCasters};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (ctx->streamCopy) {
        (void)hipStreamSynchronize(ctx->streamCopy);
        (void)hipStreamDestroy(ctx->streamCopy);
        (void)hipEventDestroy(ctx->evExpanded);
        (void)hipEventDestroy(ctx->evCopied[0]);
        (void)hipEventDestroy(ctx->evCopied[1]);
    }
    if (ctx->dReadbackFull) (void)hipFree(ctx->dReadbackFull);
    for (int k = 1; k < RT_FRAME_LANES; ++k)
        if (ctx->laneStream[k]) (void)hipStreamSynchronize(ctx->laneStream[k]), (void)hipStreamDestroy(ctx->laneStream[k]);
    rt_release_overlap_streams(ctx);
    for (hipEvent_t e : {ctx->evFork, ctx->evJoin})
        if (e) (void)hipEventDestroy(e);
    for (int k = 0; k < RT_FRAME_LANES; ++k)
        for (hipEvent_t e : {ctx->laneFork[k], ctx->laneJoin[k], ctx->evAccumulated[k], ctx->evLaneIdle[k]})
            if (e) (void)hipEventDestroy(e);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return RT_OK;
}

const char *rt_last_error(RtContext *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int rt_scene_upload(RtContext *ctx, const RtMeshVertex *vertices, uint32_t n_vertices, const uint32_t *indices,
                    uint32_t n_indices, const uint32_t *material_index, const int32_t *shadow_index_xy, uint32_t n_triangles)
{
    RT_ENTER_CHANGING(ctx);
    if (n_indices != 3ull * n_triangles) return rt_fail(ctx, RT_E_INVALID, "n_indices (%u) != 3 * n_triangles (%u)", n_indices, n_triangles);
    if (n_triangles && (!vertices || !indices || !material_index || !shadow_index_xy))
        return rt_fail(ctx, RT_E_INVALID, "null scene array");
    if (n_triangles >= 0x7FFFFFFFu) return rt_fail(ctx, RT_E_INVALID, "too many triangles");
    // Both passes over the triangles below are shared out over host threads (a scene whose vertices move comes through
    // here every time: 3.1 ms of the 496 k-triangle car's upload were these loops on one thread).
    const uint32_t nThreads = n_triangles < (1u << 16) ? 1u : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    struct Part {
        uint32_t badAt = 0xFFFFFFFFu;       // first index out of range in the part's triangles
        RtAABB nc, ca;
        uint32_t nNc = 0;
        float records[RT_NONCASTER_LIST][16];
    };
    std::vector<Part> parts(nThreads);
    auto work = [&](uint32_t k) {
        Part &part = parts[k];
        const uint32_t t0 = (uint32_t)((uint64_t)n_triangles * k / nThreads), t1 = (uint32_t)((uint64_t)n_triangles * (k + 1) / nThreads);
        for (size_t i = 3 * (size_t)t0; i < 3 * (size_t)t1; ++i)   // every index a kernel dereferences is checked here
            if (indices[i] >= n_vertices) {
                part.badAt = (uint32_t)i;
                return;
            }
        // Boxes round the OWN BOXES (as the build makes them, k_calculate: vertex extremes -+ 0.001, the same float
        // operations) of the triangles that cast no shadow and of those that do (MeshData.compute:127 reads the flag
        // from [0]); min > max: none.
        // While they are few, their records as well, in the layout and with the float operations of k_pack_leaves.
        RtAABB &nc = part.nc, &ca = part.ca;
        for (int a = 0; a < 3; ++a) nc.min[a] = ca.min[a] = rt_inf(), nc.max[a] = ca.max[a] = -rt_inf();
        nc._dummy0 = nc._dummy1 = ca._dummy0 = ca._dummy1 = 0.0f;
        for (uint32_t t = t0; t < t1; ++t) {
            const bool casts = shadow_index_xy[2 * (size_t)t] > 0;
            RtAABB &box = casts ? ca : nc;
            const float *p0 = vertices[indices[3 * (size_t)t]].position, *p1 = vertices[indices[3 * (size_t)t + 1]].position,
                        *p2 = vertices[indices[3 * (size_t)t + 2]].position;
            float mn[3], mx[3];
            for (int a = 0; a < 3; ++a) {
                mn[a] = rt_min(rt_min(p0[a], p1[a]), p2[a]) - 0.001f;
                mx[a] = rt_max(rt_max(p0[a], p1[a]), p2[a]) + 0.001f;
                if (!(mn[a] >= box.min[a])) box.min[a] = mn[a] == mn[a] ? mn[a] : -rt_inf();      // a NaN coordinate opens the box
                if (!(mx[a] <= box.max[a])) box.max[a] = mx[a] == mx[a] ? mx[a] : rt_inf();
            }
            if (casts) continue;
            if (part.nNc < RT_NONCASTER_LIST) {
                const float rec[16] = {p0[0], p0[1], p0[2], p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2], p2[0] - p0[0], p2[1] - p0[1],
                                       p2[2] - p0[2], 0.0f, mn[0], mn[1], mn[2], mx[0], mx[1], mx[2]};
                memcpy(part.records[part.nNc], rec, sizeof rec);
            }
            part.nNc += 1;
        }
    };
    {
        std::vector<std::thread> threads;
        uint32_t started = 1;                // parts [started, nThreads) are done here if no thread can be had for them
        try {
            for (; started < nThreads; ++started) threads.emplace_back(work, started);
        } catch (...) {
        }
        work(0);
        for (uint32_t k = started; k < nThreads; ++k) work(k);
        for (std::thread &th : threads) th.join();
    }
    for (const Part &part : parts)
        if (part.badAt != 0xFFFFFFFFu) return rt_fail(ctx, RT_E_INVALID, "index %u out of range at %u", indices[part.badAt], part.badAt);
    // the parts' boxes united (none of their bounds is a NaN), the records of the first few non-casters in triangle order
    RtAABB nc = parts[0].nc, ca = parts[0].ca;
    float records[RT_NONCASTER_LIST][16];
    uint32_t nNc = 0;
    for (uint32_t k = 0; k < nThreads; ++k) {
        const Part &part = parts[k];
        for (int a = 0; a < 3; ++a) {
            if (part.nc.min[a] < nc.min[a]) nc.min[a] = part.nc.min[a];
            if (part.nc.max[a] > nc.max[a]) nc.max[a] = part.nc.max[a];
            if (part.ca.min[a] < ca.min[a]) ca.min[a] = part.ca.min[a];
            if (part.ca.max[a] > ca.max[a]) ca.max[a] = part.ca.max[a];
        }
        for (uint32_t j = 0; j < part.nNc && j < RT_NONCASTER_LIST; ++j)
            if (nNc + j < RT_NONCASTER_LIST) memcpy(records[nNc + j], part.records[j], sizeof records[0]);
        nNc += part.nNc;
    }
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->nNonCasters = nNc;
    if (nNc && nNc <= RT_NONCASTER_LIST)
        RT_HIP(ctx, hipMemcpy(ctx->dNonCasters, records, (size_t)nNc * 64, hipMemcpyHostToDevice));
    ctx->nonCasterBox = nc;
    ctx->casterBox = ca;
    int rc;
    if ((rc = upload(ctx, &ctx->dVerts, vertices, n_vertices, (size_t)ctx->nVerts))) return rc;
    if ((rc = upload(ctx, &ctx->dIdx, indices, n_indices, 3 * (size_t)ctx->N))) return rc;
    if ((rc = upload(ctx, &ctx->dMatIdx, material_index, n_triangles, (size_t)ctx->N))) return rc;
    if ((rc = upload(ctx, &ctx->dShadow, shadow_index_xy, 2 * (size_t)n_triangles, 2 * (size_t)ctx->N))) return rc;
    if (n_triangles != ctx->N) rt_free_build(ctx);     // (as many triangles as before: the next build reuses the buffers)
    ctx->built = false;
    ctx->nVerts = n_vertices;
    ctx->N = n_triangles;
    ctx->sampleCount = 0;
    return RT_OK;
}

int rt_materials_upload(RtContext *ctx, const RtMaterialData *materials, uint32_t n)
{
    RT_ENTER_CHANGING(ctx);
    if (n && !materials) return rt_fail(ctx, RT_E_INVALID, "null materials");
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int rc;
    if ((rc = upload(ctx, &ctx->dMats, materials, n))) return rc;
    ctx->nMats = n;
    ctx->sampleCount = 0;   // BVH.UpdateMaterialData ends with RayTracing.SetDirty()
    return RT_OK;
}

int rt_lights_upload(RtContext *ctx, const RtSphereLight *sphere, uint32_t n_sphere, const RtAreaLight *area, uint32_t n_area,
                     const RtDiscLight *disc, uint32_t n_disc)
{
    RT_ENTER_CHANGING(ctx);
    if ((n_sphere && !sphere) || (n_area && !area) || (n_disc && !disc)) return rt_fail(ctx, RT_E_INVALID, "null light array");
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int rc;
    if ((rc = upload(ctx, &ctx->dSphere, sphere, n_sphere))) return rc;
    if ((rc = upload(ctx, &ctx->dArea, area, n_area))) return rc;
    if ((rc = upload(ctx, &ctx->dDisc, disc, n_disc))) return rc;
    ctx->nSphere = n_sphere;
    ctx->nArea = n_area;
    ctx->nDisc = n_disc;
    return RT_OK;
}

int rt_skybox_upload(RtContext *ctx, const float *faces_rgba, uint32_t size)
{
    RT_ENTER_CHANGING(ctx);
    if (size && !faces_rgba) return rt_fail(ctx, RT_E_INVALID, "null skybox");
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int rc;
    if ((rc = upload(ctx, &ctx->dCube, (const float4 *)faces_rgba, (size_t)6 * size * size))) return rc;
    ctx->cubeSize = size;
    return RT_OK;
}

int rt_bvh_build(RtContext *ctx)
{
    RT_ENTER_CHANGING(ctx);
    if (ctx->N && !ctx->dVerts) return rt_fail(ctx, RT_E_STATE, "rt_bvh_build before rt_scene_upload");
    const int rc = rt_build_impl(ctx);
    rt_timer_resolve(ctx);
    return rc ? rc : check_error_flag(ctx);
}

int rt_readback_bvh(RtContext *ctx, uint32_t *morton_codes, uint32_t *sorted_triangle_index, RtAABB *triangle_aabb,
                    RtTriangle *triangles, RtInternalNode *internal_nodes, RtLeafNode *leaf_nodes, RtAABB *bvh_data)
{
    RT_ENTER(ctx);
    if (!ctx->built) return rt_fail(ctx, RT_E_STATE, "rt_readback_bvh before rt_bvh_build");
    return rt_readback_bvh_impl(ctx, morton_codes, sorted_triangle_index, triangle_aabb, triangles, internal_nodes, leaf_nodes,
                                bvh_data);
}

int rt_readback_walk_tree(RtContext *ctx, uint32_t *nodes8, uint32_t *leaf_triangle, uint32_t *leaf_rank)
{
    RT_ENTER(ctx);
    if (!ctx->built) return rt_fail(ctx, RT_E_STATE, "rt_readback_walk_tree before rt_bvh_build");
    const uint32_t N = ctx->N;
    if (N == 0) return RT_OK;
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (nodes8 && N >= 2) RT_HIP(ctx, hipMemcpy(nodes8, ctx->dQNodes, (size_t)(N - 1) * sizeof(RtQNode), hipMemcpyDeviceToHost));
    if (leaf_rank) RT_HIP(ctx, hipMemcpy(leaf_rank, ctx->dLeafRank, (size_t)N * 4, hipMemcpyDeviceToHost));
    if (leaf_triangle) {
        std::vector<RtLeafTri> rec(N);
        RT_HIP(ctx, hipMemcpy(rec.data(), ctx->dLeafTri, (size_t)N * sizeof(RtLeafTri), hipMemcpyDeviceToHost));
        for (uint32_t j = 0; j < N; ++j) leaf_triangle[j] = rec[j].tri;
    }
    return RT_OK;
}

int rt_readback_sort_stages(RtContext *ctx, uint32_t *morton_unsorted, uint32_t *morton_sorted_raw)
{
    RT_ENTER(ctx);
    if (!ctx->built) return rt_fail(ctx, RT_E_STATE, "rt_readback_sort_stages before rt_bvh_build");
    if (ctx->N == 0) return RT_OK;
    if (morton_unsorted) RT_HIP(ctx, hipMemcpy(morton_unsorted, ctx->dMortonUnsorted, (size_t)ctx->N * 4, hipMemcpyDeviceToHost));
    if (morton_sorted_raw) RT_HIP(ctx, hipMemcpy(morton_sorted_raw, ctx->dMortonSortedRaw, (size_t)ctx->N * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_set_params(RtContext *ctx, const RtParams *p)
{
    RT_ENTER_FRAME(ctx);         // (a new seed, a moved camera: the frames in flight keep the uniforms they were launched with)
    {
        const int eagerRc = rt_eager_cancel(ctx);
        if (eagerRc) return eagerRc;
    }
    if (!p) return rt_fail(ctx, RT_E_INVALID, "null params");
    if (p->width <= 0 || p->height <= 0 || p->width > 32768 || p->height > 32768)
        return rt_fail(ctx, RT_E_INVALID, "bad image size %d x %d", p->width, p->height);
    RtParams q = *p;
    if (q.tileCount <= 1) { q.tileCount = 1; q.tileRank = 0; }
    if (q.tileSize <= 0) q.tileSize = 64;
    if (q.tileSize % 8 != 0 || q.tileSize > 1024) return rt_fail(ctx, RT_E_INVALID, "tileSize must be a multiple of 8, <= 1024");
    if (q.tileRank < 0 || q.tileRank >= q.tileCount) return rt_fail(ctx, RT_E_INVALID, "tileRank %d outside [0,%d)", q.tileRank, q.tileCount);
    RtTileMap tm;
    tm.width = q.width;
    tm.height = q.height;
    tm.tileSize = q.tileSize;
    tm.tilesX = (q.width + q.tileSize - 1) / q.tileSize;
    tm.tilesY = (q.height + q.tileSize - 1) / q.tileSize;
    tm.rank = q.tileRank;
    tm.count = q.tileCount;
    const uint32_t T = (uint32_t)tm.tilesX * (uint32_t)tm.tilesY;
    tm.nLocalTiles = (T + (uint32_t)q.tileCount - 1) / (uint32_t)q.tileCount;   // one of every group of tileCount tiles; in the last group that may be a tile beyond the image (padding)
    tm.nSlots = tm.nLocalTiles * (uint32_t)(q.tileSize * q.tileSize);
    tm.finish();
    // ray indices are 32-bit: 16 shadow rays per path slot of one frame must fit
    if ((uint64_t)tm.nLocalTiles * (uint64_t)(q.tileSize * q.tileSize) * 16ull >= (1ull << 32))
        return rt_fail(ctx, RT_E_INVALID, "%d x %d pixels are too many for one context: split the frame into more tiles (tileCount)", q.width, q.height);
    const bool geometryChanged = !ctx->paramsSet || memcmp(&tm, &ctx->tile, sizeof tm) != 0;
    ctx->params = q;
    ctx->tile = tm;
    ctx->paramsSet = true;
    if (geometryChanged) {
        const int joinRc = rt_join_lanes(ctx);
        if (joinRc) return joinRc;
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return rt_alloc_render(ctx);   // new render targets: InitRT resets samplePrePixel (RayTracing.cs:172-187)
    }
    return RT_OK;
}

// csMain: CSMain will run (needs the BVH, or the primitive lists when the BVH keyword is off);
// shadowMap: the ShadowMap kernel will run (always walks the BVH, RayTracing.compute:112).
static int ready_to_render(RtContext *ctx, bool csMain, bool shadowMap)
{
    if (!ctx->paramsSet) return rt_fail(ctx, RT_E_STATE, "rt_set_params has not been called");
    const bool brute = (ctx->params.flags & RT_FLAG_BRUTE_FORCE) != 0;
    if (csMain && brute && !ctx->bruteSet) return rt_fail(ctx, RT_E_STATE, "RT_FLAG_BRUTE_FORCE is set but rt_primitives_upload has not been called");
    if (((csMain && !brute) || shadowMap) && !ctx->built) return rt_fail(ctx, RT_E_STATE, "rt_bvh_build has not been called");
    return RT_OK;
}

int rt_dispatch(RtContext *ctx)
{
    RT_ENTER_FRAME(ctx);
    int rc = ready_to_render(ctx, true, false);
    if (rc) return rc;   // CustomRender returns silently when not initialised (RayTracing.cs:120-121); here it is an error code
    return rt_dispatch_impl(ctx);
}

int rt_dispatch_shadowmap(RtContext *ctx)
{
    RT_ENTER_FRAME(ctx);
    int rc = ready_to_render(ctx, false, true);
    if (rc) return rc;
    return rt_shadowmap_impl(ctx);
}

int rt_accumulate(RtContext *ctx, uint32_t sample_per_pixel, int use_shadow_map)
{
    RT_ENTER_FRAME(ctx);
    if (!ctx->paramsSet) return rt_fail(ctx, RT_E_STATE, "rt_set_params has not been called");
    return rt_accumulate_impl(ctx, sample_per_pixel, use_shadow_map);
}

int rt_render_frames(RtContext *ctx, const float *seeds, uint32_t n_frames)
{
    RT_ENTER(ctx);
    int rc = ready_to_render(ctx, true, ctx->paramsSet && (ctx->params.flags & RT_FLAG_MATTE_MASK) != 0);
    if (rc) return rc;
    if (n_frames && !seeds) return rt_fail(ctx, RT_E_INVALID, "null seeds");
    RtArena arena(ctx->stream);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    RT_HIP(ctx, arena.event(&e0));
    RT_HIP(ctx, arena.event(&e1));
    RT_HIP(ctx, hipEventRecord(e0, ctx->stream));
    if ((rc = rt_render_frames_impl(ctx, seeds, n_frames))) return rc;
    RT_HIP(ctx, hipEventRecord(e1, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0;
    RT_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    ctx->renderMs = ms;
    rt_timer_resolve(ctx);
    return check_error_flag(ctx);
}

int rt_set_dirty(RtContext *ctx)
{
    if (!ctx) return RT_E_INVALID;
    ctx->sampleCount = 0;
    return RT_OK;
}

uint32_t rt_sample_count(RtContext *ctx) { return ctx ? ctx->sampleCount : 0; }
uint32_t rt_local_pixel_count(RtContext *ctx) { return ctx && ctx->paramsSet ? ctx->tile.nSlots : 0; }

}   // extern "C"

float4 *rt_image_ptr(RtContext *ctx, int which)
{
    const size_t off = (size_t)ctx->lastFrame * ctx->tile.nSlots;   // the newest frame of the last batch
    switch (which) {
    case RT_IMAGE_RT: return ctx->dRt ? ctx->dRt + off : nullptr;
    case RT_IMAGE_CONVERGED: return ctx->dConv;
    case RT_IMAGE_SHADOWMAP: return ctx->dShadowMap ? ctx->dShadowMap + off : nullptr;
    default: return nullptr;
    }
}

extern "C" {

int rt_readback(RtContext *ctx, int which, float *dst_rgba)
{
    RT_ENTER(ctx);
    if (!dst_rgba) return rt_fail(ctx, RT_E_INVALID, "null destination");
    if (which == RT_IMAGE_POSTPROCESS || which == RT_IMAGE_USER) return rt_post_readback_impl(ctx, which, dst_rgba);
    if (!ctx->paramsSet) return rt_fail(ctx, RT_E_STATE, "rt_set_params has not been called");
    float4 *src = rt_image_ptr(ctx, which);
    if (!src) return rt_fail(ctx, RT_E_INVALID, "bad image selector %d", which);
    const size_t px = (size_t)ctx->params.width * ctx->params.height;
    float4 *full = nullptr;
    RtArena arena(ctx->stream);
    RT_HIP(ctx, arena.alloc(&full, px));
    RT_HIP(ctx, hipMemsetAsync(full, 0, px * sizeof(float4), ctx->stream));
    int rc = rt_expand_local_impl(ctx, src, full);
    if (rc) return rc;
    RT_HIP(ctx, hipMemcpyAsync(dst_rgba, full, px * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    rt_timer_resolve(ctx);
    return check_error_flag(ctx);
}

int rt_readback_merge(RtContext *ctx, int which, float *frame_rgba)
{
    RT_ENTER(ctx);
    if (!ctx->paramsSet) return rt_fail(ctx, RT_E_STATE, "rt_set_params has not been called");
    float4 *src = rt_image_ptr(ctx, which);
    if (!src || !frame_rgba) return rt_fail(ctx, RT_E_INVALID, "bad image selector or null destination");
    const uint32_t n = ctx->tile.nSlots;
    std::vector<float> local((size_t)n * 4);
    RT_HIP(ctx, hipMemcpyAsync(local.data(), src, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (uint32_t s = 0; s < n; ++s) {          // only this context's pixels are written: contexts that split one frame merge into one buffer
        uint32_t px, py;
        if (ctx->tile.slot_to_pixel(s, &px, &py)) memcpy(frame_rgba + ((size_t)py * ctx->params.width + px) * 4, &local[(size_t)s * 4], 16);
    }
    rt_timer_resolve(ctx);
    return check_error_flag(ctx);
}

/* The frame leaves in the background: the tiles are expanded into a full frame in device memory on the context's stream
 * (so the images are free to be rendered into again as soon as the call returns), and a copy stream takes that frame to
 * the caller's memory while the next frames are rendered. */
int rt_readback_async(RtContext *ctx, int which, float *pinned_dst_rgba)
{
    RT_ENTER_FRAME(ctx);          // a per-frame call: the two frame lanes keep running
    if (!pinned_dst_rgba) return rt_fail(ctx, RT_E_INVALID, "null destination");
    if (!ctx->paramsSet) return rt_fail(ctx, RT_E_STATE, "rt_set_params has not been called");
    if (which != RT_IMAGE_RT && which != RT_IMAGE_CONVERGED && which != RT_IMAGE_SHADOWMAP)
        return rt_fail(ctx, RT_E_INVALID, "rt_readback_async: RT_IMAGE_RT / _CONVERGED / _SHADOWMAP only");
    float4 *src = rt_image_ptr(ctx, which);
    if (!src) return rt_fail(ctx, RT_E_INVALID, "bad image selector %d", which);
    const size_t px = (size_t)ctx->params.width * ctx->params.height;
    if (!ctx->streamCopy) {
        RT_HIP(ctx, hipStreamCreateWithFlags(&ctx->streamCopy, hipStreamNonBlocking));
        RT_HIP(ctx, hipEventCreateWithFlags(&ctx->evExpanded, hipEventDisableTiming));
        RT_HIP(ctx, hipEventCreateWithFlags(&ctx->evCopied[0], hipEventDisableTiming));
        RT_HIP(ctx, hipEventCreateWithFlags(&ctx->evCopied[1], hipEventDisableTiming));
    }
    if (ctx->readbackPixels != px) {
        for (int k = 0; k < 2; ++k) {
            if (ctx->copyPending[k]) RT_HIP(ctx, hipEventSynchronize(ctx->evCopied[k]));
            ctx->copyPending[k] = false;
        }
        if (ctx->dReadbackFull) (void)hipFree(ctx->dReadbackFull);
        ctx->dReadbackFull = nullptr;
        ctx->readbackPixels = 0;
        RT_HIP(ctx, hipMalloc((void **)&ctx->dReadbackFull, 2 * (px ? px : 1) * sizeof(float4)));
        ctx->readbackPixels = px;
    }
    return rt_readback_async_impl(ctx, src, pinned_dst_rgba);
}

int rt_readback_wait(RtContext *ctx)
{
    RT_ENTER_FRAME(ctx);
    bool any = false;
    for (int k = 0; k < 2; ++k) {
        if (!ctx->copyPending[k]) continue;
        RT_HIP(ctx, hipEventSynchronize(ctx->evCopied[k]));
        ctx->copyPending[k] = false;
        any = true;
    }
    return any ? check_error_flag(ctx) : RT_OK;
}

/* This context's tiles into a full frame that already lives in device memory, on this context's device or on another
 * one of the process: the device-side form of rt_readback_merge.  The frame's device writes the pixels where this
 * device cannot reach its memory. */
int rt_merge_local_device(RtContext *ctx, int which, void *device_frame_rgba, int frame_device)
{
    RT_ENTER(ctx);
    if (!ctx->paramsSet) return rt_fail(ctx, RT_E_STATE, "rt_set_params has not been called");
    float4 *src = rt_image_ptr(ctx, which);
    if (!src || !device_frame_rgba) return rt_fail(ctx, RT_E_INVALID, "bad image selector or null destination");
    if (which != RT_IMAGE_RT && which != RT_IMAGE_CONVERGED && which != RT_IMAGE_SHADOWMAP)
        return rt_fail(ctx, RT_E_INVALID, "rt_merge_local_device: RT_IMAGE_RT / _CONVERGED / _SHADOWMAP only");
    int nDev = 0;
    RT_HIP(ctx, hipGetDeviceCount(&nDev));
    if (frame_device < 0 || frame_device >= nDev) return rt_fail(ctx, RT_E_INVALID, "frame_device %d out of range", frame_device);
    if (frame_device != ctx->device) {
        int can = 0;
        RT_HIP(ctx, hipDeviceCanAccessPeer(&can, ctx->device, frame_device));
        if (can) {
            const hipError_t e = hipDeviceEnablePeerAccess(frame_device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) can = 0;
            (void)hipGetLastError();
        }
        if (!can) {
            // no direct path: the tiles travel as they are (one contiguous copy) and the frame's own device places them
            const size_t bytes = (size_t)ctx->tile.nSlots * sizeof(float4);
            RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            float4 *staged = nullptr;
            RT_HIP(ctx, hipSetDevice(frame_device));
            hipError_t e = hipMalloc((void **)&staged, bytes ? bytes : 16);
            if (e == hipSuccess) e = hipMemcpyPeer(staged, frame_device, src, ctx->device, bytes);
            int rc = RT_OK;
            if (e == hipSuccess) {
                rc = rt_expand_local_on(ctx, nullptr, staged, (float4 *)device_frame_rgba);
                if (rc == RT_OK) e = hipDeviceSynchronize();
            }
            if (staged) (void)hipFree(staged);
            (void)hipSetDevice(ctx->device);
            if (e != hipSuccess) return rt_fail(ctx, RT_E_DEVICE, "rt_merge_local_device: staging on device %d failed: %s", frame_device, hipGetErrorString(e));
            return rc;
        }
    }
    int rc = rt_expand_local_impl(ctx, src, (float4 *)device_frame_rgba);      // writes this context's pixels only
    if (rc) return rc;
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_readback_local_device(RtContext *ctx, int which, void *device_dst)
{
    RT_ENTER(ctx);
    if (!ctx->paramsSet) return rt_fail(ctx, RT_E_STATE, "rt_set_params has not been called");
    float4 *src = rt_image_ptr(ctx, which);
    if (!src || !device_dst) return rt_fail(ctx, RT_E_INVALID, "bad image selector or null destination");
    RT_HIP(ctx, hipMemcpyAsync(device_dst, src, (size_t)ctx->tile.nSlots * sizeof(float4), hipMemcpyDeviceToDevice, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_upload_converged(RtContext *ctx, const float *src_rgba, uint32_t sample_per_pixel)
{
    RT_ENTER(ctx);
    if (!ctx->paramsSet) return rt_fail(ctx, RT_E_STATE, "rt_set_params has not been called");
    if (!src_rgba) return rt_fail(ctx, RT_E_INVALID, "null source");
    // full frame -> tile-local order on the host (rare, checkpoint restore only)
    const uint32_t n = ctx->tile.nSlots;
    std::vector<float> local((size_t)n * 4, 0.0f);
    for (uint32_t s = 0; s < n; ++s) {
        uint32_t px, py;
        if (ctx->tile.slot_to_pixel(s, &px, &py)) memcpy(&local[(size_t)s * 4], src_rgba + ((size_t)py * ctx->params.width + px) * 4, 16);
    }
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    RT_HIP(ctx, hipMemcpy(ctx->dConv, local.data(), (size_t)n * 16, hipMemcpyHostToDevice));
    ctx->sampleCount = sample_per_pixel;
    return RT_OK;
}

int rt_local_pixel_map(RtContext *ctx, uint32_t *dst)
{
    if (!ctx || !dst) return RT_E_INVALID;
    if (!ctx->paramsSet) return rt_fail(ctx, RT_E_STATE, "rt_set_params has not been called");
    for (uint32_t s = 0; s < ctx->tile.nSlots; ++s) {
        uint32_t px, py;
        dst[s] = ctx->tile.slot_to_pixel(s, &px, &py) ? py * (uint32_t)ctx->params.width + px : 0xFFFFFFFFu;
    }
    return RT_OK;
}

int rt_trace_rays_device(RtContext *ctx, const void *device_origins4, const void *device_dirs4, uint32_t n, void *device_hits4)
{
    RT_ENTER_CHANGING(ctx);
    if (!ctx->built) return rt_fail(ctx, RT_E_STATE, "rt_bvh_build has not been called");
    if (n && (!device_origins4 || !device_dirs4 || !device_hits4)) return rt_fail(ctx, RT_E_INVALID, "null ray buffers");
    RtArena arena(ctx->stream);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    RT_HIP(ctx, arena.event(&e0));
    RT_HIP(ctx, arena.event(&e1));
    RT_HIP(ctx, hipEventRecord(e0, ctx->stream));
    int rc = rt_trace_device_impl(ctx, (const float4 *)device_origins4, (const float4 *)device_dirs4, n, (RtHitRec *)device_hits4);
    if (rc) return rc;
    RT_HIP(ctx, hipEventRecord(e1, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0;
    RT_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    ctx->renderMs = ms;
    rt_timer_resolve(ctx);
    return check_error_flag(ctx);
}

int rt_trace_rays_device_static(RtContext *ctx, const void *device_origins4, const void *device_dirs4, uint32_t n, void *device_hits4)
{
    RT_ENTER_CHANGING(ctx);
    if (!ctx->built) return rt_fail(ctx, RT_E_STATE, "rt_bvh_build has not been called");
    if (n > RT_TRACE_STATIC_MAX_RAYS)
        return rt_fail(ctx, RT_E_INVALID, "rt_trace_rays_device_static: %u rays exceed the maximum of %u per call", n, RT_TRACE_STATIC_MAX_RAYS);
    if (n && (!device_origins4 || !device_dirs4 || !device_hits4)) return rt_fail(ctx, RT_E_INVALID, "null ray buffers");
    return rt_trace_device_static_impl(ctx, (const float4 *)device_origins4, (const float4 *)device_dirs4, n, (RtHitRec *)device_hits4);
}

int rt_trace_rays(RtContext *ctx, const float *origins, const float *directions, uint32_t n, float *out_t, uint32_t *out_triangle,
                  float *out_u, float *out_v)
{
    RT_ENTER_CHANGING(ctx);
    if (!ctx->built) return rt_fail(ctx, RT_E_STATE, "rt_bvh_build has not been called");
    if (n == 0) return RT_OK;
    if (!origins || !directions) return rt_fail(ctx, RT_E_INVALID, "null rays");
    float *d3 = nullptr;       // [o(3n) | d(3n)]
    float4 *d4 = nullptr;      // [o4(n) | d4(n)]
    RtHitRec *dh = nullptr;
    float *dout = nullptr;     // [t | tri | u | v] x n
    RtArena arena(ctx->stream);
    RT_HIP(ctx, arena.alloc(&d3, (size_t)n * 6));
    RT_HIP(ctx, arena.alloc(&d4, (size_t)n * 2));
    RT_HIP(ctx, arena.alloc(&dh, (size_t)n));
    RT_HIP(ctx, arena.alloc(&dout, (size_t)n * 4));
    RT_HIP(ctx, hipMemcpyAsync(d3, origins, (size_t)n * 12, hipMemcpyHostToDevice, ctx->stream));
    RT_HIP(ctx, hipMemcpyAsync(d3 + (size_t)3 * n, directions, (size_t)n * 12, hipMemcpyHostToDevice, ctx->stream));
    hipEvent_t e0 = nullptr, e1 = nullptr;     // round the walk itself: RtStats.renderMs of this call
    RT_HIP(ctx, arena.event(&e0));
    RT_HIP(ctx, arena.event(&e1));
    int rc = rt_pack_rays_impl(ctx, d3, d3 + (size_t)3 * n, n, d4, d4 + n);
    if (rc) return rc;
    RT_HIP(ctx, hipEventRecord(e0, ctx->stream));
    rc = rt_trace_device_impl(ctx, d4, d4 + n, n, dh);
    if (rc) return rc;
    RT_HIP(ctx, hipEventRecord(e1, ctx->stream));
    rc = rt_hits_to_host_impl(ctx, dh, n, dout, (uint32_t *)(dout + n), dout + (size_t)2 * n, dout + (size_t)3 * n);
    if (rc) return rc;
    if (out_t) RT_HIP(ctx, hipMemcpyAsync(out_t, dout, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (out_triangle) RT_HIP(ctx, hipMemcpyAsync(out_triangle, dout + n, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (out_u) RT_HIP(ctx, hipMemcpyAsync(out_u, dout + (size_t)2 * n, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (out_v) RT_HIP(ctx, hipMemcpyAsync(out_v, dout + (size_t)3 * n, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0;
    RT_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    ctx->renderMs = ms;
    rt_timer_resolve(ctx);
    return check_error_flag(ctx);
}

int rt_debug_eval(RtContext *ctx, int which, const float *x, const float *y, uint32_t n, float *out)
{
    RT_ENTER_CHANGING(ctx);
    if (!x || !y || !out || n == 0) return rt_fail(ctx, RT_E_INVALID, "bad rt_debug_eval arguments");
    const uint32_t nin = which == 7 ? 2u : (which == 8 ? 15u : n);
    float *dx = nullptr, *dy = nullptr, *dout = nullptr;
    RtArena arena(ctx->stream);
    RT_HIP(ctx, arena.alloc(&dx, (size_t)nin));
    RT_HIP(ctx, arena.alloc(&dy, (size_t)nin));
    RT_HIP(ctx, arena.alloc(&dout, (size_t)n));
    RT_HIP(ctx, hipMemcpyAsync(dx, x, (size_t)nin * 4, hipMemcpyHostToDevice, ctx->stream));
    RT_HIP(ctx, hipMemcpyAsync(dy, y, (size_t)(which == 7 || which == 8 ? 1 : nin) * 4, hipMemcpyHostToDevice, ctx->stream));
    int rc = rt_debug_eval_impl(ctx, which, dx, dy, n, dout);
    if (rc) return rc;
    RT_HIP(ctx, hipMemcpyAsync(out, dout, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_set_option(RtContext *ctx, int option, int value)
{
    RT_ENTER_CHANGING(ctx);
    switch (option) {
    case RT_OPT_EAGER_SHADOWMAP: ctx->optEagerShadowMap = value != 0; return RT_OK;
    case RT_OPT_PRUNE_BY_DISTANCE:
        if (value < 0 || value > 2) return rt_fail(ctx, RT_E_INVALID, "RT_OPT_PRUNE_BY_DISTANCE must be 0, 1 or 2");
        ctx->optPruneByDistance = value;
        return RT_OK;
    case RT_OPT_SHARE_LONG_WALKS: ctx->optShareLongWalks = value != 0; return RT_OK;
    case RT_OPT_WIDE_NODES: ctx->optWideNodes = value != 0; return RT_OK;
    case RT_OPT_PIPELINE_FRAMES:
        if (value < 0 || value > RT_FRAME_LANES) return rt_fail(ctx, RT_E_INVALID, "RT_OPT_PIPELINE_FRAMES must be 0 .. %d", RT_FRAME_LANES);
        ctx->optPipelineFrames = value == 0 ? 1 : (value == 1 ? RT_FRAME_LANES : value);
        return RT_OK;
    case RT_OPT_CHAIN_BOUNCES:
        if (value < 0 || value >= RT_MAX_BOUNCE) return rt_fail(ctx, RT_E_INVALID, "RT_OPT_CHAIN_BOUNCES must be 0 .. %d", RT_MAX_BOUNCE - 1);
        ctx->optChainBounces = value;
        return RT_OK;
    case RT_OPT_PACKET_SHADOW: ctx->optPacketShadow = value != 0; return RT_OK;
    case RT_OPT_PACKET_PRIMARY: ctx->optPacketPrimary = value != 0; return RT_OK;
    case RT_OPT_PACKET_TRACE: ctx->optPacketTrace = value != 0; return RT_OK;
    case RT_OPT_MATERIAL_SORT: ctx->optMaterialSort = value != 0; return RT_OK;
    case RT_OPT_TWO_STREAMS:
        if (value < -1 || value > RT_FRAME_LANES) return rt_fail(ctx, RT_E_INVALID, "RT_OPT_TWO_STREAMS must be -1 .. %d", RT_FRAME_LANES);
        ctx->optTwoStreams = value;
        return RT_OK;
    case RT_OPT_CASTER_BOX: ctx->optCasterBox = value != 0; return RT_OK;
    case RT_OPT_FIRST_HIT_SHADOW: ctx->optFirstHitShadow = value != 0; return RT_OK;
    case RT_OPT_OVERLAP_SHADOW: ctx->optOverlapShadow = value != 0; return RT_OK;
    case RT_OPT_OVERLAP_AFTER:
        if (value < 0 || value > 6) return rt_fail(ctx, RT_E_INVALID, "RT_OPT_OVERLAP_AFTER must be 0..6");
        ctx->optOverlapAfter = (uint32_t)value;
        return RT_OK;
    case RT_OPT_OVERLAP_CUS:
        if (value < 0 || value >= ctx->numCU) return rt_fail(ctx, RT_E_INVALID, "RT_OPT_OVERLAP_CUS must be 0 (the library chooses) .. compute units - 1");
        ctx->optOverlapCUs = (uint32_t)value;
        return RT_OK;
    case RT_OPT_BATCH_FRAMES:
        if (value < 0 || value > 64) return rt_fail(ctx, RT_E_INVALID, "RT_OPT_BATCH_FRAMES must be 0..64");
        ctx->optBatchFrames = (uint32_t)value;
        return RT_OK;
    default: return rt_fail(ctx, RT_E_INVALID, "unknown option");
    }
}

int rt_enable_counters(RtContext *ctx, int enable)
{
    RT_ENTER_CHANGING(ctx);
    ctx->countersOn = enable != 0;
    return RT_OK;
}

int rt_enable_kernel_timing(RtContext *ctx, int enable)
{
    RT_ENTER_CHANGING(ctx);
    rt_timer_resolve(ctx);
    ctx->timingOn = enable != 0;
    return RT_OK;
}

int rt_reset_stats(RtContext *ctx)
{
    RT_ENTER(ctx);
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    rt_timer_resolve(ctx);
    ctx->timers.clear();
    RT_HIP(ctx, hipMemset(ctx->dStats, 0, 64 * sizeof(unsigned long long)));
    return RT_OK;
}

int rt_get_stats(RtContext *ctx, RtStats *out)
{
    RT_ENTER(ctx);
    if (!out) return rt_fail(ctx, RT_E_INVALID, "null stats");
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    unsigned long long h[40];   // [0..4] all launches but the shadow rays, [32..36] shadow-ray launches
    RT_HIP(ctx, hipMemcpy(h, ctx->dStats, sizeof h, hipMemcpyDeviceToHost));
    out->rays = h[0] + h[32] + h[37];       // [37]: shadow rays of pixels that got none (k_sm_setup)
    out->raysWalked = h[0] + h[32];
    out->shadowRaysWalked = h[32];
    out->internalVisits = h[1] + h[33];
    out->leafVisits = h[2] + h[34];
    out->triTests = h[3] + h[35];
    out->hits = h[4] + h[36];
    out->paths = h[5];
    out->shadowRays = h[32] + h[37];
    out->shadowInternalVisits = h[33];
    out->shadowLeafVisits = h[34];
    out->shadowTriTests = h[35];
    out->shadowHits = h[36];
    out->buildMs = ctx->buildMs;
    out->renderMs = ctx->renderMs;
    return RT_OK;
}

/* The frames-in-flight rule by itself, for host-side checks (no GPU needed).  Not part of rtcore.h. */
uint32_t rt_debug_batch_frames(uint32_t n_slots, uint32_t option, uint32_t wanted) { return rt_batch_frames_rule(n_slots, option, wanted); }

/* The tile map's division by a fixed divisor by itself (RtDivisor), for host-side checks.  Not part of rtcore.h. */
uint32_t rt_debug_divide(uint32_t n, uint32_t d, uint32_t *rem) { return rt_divisor(d).div(n, rem); }

/* Phase profile of the one-ray-per-lane walker (library built with -DTRV_PROFILE): 4 clock sums
 * (refill, node, triangle, retire) and 6 counts, summed over waves.  Not part of rtcore.h. */
int rt_debug_walker_profile(RtContext *ctx, unsigned long long path_rays[16], unsigned long long shadow_rays[16])
{
    RT_ENTER(ctx);
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    RT_HIP(ctx, hipMemcpy(path_rays, ctx->dStats + 16, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    RT_HIP(ctx, hipMemcpy(shadow_rays, ctx->dStats + 48, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_kernel_time(RtContext *ctx, const char *kernel_name, double *out_total_ms, uint64_t *out_launches)
{
    RT_ENTER(ctx);
    if (!kernel_name) return rt_fail(ctx, RT_E_INVALID, "null kernel name");
    rt_timer_resolve(ctx);
    auto it = ctx->timers.find(kernel_name);
    if (out_total_ms) *out_total_ms = it == ctx->timers.end() ? 0.0 : it->second.totalMs;
    if (out_launches) *out_launches = it == ctx->timers.end() ? 0 : it->second.launches;
    return RT_OK;
}

int rt_synchronize(RtContext *ctx)
{
    RT_ENTER_CHANGING(ctx);
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    rt_timer_resolve(ctx);
    return check_error_flag(ctx);
}

} // extern "C"
