// rt_internal.h -- device-side layouts and the context of librtcore.so.
// Internal to the library; the public surface is include/rtcore.h.
#ifndef RT_INTERNAL_H_
#define RT_INTERNAL_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rt_math.h"
#include "../../include/rt_qh.h"
#include "../../include/rtcore.h"

#define RT_LEAF_BIT 0x80000000u
#define RT_INVALID 0xFFFFFFFFu
#ifndef RT_STACK_DEPTH
#define RT_STACK_DEPTH 32   // stack levels of a ray (16 in LDS, the rest in a spill area); a ray that needs more starts over on the stackless walk
#endif

// ---- traversal layouts (HBM) ----------------------------------------------
// One internal node, 64 B, holding BOTH children's boxes so that a visit costs
// one 64-byte fetch.  A child's box is the box the reference would test when
// it pops that child (internal: bvhData[child]; leaf: triangleAABB[tri]).
struct __attribute__((aligned(16))) RtNode64 {
    float lmin[3], lmax[3];
    float rmin[3], rmax[3];
    uint32_t left, right;      // index | RT_LEAF_BIT
    uint32_t parent, pad1;     // RT_INVALID at the root (the exact walk climbs by it instead of keeping a stack)
};
static_assert(sizeof(RtNode64) == 64, "node layout");

// The node the one-ray-per-lane walker reads, 32 B: both children's boxes as half-float planes on the
// grid of rt_qh.h (each CONTAINS the exact box of RtNode64), one word per axis = low plane | high plane << 16.
struct __attribute__((aligned(16))) RtQNode {
    uint32_t lbox[3], rbox0;               // left box x, y, z; right box x
    uint32_t rbox1, rbox2, left, right;   // right box y, z; children = index | RT_LEAF_BIT
};
static_assert(sizeof(RtQNode) == 32, "quantised node layout");

// Two levels of that tree in one record, 64 B: the boxes and references of a node's GRANDCHILDREN (of a child that is
// a leaf: the child itself), up to four; an unused place holds a box no ray passes (low planes above high planes) and
// RT_INVALID.  The walk through these records visits every other level of the tree: half as many dependent fetches.
struct __attribute__((aligned(16))) RtQNode4 {
    uint32_t box[4][3];                   // x, y, z planes of each place: low | high << 16 (binary16, grid of rt_qh.h)
    uint32_t child[4];                    // index | RT_LEAF_BIT
};
static_assert(sizeof(RtQNode4) == 64, "wide node layout");
// (RT_QH_NO_BOX, the box of an unused place, is defined beside the slab test in rt_qh.h)

// Intersection record of the triangle at sorted leaf j, 64 B: v0 and the two
// edge vectors (the same subtractions IntersectTriangle_MT97 performs first)
// and the triangle's exact padded box (triangleAABB[tri]).
struct __attribute__((aligned(16))) RtLeafTri {
    float v0[3], e1[3], e2[3];
    uint32_t tri;              // original triangle index
    float bmin[3], bmax[3];
};
static_assert(sizeof(RtLeafTri) == 64, "leaf layout");

// Shading record of the triangle at sorted leaf j.
struct __attribute__((aligned(16))) RtLeafShade {
    float n0[3], n1[3], n2[3];
    uint32_t material;
    int32_t castShadow;
    uint32_t pad0;
};
static_assert(sizeof(RtLeafShade) == 48, "shade layout");

struct RtHitRec {              // 16 B: result of one BVHRayTrace
    float t, u, v;
    uint32_t leaf;             // sorted leaf index, RT_INVALID on miss
};

// Pointers a traversal / shading kernel needs, passed by value.
struct RtSceneView {
    const RtNode64 *nodes;
    const RtQNode *qnodes;
    const RtQNode4 *wnodes;    // the same tree two levels at a time, or nullptr: walk the binary nodes (RT_OPT_WIDE_NODES)
    RtQGrid grid;
    const RtLeafTri *leafTri;
    const RtLeafShade *leafShade;
    const uint32_t *leafRank;  // reference visiting order of each leaf's triangle (decides between equal distances)
    const RtMaterialData *materials;
    uint32_t nMaterials;
    uint32_t nTriangles;
    RtAABB rootBox;            // bvhData[0] (zero box when N < 2)
    const RtSphereLight *sphereLights;
    const RtAreaLight *areaLights;
    uint32_t nSphere, nArea;
    const float4 *cube;
    uint32_t cubeSize;
    uint32_t *stackSpill;      // deep stack levels of the one-ray-per-lane walker
    float pruneScale;          // a box whose entry distance exceeds (closest hit so far) x this is not entered; +inf: none is left out
    float boundQ;              // the bounded rule (rt_traverse.h, "leaving out boxes, with a bound"): RT_BOUND_C x the largest
    float boundE;              //   edge product of the scene's small triangles, and their largest edge sum; boundQ = 0: rule off
    uint32_t shareWalks;       // non-zero: at the end of a launch idle lanes take over stacked subtrees of lanes that still walk
    unsigned long long *stats; // 8 counters or nullptr
    uint32_t *errorFlag;
};

// Tile-local slot -> global pixel mapping (see RtParams in rtcore.h).
// The brute-force primitive lists (rt_primitives_upload), passed to kernels by value.
#define RT_PRIM_SHIFT 28
#define RT_PRIM_INDEX_MASK 0x0FFFFFFFu
struct RtBruteView {
    const RtPlane *planes;
    const RtSphere *spheres;
    const RtCube *cubes;
    const RtCustomMesh *meshes;
    const float4 *triPos;      // 3 per custom-mesh triangle: world-space v0, e1 = v1 - v0, e2 = v2 - v0
    const float4 *triNrm;      // 3 per triangle: world-space n0, n1, n2
    const uint32_t *triMesh;   // mesh of each triangle
    uint32_t nPlanes, nSpheres, nCubes, nMeshes, nTris;
};

// Division of 32-bit numbers by a divisor fixed when the tile map is made: quotient and remainder by one multiply-high
// and one correction instead of the thirty-odd instructions of a division by a value the compiler does not know
// (every path of every kernel that needs its pixel's coordinates made four of those).  m = ceil(2^32 / d), so
// floor(n * m / 2^32) is the quotient or one more; 2 <= d <= 2^31 (d = 1 is answered without the multiply).
struct RtDivisor {
    uint32_t d, m;
    __host__ __device__ inline uint32_t div(uint32_t n, uint32_t *rem) const
    {
        if (d == 1u) { *rem = 0; return n; }
#if defined(__HIP_DEVICE_COMPILE__)
        uint32_t q = __umulhi(n, m);
#else
        uint32_t q = (uint32_t)(((unsigned long long)n * m) >> 32);
#endif
        uint32_t r = n - q * d;
        if (r >= d) { q -= 1u; r += d; }        // one too many: the remainder wrapped
        *rem = r;
        return q;
    }
};
inline RtDivisor rt_divisor(uint32_t d)
{
    RtDivisor k;
    k.d = d ? d : 1u;
    k.m = k.d > 1u ? (uint32_t)(((1ull << 32) + k.d - 1u) / k.d) : 0u;
    return k;
}

struct RtTileMap {
    int width, height;
    int tileSize, tilesX, tilesY;
    int rank, count;
    uint32_t nLocalTiles;
    uint32_t nSlots;           // nLocalTiles * tileSize^2
    RtDivisor byTilePixels, byTilesX, byBlocksPerRow, bySlots, byCount;   // tileSize^2, tilesX, tileSize / 8, nSlots, count (finish())

    void finish()
    {
        byTilePixels = rt_divisor((uint32_t)(tileSize * tileSize));
        byTilesX = rt_divisor((uint32_t)tilesX);
        byBlocksPerRow = rt_divisor((uint32_t)tileSize >> 3);
        bySlots = rt_divisor(nSlots);
        byCount = rt_divisor((uint32_t)count);
    }
    // Which tile is the k-th of this rank.  The tiles in row-major order go to the ranks in groups of `count`, one tile
    // of a group to each rank; the order inside group k is turned by a hash of k, so that no rank's tiles form a lattice
    // over the image (columns or diagonals of tiles, which one part of a scene can load more than another).
    __host__ __device__ inline uint32_t tile_of(uint32_t k) const
    {
        uint32_t turn, h = k * 2654435761u;
        h ^= h >> 16;
        h *= 2246822507u;
        h ^= h >> 13;
        byCount.div(h >> 8, &turn);
        uint32_t r = (uint32_t)rank + turn;
        if (r >= (uint32_t)count) r -= (uint32_t)count;
        return k * (uint32_t)count + r;
    }
    // path slot (frame * nSlots + tile-local pixel slot) -> tile-local pixel slot; *frame: which frame in flight
    __host__ __device__ inline uint32_t local_slot(uint32_t pathSlot, uint32_t *frame) const
    {
        uint32_t slot;
        *frame = bySlots.div(pathSlot, &slot);
        return slot;
    }
    __host__ __device__ inline bool slot_to_pixel(uint32_t slot, uint32_t *px, uint32_t *py) const
    {
        uint32_t within, tx, bx;
        const uint32_t k = byTilePixels.div(slot, &within);
        const uint32_t t = tile_of(k);              // (beyond the last tile for some k of the last group: y >= height)
        const uint32_t ty = byTilesX.div(t, &tx);
        const uint32_t b = within >> 6, l = within & 63u;
        const uint32_t by = byBlocksPerRow.div(b, &bx);
        const uint32_t x = tx * tileSize + bx * 8 + (l & 7u);
        const uint32_t y = ty * tileSize + by * 8 + (l >> 3);
        *px = x;
        *py = y;
        return x < (uint32_t)width && y < (uint32_t)height;
    }
};

struct RtKernelTimer {
    double totalMs = 0;
    uint64_t launches = 0;
};

struct RtPendingEvent {
    std::string name;
    hipEvent_t start, stop;
};

#ifndef RT_FRAME_LANES
#define RT_FRAME_LANES 4
#endif
#define RT_COUNTER_BLOCK_WORDS (768 * RT_FRAME_LANES) // queue counts and ray cursors (dQCount), a 128-byte line each, one set per lane
#define RT_SPILL_WORDS_PER_CU (2048 * RT_STACK_DEPTH)   // room for every stack level of 2048 lanes per CU
#define RT_NONCASTER_LIST 8                             // non-casting triangles a shadow ray is tested against one by one
// A reference to a child in a two-level record (RtQNode4): bit 31 leaf, bit 30 "the subtree holds a large triangle" (the bounded
// rule never leaves such a box out), the rest the number of the node or leaf.
#define RT_REF_LARGE 0x40000000u
#define RT_REF_INDEX 0x3FFFFFFFu
// Triangles the bounded rule counts as small: edge product |e1||e2| and edge sum |e1| + |e2| at most these (scene units:
// the reference's own cut, determinant < 1e-8, is absolute too).
#define RT_SMALL_EDGE_PRODUCT 2.44140625e-4f    /* 2^-12 */
#define RT_SMALL_EDGE_SUM 0.25f
// the point o + t'd of a triangle the reference accepts lies within RT_BOUND_C x |e1||e2| x |o - v0| x |d| of the triangle
// (derivation in rt_traverse.h: 28.9 unit roundoffs over the reference's smallest determinant, 1e-8, is 172.3)
#define RT_BOUND_C 192.0f
#define RT_SPILL_AREAS (2 * RT_FRAME_LANES)                                // walker launches that may be on the chip together: per frame lane (two), the
                                                        // launches of its own stream and the shadow-ray launch that overlaps them

struct RtPostState;   // post.hip

struct RtContext {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;        // second half of the frames in flight (rt_render_frames); every other frame of the per-frame calls
    hipEvent_t evFork = nullptr, evJoin = nullptr;
    hipStream_t streamShadow = nullptr, streamChain = nullptr;   // queues masked to most / the remaining few compute units (process-wide)
    // Several frames of the per-frame call sequence in flight (RT_OPT_PIPELINE_FRAMES): frames go round RT_FRAME_LANES lanes, each
    // with its own stream, masked streams, events, frame of path state, counter block and pair of spill areas.  The fields
    // above are lane 0's; while the calls of lane k >= 1 are made they are swapped for entry k of these (RtLaneScope).
    hipStream_t laneStream[RT_FRAME_LANES] = {};      // [1] is stream2
    hipEvent_t laneFork[RT_FRAME_LANES] = {}, laneJoin[RT_FRAME_LANES] = {};
    hipStream_t laneShadow[RT_FRAME_LANES] = {}, laneChain[RT_FRAME_LANES] = {};
    hipEvent_t evAccumulated[RT_FRAME_LANES] = {};    // after a lane's rt_accumulate: the next frame's fold waits for it
    hipEvent_t evLaneIdle[RT_FRAME_LANES] = {};
    bool accumulatedPending[RT_FRAME_LANES] = {};
    bool laneBusy[RT_FRAME_LANES] = {};               // the lane's stream may hold work that lane 0's stream has not waited for
    int optChainBounces = 3;                          // RT_OPT_CHAIN_BOUNCES: first bounce of the one-launch chain, 0 = off
    int optPipelineFrames = RT_FRAME_LANES;           // lanes in use (1: the per-frame calls go down one stream)
    uint32_t frameLane = 0, nextLane = 0; // the lane of the frame rt_dispatch last started, and of the next one
    uint32_t lastFoldLane = 0;            // the lane whose rt_accumulate came last
    uint32_t scopeLane = 0;               // the frame lane whose fields are swapped in right now (RtLaneScope)
    hipStream_t streamMain = nullptr;     // the caller's stream while ctx->stream is the chain stream
    uint32_t overlapCUs = 0;              // the split of the masked streams in use (0: none yet)
    std::vector<uint32_t> overlapSets;    // the splits whose process-wide sets of masked streams this context holds a share of
    bool overlapFailed = false;           // the runtime refused masked streams: the plain order is used
    std::string err;

    // inputs
    RtMeshVertex *dVerts = nullptr;
    uint32_t nVerts = 0;
    uint32_t *dIdx = nullptr;
    uint32_t *dMatIdx = nullptr;
    int32_t *dShadow = nullptr;
    uint32_t N = 0;
    RtMaterialData *dMats = nullptr;
    uint32_t nMats = 0;
    RtSphereLight *dSphere = nullptr;
    RtAreaLight *dArea = nullptr;
    RtDiscLight *dDisc = nullptr;
    uint32_t nSphere = 0, nArea = 0, nDisc = 0;
    float4 *dCube = nullptr;
    uint32_t cubeSize = 0;
    RtPlane *dPlanes = nullptr;
    RtSphere *dSpheres = nullptr;
    RtCube *dCubes = nullptr;
    RtCustomMesh *dMeshes = nullptr;
    float4 *dTriPos = nullptr, *dTriNrm = nullptr;
    uint32_t *dTriMesh = nullptr;
    RtBruteView brute;
    bool bruteSet = false;

    // build products
    RtAABB *dTriAABB = nullptr;
    uint32_t *dMortonUnsorted = nullptr, *dMortonSortedRaw = nullptr, *dMorton = nullptr;
    uint32_t *dSortedIdx = nullptr;
    uint32_t *dLeft = nullptr, *dRight = nullptr, *dParent = nullptr, *dLeafParent = nullptr;
    RtAABB *dBvhData = nullptr;
    RtNode64 *dNodes = nullptr;
    RtQNode *dQNodes = nullptr;
    RtQNode4 *dWNodes = nullptr;          // two levels of the traversal tree per record (k_emit_wide_nodes)
    bool optWideNodes = true;             // RT_OPT_WIDE_NODES
    RtQGrid grid;
    RtLeafTri *dLeafTri = nullptr;
    RtLeafShade *dLeafShade = nullptr;
    uint32_t *dLeafRank = nullptr;        // per traversal-tree leaf: position of its triangle in the reference's visiting order
    RtAABB rootBox;
    bool built = false;

    // render state
    RtParams params;
    bool paramsSet = false;
    RtTileMap tile;
    uint32_t allocSlots = 0;
    float4 *dRt = nullptr, *dConv = nullptr, *dShadowMap = nullptr;
    float4 *dPe = nullptr;                // path energy, by path slot
    float4 *dRay[2] = {nullptr, nullptr}; // the ray queues of a bounce and the next: 2 x float4 per entry, in queue order (render.hip)
    uint2 *dRng = nullptr;
    RtHitRec *dHit = nullptr;
    uint32_t *dQCount = nullptr;          // 32-word counter block, owned by the context
    uint32_t *dSmQueue = nullptr;         // ShadowMap pass: shaded path slots
    // A build with as many triangles as the last one (a scene whose vertices move) reuses that build's buffers and, while
    // they are no more than RT_BUILD_SCRATCH_KEEP bytes, its scratch: the 60-odd hipMalloc / hipFree pairs of a build cost
    // three times what its kernels take (the 496 k-triangle car: 5.9 ms per rt_bvh_build with them, 1.5 ms of kernels).
    uint32_t buildBuffersN = 0;           // triangle count the build's result buffers were made for (0: none)
    std::vector<std::pair<void *, size_t>> buildScratch;   // (buffer, bytes) in the order a build asks for them
    hipEvent_t buildEvents[2] = {nullptr, nullptr};
    uint32_t *dQPos = nullptr;            // per path: its place in the camera rays' queue (RT_INVALID: a padding slot)
    float *dSeeds = nullptr;              // per-frame seeds of the batch in flight
    bool optPacketShadow = false, optPacketPrimary = false, optPacketTrace = false;
    bool optMaterialSort = true;
    int optTwoStreams = -1;               // RT_OPT_TWO_STREAMS: -1 = chosen per batch, 0 = one stream, 1 = two plain streams, 2 .. 4 = that many frame lanes (each with its overlapped shadow rays)
    float4 *dNonCasters = nullptr;        // leafTri-layout records of the non-casting triangles while they are at most RT_NONCASTER_LIST
    uint32_t nNonCasters = 0;
    RtAABB nonCasterBox = {{0, 0, 0}, 0, {0, 0, 0}, 0};   // round the triangles whose castShadow flag is not set (rt_scene_upload)
    RtAABB casterBox = {{0, 0, 0}, 0, {0, 0, 0}, 0};      // round the own boxes of the triangles whose castShadow flag is set (rt_scene_upload)
    bool optCasterBox = true;             // shadow rays that cannot meet a shadow caster are not walked
    bool optFirstHitShadow = true;        // shadow rays that cannot meet a non-casting triangle stop at their first hit
    bool optOverlapShadow = true;         // walk the shadow rays while the later bounces run (render.hip, overlap_fork)
    int optPruneByDistance = 1;           // RT_OPT_PRUNE_BY_DISTANCE: 0 no box is left out, 1 what the bounded rule allows (exact), 2 what begins
                                          // 1.5 % beyond the closest hit
    float smallEdgeProduct = 0.0f, smallEdgeSum = 0.0f;   // largest |e1||e2| and |e1| + |e2| among the scene's small triangles (0: none)
    bool optShareLongWalks = true;        // idle lanes take over parts of the walks a launch ends with (RT_OPT_SHARE_LONG_WALKS)
    bool optEagerShadowMap = true;        // rt_dispatch starts the ShadowMap pass of its frame (render.hip, rt_dispatch_impl)
    int eagerShadow = 0;                  // RT_EAGER_*: what rt_dispatch left for rt_dispatch_shadowmap
    uint32_t optOverlapAfter = 0;         // ... from the extend launch of this bounce on
    uint32_t optOverlapCUs = 0;           // compute units the bounce launches keep meanwhile (0: 64, or 96 for a small batch; render.hip)
    uint32_t optBatchFrames = 0;          // 0 = by path count (rt_batch_frames)
    uint32_t batchCap = 0;                // frames the path-state buffers can hold
    uint32_t lastFrame = 0;               // frame of the batch that RT_IMAGE_RT / RT_IMAGE_SHADOWMAP show
    float4 *dSmRec = nullptr;             // ShadowMap pass: 2 x float4 per slot
    float4 *dSmRecQ = nullptr;            // the same records in the order of the queue of shaded pixels (what the shadow rays read)
    uint8_t *dSmFlags = nullptr;          // ShadowMap pass: 16 results per slot
    int numCU = 256;
    uint32_t sampleCount = 0;

    // stats / timing
    uint32_t *dStackSpill = nullptr;     // RT_SPILL_AREAS x numCU x RT_SPILL_WORDS_PER_CU words
    uint32_t spillArea = 0;              // the area rt_scene_view hands out (1 while the second stream's launches are made)
    unsigned long long *dStats = nullptr; // 64 x u64: counters [0..4] / [32..36] (shadow rays), walker profile [16..25] / [48..57]
    uint32_t *dErrorFlag = nullptr;
    bool countersOn = false;
    bool timingOn = false;
    double buildMs = 0, renderMs = 0;
    std::map<std::string, RtKernelTimer> timers;
    std::vector<RtPendingEvent> pending;
    std::vector<hipEvent_t> eventPool;

    // rt_readback_async: a full frame in device memory that the tiles are expanded into, and the stream that copies it out
    float4 *dReadbackFull = nullptr;
    size_t readbackPixels = 0;
    hipStream_t streamCopy = nullptr;
    hipEvent_t evExpanded = nullptr, evCopied[2] = {nullptr, nullptr};   // two staging frames in dReadbackFull, used in turn:
    bool copyPending[2] = {false, false};                                 // a frame is expanded while the one before is copied out
    uint32_t readbackSlot = 0;

    RtPostState *post = nullptr;          // images and scratch of the post-process chain, made on first use
};

// ---- helpers shared by the translation units -------------------------------
int rt_fail(RtContext *ctx, int code, const char *fmt, ...);
#define RT_HIP(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e__ = (call);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return rt_fail((ctx), e__ == hipErrorOutOfMemory ? RT_E_NOMEM : RT_E_DEVICE,           \
                           "%s failed: %s", #call, hipGetErrorString(e__));                        \
    } while (0)

void rt_timer_begin(RtContext *ctx, const char *name);
void rt_timer_end(RtContext *ctx);
void rt_timer_resolve(RtContext *ctx);

// api.hip
float4 *rt_image_ptr(RtContext *ctx, int which);   // tile-local device image behind RT_IMAGE_RT / _CONVERGED / _SHADOWMAP

// brute.hip
void rt_free_brute(RtContext *ctx);
void rt_release_overlap_streams(RtContext *ctx);   // render.hip

// post.hip
void rt_free_post(RtContext *ctx);
int rt_post_readback_impl(RtContext *ctx, int which, float *dst_rgba);

// build.hip
int rt_build_impl(RtContext *ctx);
int rt_readback_bvh_impl(RtContext *ctx, uint32_t *morton, uint32_t *sortedIdx, RtAABB *triAABB, RtTriangle *tris,
                         RtInternalNode *inodes, RtLeafNode *leaves, RtAABB *bvhData);
void rt_free_build(RtContext *ctx);          // the build's results, their buffers and the scratch kept for the next build
#define RT_BUILD_SCRATCH_KEEP ((size_t)1 << 30)

// render.hip
RtSceneView rt_scene_view(RtContext *ctx);
int rt_alloc_render(RtContext *ctx);
// What a call borrows and gives back when it ends, however it ends (every error path returns early): scratch
// buffers and events.  The stream is drained first, because launches in flight may still use the buffers.
struct RtArena {
    hipStream_t stream;
    std::vector<void *> buffers;
    std::vector<hipEvent_t> events;
    explicit RtArena(hipStream_t s) : stream(s) {}
    RtArena(const RtArena &) = delete;
    RtArena &operator=(const RtArena &) = delete;
    template <typename T>
    hipError_t alloc(T **p, size_t count)
    {
        const hipError_t e = hipMalloc((void **)p, (count ? count : 1) * sizeof(T));
        if (e == hipSuccess) buffers.push_back(*p);
        return e;
    }
    hipError_t event(hipEvent_t *e)
    {
        const hipError_t rc = hipEventCreate(e);
        if (rc == hipSuccess) events.push_back(*e);
        return rc;
    }
    ~RtArena()
    {
        if (!buffers.empty()) (void)hipStreamSynchronize(stream);
        for (void *q : buffers) (void)hipFree(q);
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
    }
};

enum { RT_EAGER_NONE = 0, RT_EAGER_SETUP = 1, RT_EAGER_FORKED = 2 };   // nothing / pass prepared / its rays already being walked
int rt_eager_cancel(RtContext *ctx);
int rt_join_lanes(RtContext *ctx);    // lane 0's stream waits for what lane 1's holds: after it, draining ctx->stream drains everything
uint32_t rt_batch_frames_rule(uint32_t nSlots, uint32_t option, uint32_t wanted);
void rt_free_render(RtContext *ctx);
int rt_dispatch_impl(RtContext *ctx);
int rt_shadowmap_impl(RtContext *ctx);
int rt_accumulate_impl(RtContext *ctx, uint32_t samplePerPixel, int useShadowMap);
int rt_render_frames_impl(RtContext *ctx, const float *seeds, uint32_t nFrames);
int rt_trace_device_impl(RtContext *ctx, const float4 *o, const float4 *d, uint32_t n, RtHitRec *hits);
int rt_static_scratch_init(RtContext *ctx);   // once per device, from rt_create
int rt_trace_device_static_impl(RtContext *ctx, const float4 *o, const float4 *d, uint32_t n, RtHitRec *hits);
int rt_readback_async_impl(RtContext *ctx, const float4 *src, float *pinnedDst);
int rt_expand_local_impl(RtContext *ctx, const float4 *local, float4 *full);
int rt_expand_local_on(RtContext *ctx, hipStream_t stream, const float4 *local, float4 *full);   // the same scatter on another stream (of any device that sees both buffers)
int rt_trace_primitives_impl(RtContext *ctx, const float4 *o, const float4 *d, uint32_t n, float *t, uint32_t *prim, float *normal3);
int rt_pack_rays_impl(RtContext *ctx, const float *o3, const float *d3, uint32_t n, float4 *o4, float4 *d4);
int rt_debug_eval_impl(RtContext *ctx, int which, const float *dx, const float *dy, uint32_t n, float *dout);
int rt_hits_to_host_impl(RtContext *ctx, const RtHitRec *hits, uint32_t n, float *t, uint32_t *tri, float *u, float *v);

#endif
