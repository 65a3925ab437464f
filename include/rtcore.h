/* rtcore.h -- C ABI of librtcore.so, the MI355X (gfx950) path-tracing core.
 *
 * The reference project drives its GPU path only through Unity's
 * ComputeShader / ComputeBuffer / Graphics.Blit objects; it has no native
 * plugin seam.  Each entry point below replaces one group of those calls and
 * cites the reference call site it stands in for.  A C# host binds them with
 * [DllImport("rtcore")]; every struct here is blittable and has the byte
 * layout of the reference's own GPU structs.
 *
 * Conventions
 *   - every call returns 0 (RT_OK) or a negative RT_E_* code; nothing throws
 *     or aborts across the boundary; rt_last_error() gives the message.
 *   - the caller owns every host array; the library copies during the call.
 *   - the library owns all device memory behind the opaque RtContext.
 *   - one RtContext = one GPU = one caller thread at a time.
 *   - 4x4 matrices are 16 floats in Unity Matrix4x4 memory order
 *     (column-major: element(row r, col c) = m[c*4 + r]).
 *   - image row 0 is the bottom of the view (RayTracing.compute:58).
 */
#ifndef RTCORE_H_
#define RTCORE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------ error codes */
#define RT_OK 0
#define RT_E_INVALID   (-1)  /* bad argument / call order */
#define RT_E_DEVICE    (-2)  /* HIP runtime error (no GPU, launch failure, ...) */
#define RT_E_NOMEM     (-3)
#define RT_E_STATE     (-4)  /* e.g. dispatch before rt_bvh_build */
#define RT_E_OVERFLOW  (-5)  /* the packet walker (RT_OPT_PACKET_*) ran out of stack; the default walker cannot: a ray whose
                                 stack is full starts over on a walk that needs none */

/* ------------------------------------------------- wire structs (blittable) */
/* Assets/Shaders/BVH/Constants.cginc:11-17, Assets/Scripts/BVH/SceneDataTypes.cs:6-21 */
typedef struct RtAABB {
    float min[3]; float _dummy0;
    float max[3]; float _dummy1;
} RtAABB;                                   /* 32 B */

/* Constants.cginc:20-59, SceneDataTypes.cs:23-86 */
typedef struct RtTriangle {
    float point0[3]; float _dummy0;
    float point1[3]; float _dummy1;
    float point2[3]; float _dummy2;
    float uv0[2]; float uv1[2]; float uv2[2]; float _dummy3[2];
    float normal0[3]; float _dummy4;
    float normal1[3]; float _dummy5;
    float normal2[3]; float _dummy6;
    float tangent0[3]; float _dummy7;
    float tangent1[3]; float _dummy8;
    float tangent2[3]; float _dummy9;
    int32_t materialIndex; int32_t castShadow; int32_t receiveShadow; float _dummy10;
} RtTriangle;                               /* 192 B */

/* Constants.cginc:62-79, SceneDataTypes.cs:88-120 */
#define RT_INTERNAL_NODE 0u
#define RT_LEAF_NODE 1u
typedef struct RtInternalNode {
    uint32_t leftNode, leftNodeType, rightNode, rightNodeType, parent, index;
} RtInternalNode;                           /* 24 B */
typedef struct RtLeafNode { uint32_t parent, index; } RtLeafNode; /* 8 B */

/* Assets/Shaders/BVH/BVHRayTracing.cginc:6-17, Assets/Scripts/BVH/BVH.cs:24-37 */
typedef struct RtMaterialData {
    float albedo[3];
    float metallic;
    float smoothness;
    float transparent;
    float emissionColor[3];
    uint32_t materialType;  /* 0 opaque, 1 transparent, 2 emission, 3 clear coat, 4 matte mask */
    float ior;
    float clearCoatColor[3];
} RtMaterialData;                           /* 56 B */

/* BVH.cs:15-21; MeshData.compute:27-38 (11 floats per vertex) */
typedef struct RtMeshVertex {
    float position[3]; float normal[3]; float tangent[3]; float uv[2];
} RtMeshVertex;                             /* 44 B */

/* ImportanceSampling.cginc:8-39, LightImportanceSampling.cs:16-55 */
typedef struct RtSphereLight { float position[3]; float radius; } RtSphereLight;      /* 16 B */
typedef struct RtAreaLight { float position[3]; float normal[3]; float up[3]; float size[2]; } RtAreaLight; /* 44 B */
typedef struct RtDiscLight { float position[3]; float normal[3]; float radius; } RtDiscLight; /* 28 B */

/* Brute-force primitives of chapters 2-4 (useBVH = false): the GPU structs of
 * BruteForceRayTracing.cginc:5-65 = MeshCollector.cs:46-156, field for field. */
typedef struct RtPlane {
    float normal[3], position[3], size[3];
    float albedo[3]; float metallic, smoothness, transparent; float emissionColor[3];
} RtPlane;                                  /* 72 B */
typedef struct RtSphere {
    float center[3]; float radius;
    float albedo[3]; float metallic, smoothness, transparent; float emissionColor[3];
} RtSphere;                                 /* 52 B */
typedef struct RtCube {
    float min[3], max[3];
    float albedo[3]; float metallic, smoothness, transparent; float emissionColor[3];
} RtCube;                                   /* 60 B */
typedef struct RtCustomMesh {
    float localToWorldMatrix[16];           /* Unity memory order */
    int32_t indicesOffset, indicesCount;
    float albedo[3]; float metallic, smoothness, transparent; float emissionColor[3];
} RtCustomMesh;                             /* 108 B */

/* ------------------------------------------------------------------ flags */
/* RayTracing.cs:40-48 (SamplingType) -> bits 0..2 */
#define RT_SAMPLING_UNIFORM   0u
#define RT_SAMPLING_COSINE    1u
#define RT_SAMPLING_LIGHT     2u
#define RT_SAMPLING_BSDF      3u
#define RT_SAMPLING_MIS       4u
#define RT_FLAG_SAMPLING_MASK 0x7u
/* DepthOfField.cs:22-36: DEPTH_OF_FIELD keyword.  In the reference only the
 * ShadowMap kernel honours it (RayTracing.compute:60-65 vs 105-110). */
#define RT_FLAG_DEPTH_OF_FIELD   0x10u
/* Extension, off by default: also use the DoF camera for CSMain primary rays. */
#define RT_FLAG_DOF_PRIMARY      0x20u
/* RayTracing.cs:178-182: a matte-mask object exists -> ShadowMap pass runs. */
#define RT_FLAG_MATTE_MASK       0x40u
/* RayTracing.cs:84-93: useBVH = false -> cs.DisableKeyword("BVH"); CSMain then calls
 * BruteForceRayTrace over the rt_primitives_upload lists (RayTracing.compute:73-77).
 * The ShadowMap kernel always walks the BVH (RayTracing.compute:112). */
#define RT_FLAG_BRUTE_FORCE      0x80u

/* POD mirror of every uniform the reference sets on RayTracing.compute:
 * RayTracing.cs:194-204,223-230,256-260; DepthOfField.cs:22-36. */
typedef struct RtParams {
    int32_t width, height;
    float camera2World[16];
    float cameraInverseProjection[16];
    float seed;                 /* per-frame: RayTracing.cs:230 */
    float lightParameter[4];    /* xyz = mainLight forward, w = lightRadius */
    float lightColor[4];        /* rgb = light colour, a = skyboxIntensityMulitiply */
    float shadowParameter[4];   /* rgb = shadow colour, a = shadowIntensity */
    float skyboxRotation;
    float skyboxExposure;
    float lensRadius;
    float focalLength;
    uint32_t flags;
    /* Image-tile sharding (no counterpart in the reference; single GPU there).
     * The frame is cut into tileSize x tileSize tiles in row-major order; every
     * group of tileCount consecutive tiles gives one tile to each rank, in an
     * order turned from group to group by a hash of the group's number, so that
     * no rank's tiles form columns or diagonals of the image (which tile is a
     * rank's k-th: rt_local_pixel_map tells, per pixel slot).  Small tiles even
     * the ranks' loads out: the eight shares of the 1080p bench image differ by
     * a fifth of their time in 64 x 64 tiles, by a fiftieth in 8 x 8.  Pixel
     * coordinates given to the RNG stay global, so the union of all ranks'
     * tiles is bit-identical to a single-GPU frame.  tileCount <= 1 = whole
     * frame.  tileSize: a multiple of 8, at most 1024; <= 0 = 64. */
    int32_t tileRank, tileCount, tileSize;
} RtParams;

typedef struct RtContext RtContext;

/* Work counters of the BVH traversal since the last rt_reset_stats. */
typedef struct RtStats {
    uint64_t rays;            /* BVHRayTrace calls ANSWERED: every call the reference would make for the same frames,
                                 including the 16 soft-shadow rays of ShadowMap pixels that the library settles without a
                                 walk (no ray of the pixel can meet a shadow caster); see raysWalked */
    uint64_t internalVisits;  /* internal-node boxes tested (n_I)  */
    uint64_t leafVisits;      /* leaf triangle boxes tested (n_L)  */
    uint64_t triTests;        /* triangles that passed their box (n_T) */
    uint64_t hits;            /* rays that ended with a hit (h) */
    uint64_t paths;           /* camera paths generated */
    double buildMs;           /* device time of the last rt_bvh_build */
    double renderMs;          /* device time of the last rt_render_frames, or of the walk of the last rt_trace_rays* call */
    /* the part of rays / internalVisits / leafVisits / triTests / hits due to the 16 soft-shadow
     * rays per pixel of the ShadowMap pass (Shadow.cginc:95-111) */
    uint64_t shadowRays, shadowInternalVisits, shadowLeafVisits, shadowTriTests, shadowHits;
    /* the rays that a traversal kernel actually took through the tree (rays - raysWalked were answered without a walk);
     * a ray that the first-hit rule ends early counts as walked */
    uint64_t raysWalked, shadowRaysWalked;
} RtStats;

/* ---------------------------------------------------------------- lifetime */
/* Replaces ComputeShader/ComputeBuffer object creation in RayTracing.Start
 * (RayTracing.cs:68-97).  device_id is the HIP device ordinal. */
int rt_create(int device_id, RtContext **out_ctx);
/* RayTracing.cs:319-334, BVH.cs:554-559, MeshBufferContainer.cs:115-129. Idempotent on NULL. */
int rt_destroy(RtContext *ctx);
const char *rt_last_error(RtContext *ctx);

/* ------------------------------------------------------------ scene upload */
/* The merged world-space mesh that BVH.InitMesh builds (BVH.cs:201-318) and
 * MeshBufferContainer uploads (MeshBufferContainer.cs:78-89): vertex buffer,
 * uint32 index buffer, per-triangle material index, per-triangle (cast,receive). */
int rt_scene_upload(RtContext *ctx, const RtMeshVertex *vertices, uint32_t n_vertices,
                    const uint32_t *indices, uint32_t n_indices,
                    const uint32_t *material_index, const int32_t *shadow_index_xy,
                    uint32_t n_triangles);
/* BVH.InitMaterialData / UpdateMaterialData (BVH.cs:320-325,334-352).  Also marks dirty. */
int rt_materials_upload(RtContext *ctx, const RtMaterialData *materials, uint32_t n);
/* LightImportanceSampling.Init (LightImportanceSampling.cs:57-107): a count of
 * zero disables the corresponding SPHERE_/AREA_/DISC_LIGHT keyword. */
int rt_lights_upload(RtContext *ctx, const RtSphereLight *sphere, uint32_t n_sphere,
                     const RtAreaLight *area, uint32_t n_area,
                     const RtDiscLight *disc, uint32_t n_disc);
/* cs.SetTexture(kernel,"skyboxCube",...) (RayTracing.cs:198-205).  faces = 6
 * images of size x size RGBA float32 in D3D face order +X,-X,+Y,-Y,+Z,-Z,
 * each row-major with v downwards.  size = 0 unbinds (samples read 0). */
int rt_skybox_upload(RtContext *ctx, const float *faces_rgba, uint32_t size);

/* MeshCollector.Init (MeshCollector.cs:170-200, 271-307): the four primitive lists in
 * the order the host sorted them (far to near, :279-290) and the merged vertex /
 * normal / index buffers of the custom meshes (3 floats per vertex; indices already
 * offset per mesh, :138-152).  The reference uploads one zeroed element for an empty
 * list (:299-300); the caller does the same if it wants that behaviour, a count of 0
 * simply skips the list.  indicesCount must be a multiple of 3 and every index < n_vertices. */
int rt_primitives_upload(RtContext *ctx, const RtPlane *planes, uint32_t n_planes, const RtSphere *spheres, uint32_t n_spheres,
                         const RtCube *cubes, uint32_t n_cubes, const RtCustomMesh *meshes, uint32_t n_meshes,
                         const float *vertices3, const float *normals3, uint32_t n_vertices,
                         const int32_t *indices, uint32_t n_indices);
/* BruteForceRayTrace (BruteForceRayTracing.cginc:246-279) for a batch of caller rays.
 * Outputs (any may be NULL): hit.distance (inf when Shade would see a miss), the
 * primitive that wrote the hit last as kind << 28 | index (kind 0 plane, 1 sphere,
 * 2 cube, 3 custom-mesh triangle counted over all meshes in list order; 0xFFFFFFFF
 * on a miss) and hit.normal (n x 3). */
#define RT_PRIM_PLANE 0u
#define RT_PRIM_SPHERE 1u
#define RT_PRIM_CUBE 2u
#define RT_PRIM_TRIANGLE 3u
int rt_trace_primitives(RtContext *ctx, const float *origins, const float *directions, uint32_t n,
                        float *out_t, uint32_t *out_primitive, float *out_normal);

/* --------------------------------------------------------------- BVH build */
/* MeshData.Calculate + ComputeBufferSorter.Sort + DistributeMortonCode +
 * BVHConstructor.ConstructTree/ConstructBVH (BVH.cs:128-170).
 * A scene whose vertices move is uploaded and built again (BVH.cs rebuilds on a changed transform): while the triangle
 * count stays what it was, rt_scene_upload and rt_bvh_build reuse their device buffers and the build its scratch (kept
 * between builds while it is 1 GiB at most), so such a build costs its kernels and nothing else. */
int rt_bvh_build(RtContext *ctx);

/* Debug readbacks in the reference wire structs, what BVH.OnDrawGizmos reads
 * (BVH.cs:361-367,501-505).  Any pointer may be NULL.  Array lengths:
 * morton/sorted_index/tri_aabb/triangles/leaf_nodes: N; internal_nodes/bvh_data: N-1. */
int rt_readback_bvh(RtContext *ctx, uint32_t *morton_codes, uint32_t *sorted_triangle_index,
                    RtAABB *triangle_aabb, RtTriangle *triangles,
                    RtInternalNode *internal_nodes, RtLeafNode *leaf_nodes, RtAABB *bvh_data);
/* Intermediate build products for the sorter's own checks
 * (ComputeBufferSorter.cs:155-182): Morton codes before sorting and the sorted
 * codes before DistributeMortonCode.  Either may be NULL. */
int rt_readback_sort_stages(RtContext *ctx, uint32_t *morton_unsorted, uint32_t *morton_sorted_raw);

/* The tree the rays actually walk.  No reference counterpart: the reference walks the tree
 * above, whose Morton cells (0.24 units, fixed world box) are far coarser than the triangles
 * of a typical scene.  The library builds a second tree over the same triangle boxes with
 * the same steps from finer keys (include/rt_finekey.h); results are the reference's, bit
 * for bit, because (a) which triangles a ray tests depends only on the triangles' own
 * boxes and (b) equally distant triangles are ranked by the reference's visiting order.
 * For checks of that build: internal nodes as 8 words each (six half-float (binary16) plane
 * numbers per child box in 3 words: left box, right box, then left and right child = index, or
 * leaf number | 0x80000000), N-1 of them; per leaf the original triangle index and its
 * place in the reference's visiting order (N each).  Any pointer may be NULL. */
int rt_readback_walk_tree(RtContext *ctx, uint32_t *nodes8, uint32_t *leaf_triangle, uint32_t *leaf_rank);

/* ------------------------------------------------------------------ render */
/* Every cs.Set* of RayTracing.InitShader/InitLight/InitSampling/UpdateParameters. */
int rt_set_params(RtContext *ctx, const RtParams *params);
/* cs.Dispatch(kernelHandle, ...) = CSMain, one sample per pixel into `rt`
 * (RayTracing.cs:128-132). */
int rt_dispatch(RtContext *ctx);
/* cs.Dispatch(kernelHandleShadowMap, ...) (RayTracing.cs:134-135). */
int rt_dispatch_shadowmap(RtContext *ctx);
/* Graphics.Blit(rt, convergedRT, addMaterial, pass) with _SamplePrePixel
 * (RayTracing.cs:138-150, AddShader.shader:12,47-79). */
int rt_accumulate(RtContext *ctx, uint32_t sample_per_pixel, int use_shadow_map);
/* n_frames x { seed := seeds[k]; CSMain; [ShadowMap]; accumulate; samplePrePixel++ }
 * without returning to the host in between = n_frames calls of CustomRender
 * (RayTracing.cs:118-164).  Starts from the context's own sample counter. */
int rt_render_frames(RtContext *ctx, const float *seeds, uint32_t n_frames);
/* RayTracing.SetDirty (RayTracing.cs:314-317). */
int rt_set_dirty(RtContext *ctx);
uint32_t rt_sample_count(RtContext *ctx);

#define RT_IMAGE_RT        0   /* last 1-spp frame  (RayTracing.rt) */
#define RT_IMAGE_CONVERGED 1   /* running mean      (RayTracing.convergedRT) */
#define RT_IMAGE_SHADOWMAP 2   /* RayTracing.shadowMap */
#define RT_IMAGE_POSTPROCESS 3 /* RayTracing.postProcessRT: what the last rt_post_* call drew */
#define RT_IMAGE_USER      4   /* a frame handed in with rt_image_upload */
/* Full-frame float4 readback (width*height*4 floats, row 0 = bottom).  Pixels
 * of tiles owned by other ranks read 0.  RT_IMAGE_POSTPROCESS / RT_IMAGE_USER
 * have the size of the frame they were made from (rt_image_size). */
int rt_readback(RtContext *ctx, int which, float *dst_rgba);
/* The same image written INTO a full frame the caller already holds: only the
 * pixels of this context's tiles are stored, every other pixel keeps its
 * value.  A host that drives several GPUs from one process (one context per
 * device, RtParams.tileRank = 0..tileCount-1) calls it once per context on one
 * buffer and ends up with the frame RayTracing.cs:365-370 blits; the contexts
 * write disjoint pixels, so the calls may run from different threads.
 * RT_IMAGE_RT / RT_IMAGE_CONVERGED / RT_IMAGE_SHADOWMAP only. */
int rt_readback_merge(RtContext *ctx, int which, float *frame_rgba);
/* rt_readback in the background, for a host that keeps rendering: the image is staged on the device when the call is
 * made (the library may render into it again as soon as the call returns) and copied to `pinned_dst_rgba` -- page-locked
 * host memory of the caller, width*height*4 floats, e.g. a pinned GCHandle / hipHostMalloc -- by a copy stream while the
 * next frames are traced.  The destination is complete after rt_readback_wait (also the place the traversal error flag is
 * reported); a second rt_readback_async waits for the first one's copy by itself.  The reference reads its render
 * textures back with a blocking ReadPixels (SaveTexture.cs:59-70); RT_IMAGE_RT / RT_IMAGE_CONVERGED / RT_IMAGE_SHADOWMAP only. */
int rt_readback_async(RtContext *ctx, int which, float *pinned_dst_rgba);
int rt_readback_wait(RtContext *ctx);
/* rt_readback_merge without the host: this context's tiles are written INTO a full frame (width*height float4, row 0 =
 * bottom) that lives in DEVICE memory of `frame_device`, this context's own device or another one of the process; every
 * other pixel keeps its value, so the contexts that split a frame (RayTracing.cs:365-370 blits ONE frame) merge into one
 * device buffer, one call each, without the frame crossing to the host.  Across devices the tiles are written through the
 * peer mapping where the devices allow it (over xGMI), otherwise copied as they are and placed by the frame's device.
 * The across-device paths have not run on hardware (no multi-GPU box in reach of this build). */
int rt_merge_local_device(RtContext *ctx, int which, void *device_frame_rgba, int frame_device);
/* Same image, but only this rank's tiles in tile-local order, copied to a
 * DEVICE pointer (e.g. a torch tensor) of rt_local_pixel_count()*4 floats, for
 * gathering over RCCL without a host hop. */
int rt_readback_local_device(RtContext *ctx, int which, void *device_dst);
/* Resume support: load a running mean + its sample count (checkpoint of
 * RayTracing.convergedRT + samplePrePixel, RayTracing.cs:31-33). */
int rt_upload_converged(RtContext *ctx, const float *src_rgba, uint32_t sample_per_pixel);
uint32_t rt_local_pixel_count(RtContext *ctx);
/* Global pixel index (y*width+x, or 0xFFFFFFFF for padding slots) of every
 * tile-local slot, so a gatherer can place each rank's pixels. */
int rt_local_pixel_map(RtContext *ctx, uint32_t *dst);

/* First-hit queries = BVHRayTrace (BVHRayTracing.cginc:125-133) for a batch of
 * caller rays: origins/directions are n x 3 floats; outputs (any may be NULL)
 * are distance (inf on miss), original triangle index (0xFFFFFFFF on miss) and
 * barycentric u,v.  Used for parity checks and the traversal stress test. */
int rt_trace_rays(RtContext *ctx, const float *origins, const float *directions, uint32_t n,
                  float *out_t, uint32_t *out_triangle, float *out_u, float *out_v);
/* Same query on rays already resident in device memory (n x float4 each,
 * xyz used); hit records (t,u,v,tri-as-bits) are written to device_hits. */
int rt_trace_rays_device(RtContext *ctx, const void *device_origins4, const void *device_dirs4,
                         uint32_t n, void *device_hits4);

/* rt_trace_rays_device without any allocation.  No counterpart in the reference: it is for
 * callers that cannot allocate device memory -- tight serving loops, harnesses that forbid
 * hipMalloc.  (It is not a call to put under HIP graph capture as it stands: it works on the
 * context's own stream and waits for that stream to report a stack overflow.)  The walker's scratch (ray cursor, overflow flag,
 * the stack levels a lane spills) lives in a static device array of 20 MB that every GPU the
 * library runs on holds until the process ends, instead of buffers of the context; the call
 * makes no hipMalloc / hipFree, creates no events and keeps no timing.  Same inputs, outputs
 * and results as rt_trace_rays_device, for at most RT_TRACE_STATIC_MAX_RAYS rays per call
 * (more: RT_E_INVALID).  All calls on one GPU share that GPU's array: calls that overlap on a
 * GPU (two contexts, two threads) race; one after the other is fine. */
#define RT_TRACE_STATIC_MAX_RAYS (1u << 24)
int rt_trace_rays_device_static(RtContext *ctx, const void *device_origins4, const void *device_dirs4,
                                uint32_t n, void *device_hits4);

/* Numeric self-check: evaluates one of the shared numeric rules of rt_math.h on
 * the device for n host inputs (which: 0 sin, 1 cos, 2 acos, 3 pow(x,y),
 * 4 sRGB->linear, 5 linear->sRGB, 6 float->uint as bits, 7 the Rand() stream of
 * pixel (x[0],x[1]) with seed y[0], Header.cginc:79-92; 8 DiscLightImportanceSampling,
 * ImportanceSampling.cginc:147-160 -- a function the reference defines but never calls --
 * for x = pixel(2), hit.position(3), hit.normal(3), hit.albedo(3), light.position(3),
 * light.radius and y[0] = seed, giving n <= 10 values: ray.origin, ray.direction, func, pdf).
 * No reference counterpart. */
int rt_debug_eval(RtContext *ctx, int which, const float *x, const float *y, uint32_t n, float *out);

/* ------------------------------------------------------ post-process chain */
/* The stages of PostProcessStack (PostProcessStack.cs:43-72), each one
 * `RenderImage(source, destination)` = Graphics.Blit(source, destination, material).
 * `src` is an RT_IMAGE_* selector; the destination is always RT_IMAGE_POSTPROCESS
 * (RayTracing.cs:156 passes convergedRT and postProcessRT).  Note that the
 * reference's stack hands the SAME source to every enabled stage
 * (PostProcessStack.cs:29-35, one multicast delegate), so what reaches the
 * screen is the last enabled stage applied to convergedRT; a host that wants
 * the stages chained passes RT_IMAGE_POSTPROCESS as src.  Images made from a
 * tile-sharded context (tileCount > 1) cover only this rank's tiles: gather
 * first and hand the whole frame in with rt_image_upload. */

/* A whole frame from the host (w*h float4, row 0 = bottom) -> RT_IMAGE_USER.
 * No reference counterpart: stands for a RenderTexture produced elsewhere. */
int rt_image_upload(RtContext *ctx, const float *rgba, int32_t width, int32_t height);
int rt_image_size(RtContext *ctx, int which, int32_t *width, int32_t *height);

/* PostProcessBase.RenderImage (PostProcessBase.cs:29-32) and the stack's own
 * fallback (PostProcessStack.cs:34): a plain copy. */
int rt_post_blit(RtContext *ctx, int src);
/* ColorCorrection.RenderImage (ColorCorrection.cs:22-29, ColorCorrection.shader:46-70) */
int rt_post_color_correction(RtContext *ctx, int src, float brightness, float saturation, float contrast);
/* mat.SetTexture("_LutTex", lutTex) (ColorGrading.cs:15,23): width x height float4,
 * row 0 = v 0 (the bottom row of the image file); sampled with FilterMode.Point and
 * wrap Repeat as the reference's LUT assets are imported (the .png.meta files under Assets/Textures/Lut). */
int rt_post_lut_upload(RtContext *ctx, const float *rgba, int32_t width, int32_t height);
/* ColorGrading.RenderImage (ColorGrading.cs:26-29, ColorGrading.shader:46-68) */
int rt_post_color_grading(RtContext *ctx, int src);

#ifndef RT_WRAP_CLAMP
#define RT_WRAP_CLAMP 0
#define RT_WRAP_REPEAT 1
#endif
/* Bloom's public fields (Bloom.cs:7-29) plus the two sampler facts its shaders read. */
typedef struct RtBloomParams {
    float luminanceThreshold;     /* 1 */
    int32_t downSampleStep;       /* 7; >= 2 */
    int32_t downSampleBlurSize;   /* 5; 1..15 */
    float downSampleBlurSigma;    /* 1 */
    int32_t upSampleBlurSize;     /* 5; 1..15 */
    float upSampleBlurSigma;      /* 1 */
    float bloomIntensity;         /* 1 */
    /* `_MainTex_ST.xy` of the down/up-sample materials, which the shaders use as the
     * tap spacing in uv units (DownSample.shader:80, UpSample.shader:83-84).  A
     * material made from a shader has (1, 1); {0, 0} here means (1, 1).  A host that
     * wants texel-spaced taps passes (1/w, 1/h)-like values itself. */
    float mainTexST[2];
    int32_t wrapMode;             /* RT_WRAP_CLAMP 0 (RenderTexture default) or RT_WRAP_REPEAT 1 */
} RtBloomParams;
/* Bloom.RenderImage (Bloom.cs:41-115; Threshold, DownSample, UpSample, Bloom .shader) */
int rt_post_bloom(RtContext *ctx, int src, const RtBloomParams *params);

/* SaveTexture.SavePNG up to EncodeToPNG (SaveTexture.cs:23-38,59-90): ReadPixels into an
 * RGB24 texture (8-bit quantisation of the LINEAR value), LinearToSRGB of the
 * quantised value, SetPixels (second quantisation).  dst = width*height*3 bytes,
 * row 0 = bottom. */
int rt_save_rgb24(RtContext *ctx, int which, uint8_t *dst_rgb);

/* ------------------------------------------------------------------ tuning */
/* Execution knobs; none changes a result bit, except value 2 of RT_OPT_PRUNE_BY_DISTANCE as said there.  No reference
 * counterpart (the reference has one fixed schedule: a thread per pixel, a dispatch per frame).
 *   RT_OPT_PACKET_SHADOW   0 (default): 1 = the 16 soft-shadow rays of a pixel walk the tree as one packet
 *   RT_OPT_PACKET_PRIMARY  0 (default): 1 = camera rays walk the tree in packets of 16 neighbouring pixels
 *   RT_OPT_PACKET_TRACE    0 (default): 1 = rt_trace_rays walks caller rays 16 at a time
 * (packets pay off only where the 16 rays stay inside the same few leaves; on finely
 *  tessellated scenes the one-ray-per-lane walker is about twice as fast)
 *   RT_OPT_MATERIAL_SORT   1 (default): a shading block groups its 2048 paths by material class first
 *   RT_OPT_TWO_STREAMS     how many streams the frames of a batch (rt_render_frames) go down.  -1 (default): the library
 *                          chooses -- two frame lanes (each half of the frames with a stream, counter block, masked
 *                          streams and spill areas of its own, so that one half's launch ends overlap the other half's
 *                          work) for a batch of 8 M to 20 M paths, i.e. a share of the image (an eighth of the 1080p bench
 *                          image: 12.7 -> 12.1 ms per 64 frames), one stream otherwise (the whole image loses 4 % to the
 *                          split: the walkers already fill the chip).  0: one stream; 2 .. 4: that many frame lanes; 1: two
 *                          plain streams, the overlap of RT_OPT_OVERLAP_SHADOW off.  Same bits however the frames are split.
 *   RT_OPT_BATCH_FRAMES    1..64 frames rt_render_frames keeps in flight; 0 (default): as many as give
 *                          up to 128 M paths per launch (64 for a whole 1080p frame, 16 for a 4K frame; 220 bytes of
 *                          device memory per path, halved until it fits)
 *   RT_OPT_FIRST_HIT_SHADOW 1 (default): a ShadowMap ray only reports whether its closest hit casts a shadow, so a ray
 *                          that stays clear of the box round the scene's non-casting triangles (it can only meet casters)
 *                          ends its walk at the first hit it accepts.  Same shadow map bit for bit.
 *   RT_OPT_CASTER_BOX      1 (default): a ShadowMap ray that stays clear of the box round the shadow-casting triangles is lit
 *                          whatever else it meets; a pixel none of whose 16 rays passes that box gets no rays at all
 *                          (its shadow-map value is what 16 misses give).  Same shadow map bit for bit.
 *   RT_OPT_OVERLAP_SHADOW  1 (default): in rt_render_frames, when the ShadowMap pass shares CSMain's primary hits, its rays
 *                          are walked WHILE the later bounces run: one launch on a stream masked to all compute units but
 *                          a few, the remaining bounce and shading launches on a stream masked to those few, a last launch
 *                          that drains the same queue afterwards.  Same results bit for bit (a ray's result does not
 *                          depend on which launch walks it); 6 % faster for a whole 1080p image of the car scene, 21 %
 *                          for an eighth of it.  Off while counters or kernel timing are enabled (they report launches
 *                          one by one), with RT_OPT_PACKET_SHADOW, RT_OPT_TWO_STREAMS or the brute-force flag, and where
 *                          the runtime cannot make masked streams.
 *   RT_OPT_EAGER_SHADOWMAP 1 (default): the per-frame sequence rt_dispatch, rt_dispatch_shadowmap (RayTracing.cs:131-135) is
 *                          scheduled like a batch of one frame.  When RtParams carries RT_FLAG_MATTE_MASK (the host will
 *                          dispatch the ShadowMap kernel next, with the same uniforms) and both kernels trace the same
 *                          primary ray, rt_dispatch prepares the ShadowMap pass from CSMain's primary hits and, where
 *                          RT_OPT_OVERLAP_SHADOW applies, has its rays walked while the later bounces run;
 *                          rt_dispatch_shadowmap then drains what is left and folds the 16 results per pixel.  Same
 *                          images bit for bit; 6.4 instead of 9.8 ms per 1080p frame of the car scene.  Any call
 *                          that changes the scene, the parameters or an option in between drops the prepared pass, and
 *                          rt_dispatch_shadowmap traces its own primary rays as it does with the option off or while
 *                          counters are enabled (the work counters keep the two kernels apart).  Between the two calls
 *                          RT_IMAGE_SHADOWMAP is half written: read it after rt_dispatch_shadowmap, as the reference does.
 *   RT_OPT_PRUNE_BY_DISTANCE  Which boxes a ray may leave out because they begin beyond the closest hit it holds.  The
 *                          reference enters every box the ray passes (BVHRayTracing.cginc:87-123 has no distance test) and
 *                          keeps the hit with the smallest COMPUTED distance, so a box may be left out only if no triangle
 *                          in it can be computed nearer than the hit already held -- and the distance Moller-Trumbore
 *                          computes in float32 for a ray nearly inside a triangle's plane can lie far in front of the
 *                          triangle: distance, u and v are all divided by det = e1 . (d x e2), the small difference of
 *                          large products.  Values:
 *                            0  no box is left out: the reference's walk, box for box (the work counters of
 *                               rt_enable_counters then equal the CPU checker's)
 *                            1  (default) the bounded rule, the same hits as 0 bit for bit.  The reference's own cut
 *                               (IntersectTriangle_MT97 drops det < 1e-8) bounds how far a computed distance can be off:
 *                               by 192 |e1||e2| |o - v0| at most.  For triangles that are SMALL (|e1||e2| <= 2^-12 and
 *                               |e1| + |e2| <= 1/4 in scene units -- like the cut, an absolute measure) that is a few per
 *                               cent of the distance, and a box is left out when the plane through its nearest corner at
 *                               right angles to the ray lies beyond the closest hit by more than that (derivation in
 *                               csrc/rt_traverse.h, "leaving out boxes, with a bound").  Boxes that hold a large triangle
 *                               are never left out; a scene of large triangles only is walked as with 0.  Used by camera
 *                               rays, rt_trace_rays* and bounce rays; shadow rays leave out nothing (they end at a first
 *                               hit where that settles the answer, RT_OPT_FIRST_HIT_SHADOW)
 *                            2  every ray leaves out what begins more than 1/64 beyond its closest hit, large triangles
 *                               or not: NOT the reference's result where a computed distance is that far off (within
 *                               4e-6 rad of a well-shaped triangle's plane, 4e-4 rad of a sliver's; the CPU checker's
 *                               model of the rule loses 13 hits in a million on a scene built for it,
 *                               tests/adversarial.py, tests/test_oracle_cpu.py).  The 10 M-triangle stress launch, all
 *                               large triangles: 32 instead of 106 ms
 *   RT_OPT_SHARE_LONG_WALKS 1 (default): a walker launch lasts as long as its longest ray, a chain of dependent fetches
 *                          that can run to thousands of steps.  Once a launch has no rays left to deal, lanes of a wave
 *                          that have run out of rays take over stacked subtrees of the lanes that are still walking
 *                          (the entries of a ray's stack are independent, and no result depends on the order they are
 *                          walked in: the closest hit wins, equal distances go to the triangle the reference meets
 *                          first) and hand their hit back when done.  Same results bit for bit.  Off while counters
 *                          are enabled (a ray that may stop at its first hit would be counted differently from run to
 *                          run).
 *   RT_OPT_WIDE_NODES      1 (default): the walkers fetch two levels of the traversal tree per step -- a 64-byte record with
 *                          the boxes of a node's (up to four) grandchildren -- go on with the nearest one the ray may pass
 *                          and stack the others farthest first: half as many dependent fetches per ray.  Inner boxes only
 *                          filter (a triangle is tested when its own box passes, which any walk that skips no subtree
 *                          finds), so the hits are the same bit for bit; 0 = one level per step.  The kernels that count
 *                          work (rt_enable_counters) walk the same way as the others: internalVisits counts the places of
 *                          the records fetched that hold a child (two per node visited with one level per step).
 *   RT_OPT_PIPELINE_FRAMES 1 (default): the per-frame call sequence -- rt_set_params (new seed), rt_dispatch, rt_dispatch_shadowmap,
 *                          rt_accumulate per frame, as RayTracing.CustomRender makes it (RayTracing.cs:118-164) -- keeps FOUR
 *                          frames in flight: consecutive frames go round four lanes (stream, frame of path state, counter
 *                          block, masked streams, spill areas of their own), so the launches of a frame fill the idle time
 *                          of the frames before, most of whose launches are as long as their longest ray (one frame at a
 *                          time 4.2 ms, two lanes 3.0, three 2.6, four 2.4 -- 2.15 with RT_OPT_OVERLAP_CUS as it is chosen for them; six and eight lanes are slower again).  The folds
 *                          into the running mean stay in frame order (each waits for the one before), so the images keep
 *                          their bits; RT_IMAGE_RT / RT_IMAGE_SHADOWMAP show the frame of the last rt_dispatch.  Any call
 *                          other than those four (and rt_set_dirty) first lets all lanes run out.  2, 3, 4: that many
 *                          lanes; 0: one frame at a time.  Off while counters or kernel timing are enabled, with
 *                          RT_OPT_TWO_STREAMS and with the brute-force flag.
 *   RT_OPT_CHAIN_BOUNCES   3 (default): where a call has few paths in flight (2^23 at most: the per-frame calls, a 4K frame, four
 *                          1080p frames), the bounces from the fourth on -- a few thousand rays each, whose launches last as
 *                          long as their longest ray -- are one launch: a wave takes 64 paths of the fourth bounce's queue
 *                          and brings them to their ends by itself, walk and shading in turn, so that nothing waits for a
 *                          launch (one 1080p frame at a time 3.9 -> 3.3 ms, the per-frame calls with four frames in flight
 *                          2.18 -> 1.92 ms per frame).  Paths are independent: same bits.  1..7: from that bounce on
 *                          (0-based; it has to lie after RT_OPT_OVERLAP_AFTER); 0: every bounce a launch of its own.
 *                          Off while counters are enabled and with the brute-force flag.
 *   RT_OPT_OVERLAP_AFTER   0..6, default 0: the bounce whose extend launch the overlap starts after (0: as soon as the
 *                          primary hits are known)
 *   RT_OPT_OVERLAP_CUS     compute units the bounce launches keep meanwhile (multiples of 32 measured best); default 0: 64 of
 *                          256, or 96 where the call has few paths in flight (under 24 M -- the per-frame calls above all:
 *                          the bounce chain is then what a frame waits for).  The masked streams of a split are made when
 *                          a context first renders with it and are shared by the contexts of the process on that device */
#define RT_OPT_PACKET_SHADOW 0
#define RT_OPT_PACKET_PRIMARY 1
#define RT_OPT_PACKET_TRACE 2
#define RT_OPT_BATCH_FRAMES 3
#define RT_OPT_MATERIAL_SORT 4
#define RT_OPT_TWO_STREAMS 5
#define RT_OPT_OVERLAP_SHADOW 6
#define RT_OPT_OVERLAP_AFTER 7
#define RT_OPT_OVERLAP_CUS 8
#define RT_OPT_FIRST_HIT_SHADOW 9
#define RT_OPT_CASTER_BOX 10
#define RT_OPT_EAGER_SHADOWMAP 11
#define RT_OPT_PRUNE_BY_DISTANCE 12
#define RT_OPT_SHARE_LONG_WALKS 13
#define RT_OPT_WIDE_NODES 14
#define RT_OPT_PIPELINE_FRAMES 15
#define RT_OPT_CHAIN_BOUNCES 16
int rt_set_option(RtContext *ctx, int option, int value);

/* ------------------------------------------------------------------- stats */
/* When enabled, traversal kernels count node/leaf/triangle tests (slower). */
int rt_enable_counters(RtContext *ctx, int enable);
/* When enabled, every kernel launch is bracketed by HIP events on the library's
 * stream; rt_kernel_time reports per-kernel totals. */
int rt_enable_kernel_timing(RtContext *ctx, int enable);
int rt_reset_stats(RtContext *ctx);
int rt_get_stats(RtContext *ctx, RtStats *out);
/* Name and mean device time (ms) of the dominant kernel since the last
 * rt_reset_stats, measured with HIP events on the library's own stream. */
int rt_kernel_time(RtContext *ctx, const char *kernel_name, double *out_total_ms, uint64_t *out_launches);
int rt_synchronize(RtContext *ctx);
const char *rt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RTCORE_H_ */
