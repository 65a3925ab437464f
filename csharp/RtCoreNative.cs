// RtCoreNative.cs -- P/Invoke binding of librtcore.so for the Unity host.
//
// Drop this file into Assets/Scripts/ and place librtcore.so in
// Assets/Plugins/x86_64/.  Struct layouts are the reference's own
// (Assets/Scripts/BVH/SceneDataTypes.cs, BVH.cs:15-37,
// LightImportanceSampling.cs:16-55), so existing managed arrays are passed
// as they are.  This file cannot be compiled in the build environment of the
// native library (no C# toolchain there); the same calls are exercised through
// the Python mirror raytracing_amd/host.py.
using System;
using System.Runtime.InteropServices;
using UnityEngine;

public static class RtCoreNative
{
    const string Lib = "rtcore";

    public const int RT_OK = 0, RT_E_INVALID = -1, RT_E_DEVICE = -2, RT_E_NOMEM = -3, RT_E_STATE = -4, RT_E_OVERFLOW = -5;
    public const uint RT_TRACE_STATIC_MAX_RAYS = 1u << 24;
    public const uint RT_FLAG_DEPTH_OF_FIELD = 0x10, RT_FLAG_DOF_PRIMARY = 0x20, RT_FLAG_MATTE_MASK = 0x40, RT_FLAG_BRUTE_FORCE = 0x80;
    public const int RT_IMAGE_RT = 0, RT_IMAGE_CONVERGED = 1, RT_IMAGE_SHADOWMAP = 2, RT_IMAGE_POSTPROCESS = 3, RT_IMAGE_USER = 4;
    public const int RT_WRAP_CLAMP = 0, RT_WRAP_REPEAT = 1;

    [StructLayout(LayoutKind.Sequential)]
    public struct MeshVertex { public Vector3 position, normal, tangent; public Vector2 uv; }          // 44 B

    [StructLayout(LayoutKind.Sequential)]
    public struct MaterialData                                                                          // 56 B
    {
        public Vector3 albedo; public float metallic, smoothness, transparent; public Vector3 emissionColor;
        public uint materialType; public float ior; public Vector3 clearCoatColor;
    }

    [StructLayout(LayoutKind.Sequential)] public struct SphereLight { public Vector3 position; public float radius; }
    [StructLayout(LayoutKind.Sequential)] public struct AreaLight { public Vector3 position, normal, up; public Vector2 size; }
    [StructLayout(LayoutKind.Sequential)] public struct DiscLight { public Vector3 position, normal; public float radius; }

    [StructLayout(LayoutKind.Sequential)]
    public struct RtParams
    {
        public int width, height;
        public Matrix4x4 camera2World;              // Unity memory order (column-major), as the library expects
        public Matrix4x4 cameraInverseProjection;
        public float seed;
        public Vector4 lightParameter, lightColor, shadowParameter;
        public float skyboxRotation, skyboxExposure, lensRadius, focalLength;
        public uint flags;
        public int tileRank, tileCount, tileSize;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct RtStats                            // 17 eight-byte fields, 136 B
    {
        public ulong rays, internalVisits, leafVisits, triTests, hits, paths; public double buildMs, renderMs;
        public ulong shadowRays, shadowInternalVisits, shadowLeafVisits, shadowTriTests, shadowHits;
        public ulong raysWalked, shadowRaysWalked;   // rays / shadowRays count every ray ANSWERED; these two only the ones a tree walk answered
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct RtBloomParams                      // Bloom.cs:7-29 + the two sampler facts its shaders read
    {
        public float luminanceThreshold; public int downSampleStep, downSampleBlurSize; public float downSampleBlurSigma;
        public int upSampleBlurSize; public float upSampleBlurSigma, bloomIntensity;
        public Vector2 mainTexST;                    // (0,0) = the material default (1,1)
        public int wrapMode;
    }

    // Brute-force primitive records (useBVH = false).  Field for field the private nested structs of
    // MeshCollector (MeshCollector.cs:47-156: normal/position/size, center/radius, min/max, matrix/offset/count,
    // each followed by the Primitive block albedo, metallic, smoothness, transparent, emissionColor); the host
    // either makes those internal and passes its own arrays, or fills these.
    [StructLayout(LayoutKind.Sequential)]
    public struct Plane                                                                                 // 72 B
    {
        public Vector3 normal, position, size;
        public Vector3 albedo; public float metallic, smoothness, transparent; public Vector3 emissionColor;
    }
    [StructLayout(LayoutKind.Sequential)]
    public struct Sphere                                                                                // 52 B
    {
        public Vector3 center; public float radius;
        public Vector3 albedo; public float metallic, smoothness, transparent; public Vector3 emissionColor;
    }
    [StructLayout(LayoutKind.Sequential)]
    public struct Cube                                                                                  // 60 B
    {
        public Vector3 min, max;
        public Vector3 albedo; public float metallic, smoothness, transparent; public Vector3 emissionColor;
    }
    [StructLayout(LayoutKind.Sequential)]
    public struct CustomMesh                                                                            // 108 B
    {
        public Matrix4x4 localToWorldMatrix; public int indicesOffset, indicesCount;
        public Vector3 albedo; public float metallic, smoothness, transparent; public Vector3 emissionColor;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct RtHit { public float t, u, v; public uint triangle; }                                // 16 B, rt_trace_rays_device*

    // ---- context
    [DllImport(Lib)] public static extern int rt_create(int deviceId, out IntPtr ctx);
    [DllImport(Lib)] public static extern int rt_destroy(IntPtr ctx);
    [DllImport(Lib)] public static extern IntPtr rt_last_error(IntPtr ctx);
    [DllImport(Lib)] public static extern IntPtr rt_version();
    [DllImport(Lib)] public static extern int rt_synchronize(IntPtr ctx);

    // ---- scene (BVH.Init, LightImportanceSampling.Init, RayTracing.InitShader)
    [DllImport(Lib)] public static extern int rt_scene_upload(IntPtr ctx, MeshVertex[] vertices, uint nVertices, uint[] indices, uint nIndices,
                                                               uint[] materialIndex, Vector2Int[] shadowIndex, uint nTriangles);
    [DllImport(Lib)] public static extern int rt_materials_upload(IntPtr ctx, MaterialData[] materials, uint n);
    [DllImport(Lib)] public static extern int rt_lights_upload(IntPtr ctx, SphereLight[] sphere, uint nSphere, AreaLight[] area, uint nArea,
                                                                DiscLight[] disc, uint nDisc);
    [DllImport(Lib)] public static extern int rt_skybox_upload(IntPtr ctx, float[] facesRgba, uint size);
    [DllImport(Lib)] public static extern int rt_primitives_upload(IntPtr ctx, Plane[] planes, uint nPlanes, Sphere[] spheres, uint nSpheres, Cube[] cubes, uint nCubes,
                                                                   CustomMesh[] customMeshes, uint nMeshes, Vector3[] vertices, Vector3[] normals, uint nVertices,
                                                                   int[] indices, uint nIndices);
    [DllImport(Lib)] public static extern int rt_trace_primitives(IntPtr ctx, Vector3[] origins, Vector3[] directions, uint n, float[] outT, uint[] outPrimitive,
                                                                  Vector3[] outNormal);

    // ---- BVH (BVH.Build and the buffers BVH.cs:133-160 binds)
    [DllImport(Lib)] public static extern int rt_bvh_build(IntPtr ctx);
    [DllImport(Lib)] public static extern int rt_readback_bvh(IntPtr ctx, uint[] morton, uint[] sortedIndex, AABB[] triangleAabb, Triangle[] triangles,
                                                               InternalNode[] internalNodes, LeafNode[] leafNodes, AABB[] bvhData);
    [DllImport(Lib)] public static extern int rt_readback_sort_stages(IntPtr ctx, uint[] mortonUnsorted, uint[] mortonSortedRaw);
    [DllImport(Lib)] public static extern int rt_readback_walk_tree(IntPtr ctx, uint[] nodes8, uint[] leafTriangle, uint[] leafRank);

    // ---- render (RayTracing.CustomRender)
    [DllImport(Lib)] public static extern int rt_set_params(IntPtr ctx, ref RtParams p);
    [DllImport(Lib)] public static extern int rt_dispatch(IntPtr ctx);
    [DllImport(Lib)] public static extern int rt_dispatch_shadowmap(IntPtr ctx);
    [DllImport(Lib)] public static extern int rt_accumulate(IntPtr ctx, uint samplePerPixel, int useShadowMap);
    [DllImport(Lib)] public static extern int rt_render_frames(IntPtr ctx, float[] seeds, uint nFrames);
    [DllImport(Lib)] public static extern int rt_set_dirty(IntPtr ctx);
    [DllImport(Lib)] public static extern uint rt_sample_count(IntPtr ctx);
    [DllImport(Lib)] public static extern int rt_readback(IntPtr ctx, int which, Color[] dstRgba);     // Color = 4 floats
    [DllImport(Lib)] public static extern int rt_upload_converged(IntPtr ctx, Color[] srcRgba, uint samplePerPixel);

    // ---- one frame split over several GPUs: one context per device, RtParams.tileRank = 0..tileCount-1.
    // In one process: render every context (a thread each), then rt_readback_merge each into ONE Color[width*height].
    // One process per GPU: rt_readback_local_device into a device buffer that the process group gathers;
    // rt_local_pixel_map says which pixel each tile-local slot is.
    [DllImport(Lib)] public static extern int rt_readback_merge(IntPtr ctx, int which, Color[] frameRgba);
    [DllImport(Lib)] public static extern int rt_readback_local_device(IntPtr ctx, int which, IntPtr deviceDst);
    // the same merge into a frame that stays in device memory (of this context's device or another one): no host hop
    [DllImport(Lib)] public static extern int rt_merge_local_device(IntPtr ctx, int which, IntPtr deviceFrameRgba, int frameDevice);
    // rt_readback in the background into page-locked memory (a pinned GCHandle's address); complete after rt_readback_wait
    [DllImport(Lib)] public static extern int rt_readback_async(IntPtr ctx, int which, IntPtr pinnedDstRgba);
    [DllImport(Lib)] public static extern int rt_readback_wait(IntPtr ctx);
    [DllImport(Lib)] public static extern uint rt_local_pixel_count(IntPtr ctx);
    [DllImport(Lib)] public static extern int rt_local_pixel_map(IntPtr ctx, uint[] dst);

    // ---- bare ray queries (Tracing.cginc BVHRayTrace without shading)
    [DllImport(Lib)] public static extern int rt_trace_rays(IntPtr ctx, Vector3[] origins, Vector3[] directions, uint n, float[] outT, uint[] outTriangle,
                                                            float[] outU, float[] outV);
    [DllImport(Lib)] public static extern int rt_trace_rays_device(IntPtr ctx, IntPtr deviceOrigins4, IntPtr deviceDirs4, uint n, IntPtr deviceHits4);
    // the same, for callers that must not allocate device memory (n <= RT_TRACE_STATIC_MAX_RAYS; calls on one GPU must not overlap)
    [DllImport(Lib)] public static extern int rt_trace_rays_device_static(IntPtr ctx, IntPtr deviceOrigins4, IntPtr deviceDirs4, uint n, IntPtr deviceHits4);

    // ---- options, counters, timing
    [DllImport(Lib)] public static extern int rt_set_option(IntPtr ctx, int option, int value);
    public const int RT_OPT_PACKET_SHADOW = 0, RT_OPT_PACKET_PRIMARY = 1, RT_OPT_PACKET_TRACE = 2, RT_OPT_BATCH_FRAMES = 3,
                     RT_OPT_MATERIAL_SORT = 4, RT_OPT_TWO_STREAMS = 5, RT_OPT_OVERLAP_SHADOW = 6, RT_OPT_OVERLAP_AFTER = 7,
                     RT_OPT_OVERLAP_CUS = 8, RT_OPT_FIRST_HIT_SHADOW = 9, RT_OPT_CASTER_BOX = 10, RT_OPT_EAGER_SHADOWMAP = 11,
                     RT_OPT_PRUNE_BY_DISTANCE = 12, RT_OPT_SHARE_LONG_WALKS = 13, RT_OPT_WIDE_NODES = 14, RT_OPT_PIPELINE_FRAMES = 15,
                     RT_OPT_CHAIN_BOUNCES = 16;
    [DllImport(Lib)] public static extern int rt_enable_counters(IntPtr ctx, int enable);
    [DllImport(Lib)] public static extern int rt_enable_kernel_timing(IntPtr ctx, int enable);
    [DllImport(Lib)] public static extern int rt_reset_stats(IntPtr ctx);
    [DllImport(Lib)] public static extern int rt_get_stats(IntPtr ctx, out RtStats stats);
    [DllImport(Lib, CharSet = CharSet.Ansi)] public static extern int rt_kernel_time(IntPtr ctx, string kernelName, out double totalMs, out ulong launches);
    [DllImport(Lib)] public static extern int rt_debug_eval(IntPtr ctx, int which, float[] x, float[] y, uint n, float[] result);

    // ---- post-process chain: the destination is always RT_IMAGE_POSTPROCESS (RayTracing.postProcessRT)
    [DllImport(Lib)] public static extern int rt_image_upload(IntPtr ctx, Color[] rgba, int width, int height);
    [DllImport(Lib)] public static extern int rt_image_size(IntPtr ctx, int which, out int width, out int height);
    [DllImport(Lib)] public static extern int rt_post_blit(IntPtr ctx, int src);
    [DllImport(Lib)] public static extern int rt_post_color_correction(IntPtr ctx, int src, float brightness, float saturation, float contrast);
    [DllImport(Lib)] public static extern int rt_post_lut_upload(IntPtr ctx, Color[] rgba, int width, int height);
    [DllImport(Lib)] public static extern int rt_post_color_grading(IntPtr ctx, int src);
    [DllImport(Lib)] public static extern int rt_post_bloom(IntPtr ctx, int src, ref RtBloomParams p);
    [DllImport(Lib)] public static extern int rt_save_rgb24(IntPtr ctx, int which, byte[] dstRgb);

    public static void Check(IntPtr ctx, int rc)
    {
        if (rc != RT_OK) Debug.LogError("rtcore error " + rc + ": " + Marshal.PtrToStringAnsi(rt_last_error(ctx)));
    }

    // One frame over several GPUs from one process: `contexts[i]` lives on device i and has been given
    // RtParams.tileRank = i, tileCount = contexts.Length.  Each context renders on a thread of its own;
    // the tiles are then merged into `frame` (width*height Colors, row 0 = bottom).
    public static void RenderSplitFrame(IntPtr[] contexts, float[] seeds, Color[] frame)
    {
        var threads = new System.Threading.Thread[contexts.Length];
        var codes = new int[contexts.Length];
        for (int i = 0; i < contexts.Length; ++i)
        {
            int k = i;
            threads[k] = new System.Threading.Thread(() => { codes[k] = rt_render_frames(contexts[k], seeds, (uint)seeds.Length); });
            threads[k].Start();
        }
        for (int i = 0; i < contexts.Length; ++i) { threads[i].Join(); Check(contexts[i], codes[i]); }
        for (int i = 0; i < contexts.Length; ++i) Check(contexts[i], rt_readback_merge(contexts[i], RT_IMAGE_CONVERGED, frame));
    }

    // The same with the frame kept on a GPU: every context writes its tiles into `deviceFrame` (width*height float4 in the
    // memory of device `frameDevice`, e.g. the one that presents or post-processes), device to device.
    public static void RenderSplitFrameOnDevice(IntPtr[] contexts, float[] seeds, IntPtr deviceFrame, int frameDevice)
    {
        var threads = new System.Threading.Thread[contexts.Length];
        var codes = new int[contexts.Length];
        for (int i = 0; i < contexts.Length; ++i)
        {
            int k = i;
            threads[k] = new System.Threading.Thread(() =>
            {
                codes[k] = rt_render_frames(contexts[k], seeds, (uint)seeds.Length);
                if (codes[k] == RT_OK) codes[k] = rt_merge_local_device(contexts[k], RT_IMAGE_CONVERGED, deviceFrame, frameDevice);
            });
            threads[k].Start();
        }
        for (int i = 0; i < contexts.Length; ++i) { threads[i].Join(); Check(contexts[i], codes[i]); }
    }
}
