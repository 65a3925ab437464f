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
    public struct RtStats
    {
        public ulong rays, internalVisits, leafVisits, triTests, hits, paths; public double buildMs, renderMs;
        public ulong shadowRays, shadowInternalVisits, shadowLeafVisits, shadowTriTests, shadowHits;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct RtBloomParams                      // Bloom.cs:7-29 + the two sampler facts its shaders read
    {
        public float luminanceThreshold; public int downSampleStep, downSampleBlurSize; public float downSampleBlurSigma;
        public int upSampleBlurSize; public float upSampleBlurSigma, bloomIntensity;
        public Vector2 mainTexST;                    // (0,0) = the material default (1,1)
        public int wrapMode;
    }

    [DllImport(Lib)] public static extern int rt_create(int deviceId, out IntPtr ctx);
    [DllImport(Lib)] public static extern int rt_destroy(IntPtr ctx);
    [DllImport(Lib)] public static extern IntPtr rt_last_error(IntPtr ctx);
    [DllImport(Lib)] public static extern int rt_scene_upload(IntPtr ctx, MeshVertex[] vertices, uint nVertices, uint[] indices, uint nIndices,
                                                               uint[] materialIndex, Vector2Int[] shadowIndex, uint nTriangles);
    [DllImport(Lib)] public static extern int rt_materials_upload(IntPtr ctx, MaterialData[] materials, uint n);
    [DllImport(Lib)] public static extern int rt_lights_upload(IntPtr ctx, SphereLight[] sphere, uint nSphere, AreaLight[] area, uint nArea,
                                                                DiscLight[] disc, uint nDisc);
    [DllImport(Lib)] public static extern int rt_skybox_upload(IntPtr ctx, float[] facesRgba, uint size);
    [DllImport(Lib)] public static extern int rt_bvh_build(IntPtr ctx);
    [DllImport(Lib)] public static extern int rt_readback_bvh(IntPtr ctx, uint[] morton, uint[] sortedIndex, AABB[] triangleAabb, Triangle[] triangles,
                                                               InternalNode[] internalNodes, LeafNode[] leafNodes, AABB[] bvhData);
    [DllImport(Lib)] public static extern int rt_readback_walk_tree(IntPtr ctx, uint[] nodes8, uint[] leafTriangle, uint[] leafRank);
    [DllImport(Lib)] public static extern int rt_set_params(IntPtr ctx, ref RtParams p);
    [DllImport(Lib)] public static extern int rt_dispatch(IntPtr ctx);
    [DllImport(Lib)] public static extern int rt_dispatch_shadowmap(IntPtr ctx);
    [DllImport(Lib)] public static extern int rt_accumulate(IntPtr ctx, uint samplePerPixel, int useShadowMap);
    [DllImport(Lib)] public static extern int rt_render_frames(IntPtr ctx, float[] seeds, uint nFrames);
    [DllImport(Lib)] public static extern int rt_set_dirty(IntPtr ctx);
    [DllImport(Lib)] public static extern uint rt_sample_count(IntPtr ctx);
    [DllImport(Lib)] public static extern int rt_readback(IntPtr ctx, int which, Color[] dstRgba);     // Color = 4 floats
    [DllImport(Lib)] public static extern int rt_upload_converged(IntPtr ctx, Color[] srcRgba, uint samplePerPixel);
    [DllImport(Lib)] public static extern int rt_get_stats(IntPtr ctx, out RtStats stats);
    [DllImport(Lib)] public static extern int rt_set_option(IntPtr ctx, int option, int value);
    public const int RT_OPT_PACKET_SHADOW = 0, RT_OPT_PACKET_PRIMARY = 1, RT_OPT_PACKET_TRACE = 2, RT_OPT_BATCH_FRAMES = 3,
                     RT_OPT_MATERIAL_SORT = 4, RT_OPT_TWO_STREAMS = 5, RT_OPT_OVERLAP_SHADOW = 6, RT_OPT_OVERLAP_AFTER = 7,
                     RT_OPT_OVERLAP_CUS = 8, RT_OPT_FIRST_HIT_SHADOW = 9, RT_OPT_CASTER_BOX = 10;

    // brute-force primitives (useBVH = false): T = the host's own Plane / Sphere / Cube / CustomMesh structs (MeshCollector.cs:46-156)
    [DllImport(Lib)] public static extern int rt_primitives_upload(IntPtr ctx, Array planes, uint nPlanes, Array spheres, uint nSpheres, Array cubes, uint nCubes,
                                                                   Array customMeshes, uint nMeshes, Vector3[] vertices, Vector3[] normals, uint nVertices,
                                                                   int[] indices, uint nIndices);
    [DllImport(Lib)] public static extern int rt_trace_primitives(IntPtr ctx, Vector3[] origins, Vector3[] directions, uint n, float[] outT, uint[] outPrimitive,
                                                                  Vector3[] outNormal);

    // post-process chain: the destination is always RT_IMAGE_POSTPROCESS (RayTracing.postProcessRT)
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
}
