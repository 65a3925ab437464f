"""Blittable wire formats of the rtcore C ABI as numpy dtypes / ctypes structs.

Byte layouts follow the reference's GPU structs:
Assets/Shaders/BVH/Constants.cginc:11-78, Assets/Scripts/BVH/SceneDataTypes.cs:6-120,
Assets/Shaders/BVH/BVHRayTracing.cginc:6-17, Assets/Scripts/BVH/BVH.cs:15-37,
Assets/Shaders/RayTracing/ImportanceSampling.cginc:8-39.
"""
import ctypes as C

import numpy as np

AABB = np.dtype([("min", "<f4", 3), ("_d0", "<f4"), ("max", "<f4", 3), ("_d1", "<f4")])
TRIANGLE = np.dtype([
    ("point0", "<f4", 3), ("_d0", "<f4"), ("point1", "<f4", 3), ("_d1", "<f4"),
    ("point2", "<f4", 3), ("_d2", "<f4"),
    ("uv0", "<f4", 2), ("uv1", "<f4", 2), ("uv2", "<f4", 2), ("_d3", "<f4", 2),
    ("normal0", "<f4", 3), ("_d4", "<f4"), ("normal1", "<f4", 3), ("_d5", "<f4"),
    ("normal2", "<f4", 3), ("_d6", "<f4"),
    ("tangent0", "<f4", 3), ("_d7", "<f4"), ("tangent1", "<f4", 3), ("_d8", "<f4"),
    ("tangent2", "<f4", 3), ("_d9", "<f4"),
    ("materialIndex", "<i4"), ("castShadow", "<i4"), ("receiveShadow", "<i4"), ("_d10", "<f4"),
])
INTERNAL_NODE = np.dtype([("leftNode", "<u4"), ("leftNodeType", "<u4"), ("rightNode", "<u4"),
                          ("rightNodeType", "<u4"), ("parent", "<u4"), ("index", "<u4")])
LEAF_NODE = np.dtype([("parent", "<u4"), ("index", "<u4")])
MATERIAL = np.dtype([("albedo", "<f4", 3), ("metallic", "<f4"), ("smoothness", "<f4"),
                     ("transparent", "<f4"), ("emissionColor", "<f4", 3), ("materialType", "<u4"),
                     ("ior", "<f4"), ("clearCoatColor", "<f4", 3)])
MESH_VERTEX = np.dtype([("position", "<f4", 3), ("normal", "<f4", 3), ("tangent", "<f4", 3), ("uv", "<f4", 2)])
SPHERE_LIGHT = np.dtype([("position", "<f4", 3), ("radius", "<f4")])
AREA_LIGHT = np.dtype([("position", "<f4", 3), ("normal", "<f4", 3), ("up", "<f4", 3), ("size", "<f4", 2)])
DISC_LIGHT = np.dtype([("position", "<f4", 3), ("normal", "<f4", 3), ("radius", "<f4")])

# BruteForceRayTracing.cginc:5-65 = MeshCollector.cs:46-156
_SURFACE = [("albedo", "<f4", 3), ("metallic", "<f4"), ("smoothness", "<f4"), ("transparent", "<f4"), ("emissionColor", "<f4", 3)]
PLANE = np.dtype([("normal", "<f4", 3), ("position", "<f4", 3), ("size", "<f4", 3)] + _SURFACE)
SPHERE = np.dtype([("center", "<f4", 3), ("radius", "<f4")] + _SURFACE)
CUBE = np.dtype([("min", "<f4", 3), ("max", "<f4", 3)] + _SURFACE)
CUSTOM_MESH = np.dtype([("localToWorldMatrix", "<f4", 16), ("indicesOffset", "<i4"), ("indicesCount", "<i4")] + _SURFACE)
assert PLANE.itemsize == 72 and SPHERE.itemsize == 52 and CUBE.itemsize == 60 and CUSTOM_MESH.itemsize == 108
PRIM_PLANE, PRIM_SPHERE, PRIM_CUBE, PRIM_TRIANGLE = 0, 1, 2, 3

assert AABB.itemsize == 32 and TRIANGLE.itemsize == 192          # MeshBufferContainer.cs:54-62
assert INTERNAL_NODE.itemsize == 24 and LEAF_NODE.itemsize == 8
assert MATERIAL.itemsize == 56 and MESH_VERTEX.itemsize == 44
assert SPHERE_LIGHT.itemsize == 16 and AREA_LIGHT.itemsize == 44 and DISC_LIGHT.itemsize == 28

INTERNAL, LEAF = 0, 1

# RayTracing.cs:40-48
SAMPLING_UNIFORM, SAMPLING_COSINE, SAMPLING_LIGHT, SAMPLING_BSDF, SAMPLING_MIS = range(5)
FLAG_DEPTH_OF_FIELD = 0x10
FLAG_DOF_PRIMARY = 0x20
FLAG_MATTE_MASK = 0x40
FLAG_BRUTE_FORCE = 0x80

IMAGE_RT, IMAGE_CONVERGED, IMAGE_SHADOWMAP, IMAGE_POSTPROCESS, IMAGE_USER = 0, 1, 2, 3, 4
WRAP_CLAMP, WRAP_REPEAT = 0, 1


class RtParams(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("camera2World", C.c_float * 16),
        ("cameraInverseProjection", C.c_float * 16),
        ("seed", C.c_float),
        ("lightParameter", C.c_float * 4),
        ("lightColor", C.c_float * 4),
        ("shadowParameter", C.c_float * 4),
        ("skyboxRotation", C.c_float), ("skyboxExposure", C.c_float),
        ("lensRadius", C.c_float), ("focalLength", C.c_float),
        ("flags", C.c_uint32),
        ("tileRank", C.c_int32), ("tileCount", C.c_int32), ("tileSize", C.c_int32),
    ]

    def copy(self):
        other = RtParams()
        C.memmove(C.byref(other), C.byref(self), C.sizeof(RtParams))
        return other


class RtStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("internalVisits", C.c_uint64), ("leafVisits", C.c_uint64),
                ("triTests", C.c_uint64), ("hits", C.c_uint64), ("paths", C.c_uint64),
                ("buildMs", C.c_double), ("renderMs", C.c_double),
                ("shadowRays", C.c_uint64), ("shadowInternalVisits", C.c_uint64), ("shadowLeafVisits", C.c_uint64),
                ("shadowTriTests", C.c_uint64), ("shadowHits", C.c_uint64),
                ("raysWalked", C.c_uint64), ("shadowRaysWalked", C.c_uint64)]

# rt_set_option
OPT_PACKET_SHADOW, OPT_PACKET_PRIMARY, OPT_PACKET_TRACE, OPT_BATCH_FRAMES, OPT_MATERIAL_SORT, OPT_TWO_STREAMS = 0, 1, 2, 3, 4, 5
OPT_OVERLAP_SHADOW, OPT_OVERLAP_AFTER, OPT_OVERLAP_CUS = 6, 7, 8
OPT_FIRST_HIT_SHADOW, OPT_CASTER_BOX, OPT_EAGER_SHADOWMAP, OPT_PRUNE_BY_DISTANCE = 9, 10, 11, 12
OPT_SHARE_LONG_WALKS, OPT_WIDE_NODES, OPT_PIPELINE_FRAMES, OPT_CHAIN_BOUNCES = 13, 14, 15, 16


class RtBloomParams(C.Structure):
    """Bloom's public fields with their defaults (Bloom.cs:7-29)."""
    _fields_ = [("luminanceThreshold", C.c_float), ("downSampleStep", C.c_int32), ("downSampleBlurSize", C.c_int32),
                ("downSampleBlurSigma", C.c_float), ("upSampleBlurSize", C.c_int32), ("upSampleBlurSigma", C.c_float),
                ("bloomIntensity", C.c_float), ("mainTexST", C.c_float * 2), ("wrapMode", C.c_int32)]

    def __init__(self, **kw):
        super().__init__()
        self.luminanceThreshold, self.downSampleStep = 1.0, 7
        self.downSampleBlurSize, self.downSampleBlurSigma = 5, 1.0
        self.upSampleBlurSize, self.upSampleBlurSigma = 5, 1.0
        self.bloomIntensity = 1.0
        self.mainTexST[0], self.mainTexST[1] = 1.0, 1.0
        self.wrapMode = WRAP_CLAMP
        for k, v in kw.items():
            if k == "mainTexST":
                self.mainTexST[0], self.mainTexST[1] = v
            else:
                setattr(self, k, v)
