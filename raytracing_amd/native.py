"""ctypes binding of librtcore.so (include/rtcore.h).

This is the only route from Python to the HIP kernels.  There is no fallback:
if the shared library is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os

import numpy as np

from . import wire

_HERE = os.path.dirname(os.path.abspath(__file__))
# RTCORE_LIB lets a developer A/B another build of the same library; the default is the in-tree one.
LIB_PATH = os.environ.get("RTCORE_LIB") or os.path.join(_HERE, "librtcore.so")

# every symbol include/rtcore.h declares
EXPORTS = [
    "rt_create", "rt_destroy", "rt_last_error", "rt_scene_upload", "rt_materials_upload", "rt_lights_upload",
    "rt_skybox_upload", "rt_bvh_build", "rt_readback_bvh", "rt_readback_sort_stages", "rt_set_params", "rt_dispatch",
    "rt_dispatch_shadowmap", "rt_accumulate", "rt_render_frames", "rt_set_dirty", "rt_sample_count", "rt_readback",
    "rt_readback_local_device", "rt_upload_converged", "rt_local_pixel_count", "rt_local_pixel_map", "rt_trace_rays",
    "rt_trace_rays_device", "rt_enable_counters", "rt_enable_kernel_timing", "rt_reset_stats", "rt_get_stats",
    "rt_kernel_time", "rt_synchronize", "rt_version", "rt_debug_eval", "rt_set_option",
    "rt_image_upload", "rt_image_size", "rt_post_blit", "rt_post_color_correction", "rt_post_lut_upload",
    "rt_post_color_grading", "rt_post_bloom", "rt_save_rgb24", "rt_primitives_upload", "rt_trace_primitives",
    "rt_readback_walk_tree", "rt_trace_rays_device_static", "rt_readback_merge", "rt_merge_local_device",
    "rt_readback_async", "rt_readback_wait",
]

_lib = None


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"rtcore error {code}: {message}")
        self.code = code


def _open(path):
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} is missing: build it with `make -C raytracing_amd/csrc`")
    # One HIP runtime per process: the torch wheel carries its own libamdhip64 (soname libamdhip64.so.7, file
    # name libamdhip64.so).  Loaded first, it also satisfies librtcore's libamdhip64.so.7; loaded second, the
    # loader takes it for a different library, and the second runtime in a process finds no devices.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    lib.rt_last_error.restype = C.c_char_p
    lib.rt_version.restype = C.c_char_p
    lib.rt_sample_count.restype = C.c_uint32
    lib.rt_local_pixel_count.restype = C.c_uint32
    for name in EXPORTS:
        getattr(lib, name)
    return lib


def load(path=None):
    """Load librtcore.so (or another build of it at `path`); raises if it has not been built
    (python __graft_entry__.py build)."""
    global _lib
    if path is not None:
        return _open(path)
    if _lib is None:
        _lib = _open(LIB_PATH)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class RtCore:
    """One context = one GPU (rt_create .. rt_destroy)."""

    def __init__(self, device_id=0, lib_path=None):
        self.lib = load(lib_path)
        self.ctx = C.c_void_p()
        rc = self.lib.rt_create(C.c_int(device_id), C.byref(self.ctx))
        if rc != 0:
            raise RtError(rc, "rt_create failed (no usable HIP device?)")
        self.n_triangles = 0
        self.params = None
        self.device_id = device_id

    def _check(self, rc):
        if rc != 0:
            raise RtError(rc, self.lib.rt_last_error(self.ctx).decode())

    def close(self):
        if self.ctx:
            self.lib.rt_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- scene
    def scene_upload(self, vertices, indices, material_index, shadow_index):
        v = np.ascontiguousarray(vertices, wire.MESH_VERTEX)
        i = np.ascontiguousarray(indices, np.uint32)
        mi = np.ascontiguousarray(material_index, np.uint32)
        sh = np.ascontiguousarray(shadow_index, np.int32)
        self._check(self.lib.rt_scene_upload(self.ctx, _p(v), C.c_uint32(len(v)), _p(i), C.c_uint32(len(i)), _p(mi), _p(sh),
                                             C.c_uint32(len(mi))))
        self.n_triangles = len(mi)

    def materials_upload(self, materials):
        m = np.ascontiguousarray(materials, wire.MATERIAL)
        self._check(self.lib.rt_materials_upload(self.ctx, _p(m), C.c_uint32(len(m))))

    def lights_upload(self, sphere=None, area=None, disc=None):
        s = np.ascontiguousarray(sphere if sphere is not None else np.zeros(0, wire.SPHERE_LIGHT), wire.SPHERE_LIGHT)
        a = np.ascontiguousarray(area if area is not None else np.zeros(0, wire.AREA_LIGHT), wire.AREA_LIGHT)
        d = np.ascontiguousarray(disc if disc is not None else np.zeros(0, wire.DISC_LIGHT), wire.DISC_LIGHT)
        self._check(self.lib.rt_lights_upload(self.ctx, _p(s), C.c_uint32(len(s)), _p(a), C.c_uint32(len(a)), _p(d),
                                              C.c_uint32(len(d))))

    def skybox_upload(self, faces):
        if faces is None:
            self._check(self.lib.rt_skybox_upload(self.ctx, None, C.c_uint32(0)))
            return
        f = np.ascontiguousarray(faces, np.float32)
        assert f.ndim == 4 and f.shape[0] == 6 and f.shape[1] == f.shape[2] and f.shape[3] == 4
        self._check(self.lib.rt_skybox_upload(self.ctx, _p(f), C.c_uint32(f.shape[1])))

    def primitives_upload(self, prims):
        """MeshCollector.Init: `prims` is a scenes.Primitives."""
        pl = np.ascontiguousarray(prims.planes, wire.PLANE)
        sp = np.ascontiguousarray(prims.spheres, wire.SPHERE)
        cu = np.ascontiguousarray(prims.cubes, wire.CUBE)
        me = np.ascontiguousarray(prims.meshes, wire.CUSTOM_MESH)
        v = np.ascontiguousarray(prims.vertices, np.float32).reshape(-1, 3)
        n = np.ascontiguousarray(prims.normals, np.float32).reshape(-1, 3)
        assert v.shape == n.shape
        i = np.ascontiguousarray(prims.indices, np.int32)
        self._check(self.lib.rt_primitives_upload(self.ctx, _p(pl), C.c_uint32(len(pl)), _p(sp), C.c_uint32(len(sp)), _p(cu),
                                                  C.c_uint32(len(cu)), _p(me), C.c_uint32(len(me)), _p(v), _p(n),
                                                  C.c_uint32(len(v)), _p(i), C.c_uint32(len(i))))

    def trace_primitives(self, origins, directions):
        o = np.ascontiguousarray(origins, np.float32)
        d = np.ascontiguousarray(directions, np.float32)
        n = len(o)
        t, prim, nrm = np.zeros(n, np.float32), np.zeros(n, np.uint32), np.zeros((n, 3), np.float32)
        self._check(self.lib.rt_trace_primitives(self.ctx, _p(o), _p(d), C.c_uint32(n), _p(t), _p(prim), _p(nrm)))
        return dict(t=t, primitive=prim, normal=nrm)

    def upload(self, scene):
        """Everything BVH.Init (or MeshCollector.Init) + LightImportanceSampling.Init + InitShader upload."""
        if scene.primitives is not None:
            self.primitives_upload(scene.primitives)
        if scene.n_triangles == 0 and scene.primitives is not None:
            self.lights_upload(scene.sphere_lights, scene.area_lights, scene.disc_lights)
            self.skybox_upload(scene.skybox)
            return
        self.scene_upload(scene.vertices, scene.indices, scene.material_index, scene.shadow_index)
        self.materials_upload(scene.materials)
        self.lights_upload(scene.sphere_lights, scene.area_lights, scene.disc_lights)
        self.skybox_upload(scene.skybox)

    # ---- build
    def bvh_build(self):
        self._check(self.lib.rt_bvh_build(self.ctx))

    def readback_bvh(self):
        n = self.n_triangles
        out = dict(morton=np.zeros(n, np.uint32), sorted_index=np.zeros(n, np.uint32),
                   tri_aabb=np.zeros(n, wire.AABB), triangles=np.zeros(n, wire.TRIANGLE),
                   internal=np.zeros(max(n - 1, 0), wire.INTERNAL_NODE), leaves=np.zeros(n, wire.LEAF_NODE),
                   bvh_data=np.zeros(max(n - 1, 0), wire.AABB))
        self._check(self.lib.rt_readback_bvh(self.ctx, _p(out["morton"]), _p(out["sorted_index"]), _p(out["tri_aabb"]),
                                             _p(out["triangles"]), _p(out["internal"]), _p(out["leaves"]),
                                             _p(out["bvh_data"])))
        un, raw = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        self._check(self.lib.rt_readback_sort_stages(self.ctx, _p(un), _p(raw)))
        out["morton_unsorted"], out["morton_sorted_raw"] = un, raw
        return out

    def readback_walk_tree(self):
        """The tree the rays walk: nodes u32[N-1, 8], per leaf the triangle and its reference visiting rank."""
        n = self.n_triangles
        nodes, tri, rank = np.zeros((max(n - 1, 0), 8), np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        self._check(self.lib.rt_readback_walk_tree(self.ctx, _p(nodes), _p(tri), _p(rank)))
        return dict(nodes=nodes, leaf_triangle=tri, leaf_rank=rank)

    # ---- render
    def set_params(self, params):
        self.params = params.copy()
        self._check(self.lib.rt_set_params(self.ctx, C.byref(self.params)))

    def dispatch(self):
        self._check(self.lib.rt_dispatch(self.ctx))

    def dispatch_shadowmap(self):
        self._check(self.lib.rt_dispatch_shadowmap(self.ctx))

    def accumulate(self, sample_per_pixel, use_shadow_map=False):
        self._check(self.lib.rt_accumulate(self.ctx, C.c_uint32(sample_per_pixel), C.c_int(int(use_shadow_map))))

    def render_frames(self, seeds):
        s = np.ascontiguousarray(seeds, np.float32)
        self._check(self.lib.rt_render_frames(self.ctx, _p(s), C.c_uint32(len(s))))

    def set_dirty(self):
        self._check(self.lib.rt_set_dirty(self.ctx))

    @property
    def sample_count(self):
        return int(self.lib.rt_sample_count(self.ctx))

    def readback(self, which=wire.IMAGE_CONVERGED):
        w, h = self.image_size(which)
        img = np.zeros((h, w, 4), np.float32)
        self._check(self.lib.rt_readback(self.ctx, C.c_int(which), _p(img)))
        return img

    def readback_merge(self, which, frame):
        """Write this context's tiles into `frame` (float32 [height, width, 4], C-contiguous); other pixels are kept."""
        if frame.dtype != np.float32 or not frame.flags.c_contiguous or frame.shape != (self.params.height, self.params.width, 4):
            raise ValueError("frame must be a C-contiguous float32 array of shape (height, width, 4)")
        self._check(self.lib.rt_readback_merge(self.ctx, C.c_int(which), _p(frame)))

    def readback_into(self, which, frame):
        """rt_readback into a float32 array the caller already holds (height x width x 4)."""
        if frame.dtype != np.float32 or not frame.flags["C_CONTIGUOUS"] or frame.size != self.params.width * self.params.height * 4:
            raise ValueError("frame must be a C-contiguous float32 array of height x width x 4")
        self._check(self.lib.rt_readback(self.ctx, C.c_int(which), _p(frame)))

    def merge_local_device(self, which, device_frame_ptr, frame_device=None):
        """This context's tiles into a full float4 frame in device memory (of `frame_device`, default: this context's)."""
        dev = self.device_id if frame_device is None else frame_device
        self._check(self.lib.rt_merge_local_device(self.ctx, C.c_int(which), C.c_void_p(device_frame_ptr), C.c_int(dev)))

    def readback_async(self, which, pinned_host_ptr):
        """Full-frame readback in the background into page-locked host memory; complete after readback_wait()."""
        self._check(self.lib.rt_readback_async(self.ctx, C.c_int(which), C.c_void_p(pinned_host_ptr)))

    def readback_wait(self):
        self._check(self.lib.rt_readback_wait(self.ctx))

    def readback_local_device(self, which, device_ptr):
        self._check(self.lib.rt_readback_local_device(self.ctx, C.c_int(which), C.c_void_p(device_ptr)))

    def upload_converged(self, image, sample_per_pixel):
        img = np.ascontiguousarray(image, np.float32)
        self._check(self.lib.rt_upload_converged(self.ctx, _p(img), C.c_uint32(sample_per_pixel)))

    @property
    def local_pixel_count(self):
        return int(self.lib.rt_local_pixel_count(self.ctx))

    def local_pixel_map(self):
        m = np.zeros(self.local_pixel_count, np.uint32)
        self._check(self.lib.rt_local_pixel_map(self.ctx, _p(m)))
        return m

    def trace_rays(self, origins, directions):
        o = np.ascontiguousarray(origins, np.float32)
        d = np.ascontiguousarray(directions, np.float32)
        n = len(o)
        t, tri = np.zeros(n, np.float32), np.zeros(n, np.uint32)
        u, v = np.zeros(n, np.float32), np.zeros(n, np.float32)
        self._check(self.lib.rt_trace_rays(self.ctx, _p(o), _p(d), C.c_uint32(n), _p(t), _p(tri), _p(u), _p(v)))
        return dict(t=t, triangle=tri, u=u, v=v)

    def trace_rays_device(self, origins4_ptr, dirs4_ptr, n, hits4_ptr):
        self._check(self.lib.rt_trace_rays_device(self.ctx, C.c_void_p(origins4_ptr), C.c_void_p(dirs4_ptr), C.c_uint32(n),
                                                  C.c_void_p(hits4_ptr)))

    def trace_rays_device_static(self, origins4_ptr, dirs4_ptr, n, hits4_ptr):
        """rt_trace_rays_device through the library's static scratch array: no allocation in the call."""
        self._check(self.lib.rt_trace_rays_device_static(self.ctx, C.c_void_p(origins4_ptr), C.c_void_p(dirs4_ptr), C.c_uint32(n),
                                                         C.c_void_p(hits4_ptr)))

    def debug_eval(self, which, x, y=None, n=None):
        x = np.ascontiguousarray(x, np.float32)
        y = x if y is None else np.ascontiguousarray(y, np.float32)
        n = x.size if n is None else n
        out = np.zeros(n, np.float32)
        self._check(self.lib.rt_debug_eval(self.ctx, C.c_int(which), _p(x), _p(y), C.c_uint32(n), _p(out)))
        return out

    # ---- post-process chain (whole frames; the destination is always IMAGE_POSTPROCESS)
    def image_upload(self, image):
        img = np.ascontiguousarray(image, np.float32)
        assert img.ndim == 3 and img.shape[2] == 4
        self._check(self.lib.rt_image_upload(self.ctx, _p(img), C.c_int32(img.shape[1]), C.c_int32(img.shape[0])))

    def image_size(self, which):
        w, h = C.c_int32(), C.c_int32()
        self._check(self.lib.rt_image_size(self.ctx, C.c_int(which), C.byref(w), C.byref(h)))
        return w.value, h.value

    def post_blit(self, src=wire.IMAGE_CONVERGED):
        self._check(self.lib.rt_post_blit(self.ctx, C.c_int(src)))

    def post_color_correction(self, src, brightness, saturation, contrast):
        self._check(self.lib.rt_post_color_correction(self.ctx, C.c_int(src), C.c_float(brightness), C.c_float(saturation),
                                                      C.c_float(contrast)))

    def post_lut_upload(self, lut):
        t = np.ascontiguousarray(lut, np.float32)
        assert t.ndim == 3 and t.shape[2] == 4
        self._check(self.lib.rt_post_lut_upload(self.ctx, _p(t), C.c_int32(t.shape[1]), C.c_int32(t.shape[0])))

    def post_color_grading(self, src):
        self._check(self.lib.rt_post_color_grading(self.ctx, C.c_int(src)))

    def post_bloom(self, src, bloom_params):
        self._check(self.lib.rt_post_bloom(self.ctx, C.c_int(src), C.byref(bloom_params)))

    def save_rgb24(self, which=wire.IMAGE_CONVERGED):
        w, h = self.image_size(which)
        out = np.zeros((h, w, 3), np.uint8)
        self._check(self.lib.rt_save_rgb24(self.ctx, C.c_int(which), _p(out)))
        return out

    # ---- tuning
    def set_option(self, option, value):
        self._check(self.lib.rt_set_option(self.ctx, C.c_int(int(option)), C.c_int(int(value))))

    # ---- stats
    def enable_counters(self, on=True):
        self._check(self.lib.rt_enable_counters(self.ctx, C.c_int(int(on))))

    def enable_kernel_timing(self, on=True):
        self._check(self.lib.rt_enable_kernel_timing(self.ctx, C.c_int(int(on))))

    def reset_stats(self):
        self._check(self.lib.rt_reset_stats(self.ctx))

    def stats(self):
        s = wire.RtStats()
        self._check(self.lib.rt_get_stats(self.ctx, C.byref(s)))
        return {name: getattr(s, name) for name, _ in wire.RtStats._fields_}

    def kernel_time(self, name):
        ms, n = C.c_double(), C.c_uint64()
        self._check(self.lib.rt_kernel_time(self.ctx, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def synchronize(self):
        self._check(self.lib.rt_synchronize(self.ctx))
