"""CPU tests of the host side: the C-ABI library loads and exports what the
header declares, struct layouts match the C compiler's, scene/host logic, and
the world_size-2 exchange path on the gloo backend.  No GPU compute here."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from raytracing_amd import dist as rtdist
from raytracing_amd import host, native, scenes, wire

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    txt = open(os.path.join(ROOT, "include", "rtcore.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = native.load()
    declared = _header_functions()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/rtcore.h but not exported by librtcore.so"
    assert sorted(native.EXPORTS) == declared, "raytracing_amd.native.EXPORTS is out of sync with include/rtcore.h"
    assert b"gfx950" in lib.rt_version()


def test_python_constants_follow_the_header():
    """Option numbers, flags and image selectors of raytracing_amd.wire are the header's."""
    txt = open(os.path.join(ROOT, "include", "rtcore.h")).read()
    defines = {k: int(v, 0) for k, v in re.findall(r"^#define\s+(RT_[A-Z_0-9]+)\s+(0x[0-9A-Fa-f]+|\d+)u?\s*$", txt, flags=re.M)}
    options = {k[len("RT_"):]: v for k, v in defines.items() if k.startswith("RT_OPT_")}
    assert len(options) >= 11
    for name, value in options.items():
        assert getattr(wire, name) == value, name
    for name in ("FLAG_BRUTE_FORCE", "FLAG_MATTE_MASK", "FLAG_DEPTH_OF_FIELD", "FLAG_DOF_PRIMARY", "IMAGE_CONVERGED", "IMAGE_RT",
                 "IMAGE_SHADOWMAP", "IMAGE_POSTPROCESS", "IMAGE_USER"):
        if "RT_" + name in defines:
            assert getattr(wire, name) == defines["RT_" + name], name


def test_no_gpu_means_error_code_not_crash():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = native.load()
    ctx = C.c_void_p()
    assert lib.rt_create(0, C.byref(ctx)) == -2 and not ctx.value          # RT_E_DEVICE
    assert lib.rt_destroy(None) == 0                                        # idempotent on NULL
    assert lib.rt_dispatch(None) == -1                                      # RT_E_INVALID
    with pytest.raises(native.RtError):
        native.RtCore(0)


def test_struct_layouts_match_the_c_compiler():
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "rtcore.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu ", sizeof(RtAABB), sizeof(RtTriangle), sizeof(RtInternalNode), sizeof(RtLeafNode),
         sizeof(RtMaterialData), sizeof(RtMeshVertex), sizeof(RtSphereLight), sizeof(RtAreaLight), sizeof(RtDiscLight), sizeof(RtParams));
  printf("%zu %zu %zu %zu %zu %zu %zu\n", offsetof(RtParams, seed), offsetof(RtParams, lightParameter), offsetof(RtParams, skyboxRotation),
         offsetof(RtParams, flags), offsetof(RtParams, tileRank), sizeof(RtStats), offsetof(RtTriangle, materialIndex));
  return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        out = list(map(int, subprocess.check_output([os.path.join(d, "t")]).split()))
    assert out[:9] == [32, 192, 24, 8, 56, 44, 16, 44, 28]                 # MeshBufferContainer.cs:54-62 asserts 192 / 32 too
    P = wire.RtParams
    assert out[9] == C.sizeof(P)
    assert out[10:15] == [P.seed.offset, P.lightParameter.offset, P.skyboxRotation.offset, P.flags.offset, P.tileRank.offset]
    assert out[15] == C.sizeof(wire.RtStats) and out[16] == wire.TRIANGLE.fields["materialIndex"][1]


def test_batch_frames_rule_keeps_ray_indices_in_32_bits():
    """Frames in flight: 128 M paths at most and never more than 64 frames; an explicit option is held to the
    same path budget, so 16 shadow rays per path still index in 32 bits (the rule is a host function)."""
    lib = native.load()
    f = lib.rt_debug_batch_frames
    f.restype = C.c_uint32
    f.argtypes = [C.c_uint32] * 3
    hd, uhd = 1920 * 1080, 3840 * 2160
    assert f(hd, 0, 1000) == 64 and f(uhd, 0, 1000) == 16 and f(hd, 0, 5) == 5 and f(hd, 0, 0) == 1
    assert f(hd, 8, 1000) == 8 and f(hd, 1000, 1000) == 64 and f(uhd, 64, 1000) == 16      # the option cannot lift the path budget
    assert f(0, 0, 3) == 3 and f(1 << 27, 0, 9) == 1 and f((1 << 28) - 1, 64, 9) == 1
    for slots in (1, 4096, hd, uhd, 7680 * 4320, (1 << 28) - 1):
        for option in (0, 1, 7, 64, 4000):
            got = f(slots, option, 1 << 20)
            assert 1 <= got <= 64 and (got == 1 or slots * got * 16 < 1 << 32), (slots, option, got)


def test_division_by_a_divisor_fixed_with_the_tile_map():
    """Kernels turn a path slot into pixel coordinates with one multiply-high and one correction per division
    (RtDivisor: tile pixels, tiles per row, pixel blocks per tile row, slots per frame).  Quotient and remainder
    must be the integer ones for every numerator below 2^32 and every divisor up to 2^31: edge numerators for many
    divisors, and random pairs."""
    import numpy as np
    lib = native.load()
    f = lib.rt_debug_divide
    f.restype = C.c_uint32
    f.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    rem = C.c_uint32()
    rs = np.random.Generator(np.random.PCG64(17))
    divisors = [1, 2, 3, 5, 7, 8, 30, 34, 60, 64, 4096, 4097, 65535, 65536, 65537, 1920 * 1080, 2088960, 8355840, (1 << 27) + 1,
                (1 << 28) - 1, (1 << 31) - 1, 1 << 31] + [int(x) for x in rs.integers(1, 1 << 31, 200)]
    for d in divisors:
        edge = [0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, (1 << 32) - 1, (1 << 32) - d, ((1 << 32) // d) * d - 1, ((1 << 32) // d) * d]
        for n in edge + [int(x) for x in rs.integers(0, 1 << 32, 60)]:
            n = min(max(n, 0), (1 << 32) - 1)
            q = f(n, d, C.byref(rem))
            assert (q, rem.value) == (n // d, n % d), (n, d, q, rem.value)


def _c_prototypes():
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "rtcore.h")).read(), flags=re.S)
    protos = {}
    for m in re.finditer(r"^(int|uint32_t|const char \*)\s*(rt_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.M | re.S):
        args = [" ".join(a.split()) for a in m.group(3).split(",")]
        protos[m.group(2)] = (m.group(1).strip(), [] if args == ["void"] else args)
    return protos


def _kind_c(arg):
    if "*" in arg:
        return "pointer"
    t = arg.rsplit(" ", 1)[0].replace("const ", "")
    return {"int": "i32", "int32_t": "i32", "uint32_t": "u32", "float": "f32"}[t]


def _kind_cs(arg):
    words = arg.split()
    if words[0] in ("ref", "out") or words[0].endswith("[]") or words[0] in ("IntPtr", "string"):
        return "pointer"
    return {"int": "i32", "uint": "u32", "float": "f32"}[words[0]]


def test_csharp_binding_declares_every_export_with_matching_arguments():
    """The Unity-side binding cannot be compiled here, so its declarations are checked as text:
    every prototype of the header has exactly one [DllImport], with the same number of arguments and
    the same scalar-or-pointer kind in every position, and no argument is an untyped System.Array."""
    text = open(os.path.join(ROOT, "csharp", "RtCoreNative.cs")).read()
    code = re.sub(r"//[^\n]*", "", text)
    found = re.findall(r"\[DllImport\(Lib[^\]]*\)\]\s*public static extern (\w+) (rt_\w+)\(([^)]*)\)\s*;", code)
    names = [n for _, n, _ in found]
    assert len(names) == len(set(names))
    protos = _c_prototypes()
    assert set(names) == set(protos), (set(protos) - set(names), set(names) - set(protos))
    assert set(protos) == set(native.EXPORTS)
    for ret, name, args in found:
        c_ret, c_args = protos[name]
        cs_args = [" ".join(a.split()) for a in args.split(",")] if args.strip() else []
        assert len(cs_args) == len(c_args), name
        assert ret == {"int": "int", "uint32_t": "uint", "const char *": "IntPtr"}[c_ret], name
        for a, b in zip(cs_args, c_args):
            assert not a.startswith("Array "), (name, a)
            assert _kind_cs(a) == _kind_c(b), (name, a, b)
    # the counters struct: as many 8-byte fields as the C struct holds, doubles in the same places
    body = re.search(r"public struct RtStats[^{]*\{(.*?)\n    \}", code, flags=re.S).group(1)
    fields = []
    for kind, lst in re.findall(r"public (ulong|double) ([^;]+);", body):
        fields += [kind] * len(lst.split(","))
    want = ["double" if isinstance(t, type) and t is C.c_double else "ulong" for _, t in wire.RtStats._fields_]
    assert fields == want
    for const in ("RT_OPT_EAGER_SHADOWMAP = 11", "RT_TRACE_STATIC_MAX_RAYS = 1u << 24", "RT_IMAGE_USER = 4"):
        assert const in code


# ------------------------------------------------------------------ scene / host logic
def test_material_data_follows_the_reference_override_order():
    m = scenes.material_data((0.5, 0.5, 0.5, 0.25), 0.0, 0.0)
    assert abs(float(m["albedo"][0]) - 0.21404114) < 1e-6                   # Color.linear
    assert m["metallic"] == np.float32(0.01) and m["smoothness"] == np.float32(0.01) and m["transparent"] == -1
    t = scenes.material_data((0.5, 0.5, 0.5, 0.25), mode=3, material_type=3, clear_coat_color=(1, 0, 0))
    assert t["materialType"] == 1 and t["transparent"] == np.float32(0.25) and tuple(t["clearCoatColor"]) == (1, 0, 0)
    e = scenes.material_data((1, 1, 1, 0.5), mode=3, emission=(2, 3, 4))
    assert e["materialType"] == 2 and tuple(e["emissionColor"]) == (2, 3, 4)   # emission wins over transparent
    assert scenes.material_data()["ior"] == 1.0


def test_material_presets_follow_the_reference_table():
    """CustomMaterialInfo.cs:21-102 on stand-in renderers Sphere_0..Sphere_5 (ratio k/5), without a device."""
    from types import SimpleNamespace
    rs = [SimpleNamespace(name=f"Sphere_{k}", sharedMaterial=host.Material(color=(0.8, 0.6, 0.4, 1.0))) for k in range(6)]
    info = host.CustomMaterialInfo(SimpleNamespace(core=None), rs)
    table = lambda: np.array([r.sharedMaterial.MaterialData() for r in rs], wire.MATERIAL)
    ratio = np.arange(6, dtype=np.float32) / 5
    info.OnMaterialChanged(0)                                  # dielectric, smoothness sweep
    m = table()
    assert (m["metallic"] == np.float32(0.01)).all() and (m["materialType"] == 0).all() and (m["transparent"] == -1).all()
    np.testing.assert_array_equal(m["smoothness"], np.maximum(ratio, np.float32(0.01)))
    info.OnMaterialChanged(1)                                  # metal, smoothness sweep
    m = table()
    assert (m["metallic"] == 1).all() and (m["ior"] == 1).all()
    info.OnMaterialChanged(3)                                  # glass, ior sweep 1.0 .. 1.5
    m = table()
    assert (m["materialType"] == 1).all() and (m["transparent"] == np.float32(0.1)).all() and (m["smoothness"] == 1).all()
    np.testing.assert_allclose(m["ior"], 1 + ratio * 0.5, rtol=1e-6)
    info.OnMaterialChanged(4)                                  # clear coat: opaque again (mode 0), type 3
    m = table()
    assert (m["materialType"] == 3).all() and (m["transparent"] == -1).all() and (m["metallic"] == 1).all()
    info.OnMaterialChanged(5)                                  # transparency sweep
    m = table()
    np.testing.assert_array_equal(m["transparent"], ratio)
    assert (m["ior"] == np.float32(1.1)).all()
    gold = np.load(os.path.join(ROOT, "tests", "golden", "unity_chapter6_5_scene.npz"))["materials_preset"]
    assert gold.shape[0] == 6 and gold.dtype == wire.MATERIAL


def test_cornell_scene_geometry():
    sc = scenes.cornell_box(32, 32, sphere_segments=16)
    assert sc.n_triangles == len(sc.indices) // 3 == len(sc.shadow_index)
    assert len(sc.area_lights) == 1 and len(sc.sphere_lights) == 0
    al = sc.area_lights[0]
    np.testing.assert_allclose(al["normal"], [0, -1, 0], atol=1e-6)          # lamp faces down
    np.testing.assert_allclose(al["size"], [6, 6.0000012], rtol=1e-6)
    p = sc.vertices["position"][sc.indices.reshape(-1, 3)]
    fn = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
    np.testing.assert_allclose(fn[0] / np.linalg.norm(fn[0]), [0, 1, 0], atol=1e-6)    # floor faces up
    np.testing.assert_allclose(fn[2] / np.linalg.norm(fn[2]), [-1, 0, 0], atol=1e-6)   # right wall faces -x
    assert (np.abs(sc.vertices["position"]).max() < 125).all()              # inside the Morton box
    c2w = np.array(sc.params.camera2World[:]).reshape(4, 4).T
    np.testing.assert_allclose(c2w[:3, 3], [0, 5.8, -21.2], rtol=1e-6)
    np.testing.assert_allclose(c2w[:3, 2], [0, 0, -1])                       # camera looks down -Z in camera space


def test_pcg32_reference_values():
    g = scenes.PCG32(42, stream=54)
    assert [g.next_u32() for _ in range(3)] == [0xa15c02b7, 0x7b47f409, 0xba1d3330]   # published PCG32 demo output
    s = scenes.frame_seeds(5, seed=1)
    assert s.dtype == np.float32 and ((s >= 0) & (s < 1)).all() and len(set(s.tolist())) == 5


def test_host_mirror_merges_like_bvh_initmesh():
    grey, red = host.Material((0.6, 0.6, 0.6, 1)), host.Material((0.9, 0, 0, 1))
    q = scenes.quad_mesh()
    mrs = [host.MeshRenderer(q, grey, host.Transform((0, 0, 0), (0.7071068, 0, 0, 0.7071068), (20, 20, 1)), shadowCastingOn=False),
           host.MeshRenderer(q, red, host.Transform((10, 7.5, 0))),
           host.MeshRenderer(q, grey, host.Transform((0, 15, 0)), enabled=False),
           host.MeshRenderer(q, grey, host.Transform((0, 15, 0)))]
    bvh = host.BVH(mrs)
    (v, i, mi, sh), mats = bvh.InitMesh()
    assert len(v) == 12 and len(i) == 18 and mi.tolist() == [0, 0, 1, 1, 0, 0]          # disabled renderer skipped, material reused
    assert mats == [grey, red] and sh.tolist()[:2] == [[0, 1], [0, 1]] and sh.tolist()[2] == [1, 1]
    assert i.reshape(-1, 3)[2].tolist() == [4, 7, 5]                                     # index offset per mesh
    np.testing.assert_allclose(v["normal"][0], [0, 1, 0], atol=1e-5)                     # (0,0,-1) rotated +90 deg about X
    np.testing.assert_allclose(v["tangent"][0], [20, 0, 0], atol=1e-5)                   # TransformVector: scaled, not renormalised
    md = grey.MaterialData()
    assert md["materialType"] == 0 and md["transparent"] == -1


def test_tile_map_partitions_the_frame():
    w, h, ts = 200, 136, 32
    seen = np.zeros(w * h, int)
    for world in (1, 3):
        seen[:] = 0
        for r in range(world):
            m = rtdist.tile_pixel_map(w, h, ts, r, world)
            assert len(m) % (ts * ts) == 0
            v = m[m != rtdist.PAD]
            seen[v] += 1
        assert (seen == 1).all()
    m = rtdist.tile_pixel_map(64, 64, 64, 0, 1)
    assert m[:8].tolist() == list(range(8)) and m[8] == 64 and m[64] == 8          # 8x8 blocks, row-major inside


def test_tile_gatherer_places_by_a_precomputed_permutation():
    import torch
    w, h, ts = 200, 136, 32
    for n in (1, 2, 5):
        g = rtdist.TileGatherer(w, h, ts, torch.device("cpu"), contexts=n)
        assert g.source.dtype == torch.int64 and len(g.source) == w * h and len(set(g.source.tolist())) == w * h
        for twice in range(2):                                                     # buffers are reused from image to image
            for r in range(n):
                m = rtdist.tile_pixel_map(w, h, ts, r, n)
                local = np.full((len(m), 4), -7.0, np.float32)                     # padding slots carry a value that must not land
                ok = m != rtdist.PAD
                local[ok, 0], local[ok, 1], local[ok, 2] = m[ok], r, twice
                g.rank_buffer(r).copy_(torch.from_numpy(local.reshape(-1)))
            f = g.place().numpy()
            assert np.array_equal(f[..., 0].reshape(-1), np.arange(w * h, dtype=np.float32))
            ty, tx = np.mgrid[0:h, 0:w]
            owner = rtdist.tile_owner((ty // ts) * ((w + ts - 1) // ts) + tx // ts, n)
            assert np.array_equal(f[..., 1], owner.astype(np.float32)) and (f[..., 2] == twice).all()
        with pytest.raises(RuntimeError):
            g.gather()
    single = rtdist.TileGatherer(64, 64, 64, torch.device("cpu"))                  # no process group: world of one
    single.local_buffer().copy_(torch.arange(64 * 64 * 4, dtype=torch.float32))
    m = rtdist.tile_pixel_map(64, 64, 64, 0, 1)
    assert np.array_equal(single.gather().numpy().reshape(-1, 4)[m][:, 0], np.arange(64 * 64, dtype=np.float32) * 4)


def test_scene_broadcast_refuses_primitive_lists():
    import torch
    with pytest.raises(ValueError):                                                # raised before any collective is entered
        rtdist.broadcast_scene(scenes.primitive_room(32, 32), torch.device("cpu"))


# ------------------------------------------------------------------ world_size 2 on gloo
def _worker(rank, world, port, tmp):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    sc = scenes.cornell_box(96, 80, sphere_segments=8) if rank == 0 else None
    got = rtdist.broadcast_scene(sc, dev)
    ref = scenes.cornell_box(96, 80, sphere_segments=8)
    for k in rtdist.SCENE_ARRAYS:
        a, b = getattr(got, k), getattr(ref, k)
        assert (a is None and b is None) or (a.dtype == b.dtype and a.tobytes() == b.tobytes()), k
    assert bytes(got.params) == bytes(ref.params)
    # each rank "renders" value = global pixel index into its own tile-local buffer
    g = rtdist.TileGatherer(96, 80, 32, dev)
    m = rtdist.tile_pixel_map(96, 80, 32, rank, world)
    local = np.zeros((len(m), 4), np.float32)
    local[:, 0] = np.where(m != rtdist.PAD, m, -1)
    local[:, 1] = rank
    g.local_buffer().copy_(torch.from_numpy(local.reshape(-1)))
    frame = g.gather()
    if rank == 0:
        f = frame.numpy()
        assert np.array_equal(f[..., 0].reshape(-1), np.arange(96 * 80, dtype=np.float32))
        tiles_x = 3
        ty, tx = np.mgrid[0:80, 0:96]
        owner = rtdist.tile_owner((ty // 32) * tiles_x + tx // 32, world)
        assert np.array_equal(f[..., 1], owner.astype(np.float32))
        open(os.path.join(tmp, "ok"), "w").write("ok")
    else:
        assert frame is None
    dist.destroy_process_group()


def test_scene_broadcast_and_tile_gather_world2_gloo():
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as tmp:
        port = 29500 + (os.getpid() % 2000)
        mp.spawn(_worker, args=(2, port, tmp), nprocs=2, join=True)
        assert os.path.exists(os.path.join(tmp, "ok"))
