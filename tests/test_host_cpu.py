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
        owner = ((ty // 32) * tiles_x + tx // 32) % world
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
