"""Synthetic scene inputs for the rtcore path tracer.

Everything the reference obtains from Unity objects (merged world-space mesh,
MaterialData table, light descriptors, camera matrices) is generated here from
fixed seeds, in the exact array formats `rt_scene_upload` & co. take.  The host
data preparation restated here:
  - mesh merge:       Assets/Scripts/BVH/BVH.cs:201-318 (TransformPoint for
                      positions, TransformVector for normals/tangents)
  - MaterialData:     Assets/Scripts/BVH/BVH.cs:38-76
  - light structs:    Assets/Scripts/RayTracing/LightImportanceSampling.cs:16-55
  - camera uniforms:  Assets/Scripts/RayTracing/RayTracing.cs:223-224
The reference's own meshes are binary FBX (and the car mesh and HDRIs are absent
from its checkout), so spheres, boxes and the car are procedural stand-ins.
"""
from dataclasses import dataclass, field

import numpy as np

from . import wire

F32 = np.float32


# ----------------------------------------------------------------- PCG32
class PCG32:
    """Host-side generator for seeds and synthetic geometry (PCG-XSH-RR 64/32)."""

    MULT = 6364136223846793005
    MASK = (1 << 64) - 1

    def __init__(self, seed, stream=54):
        self.state = 0
        self.inc = ((stream << 1) | 1) & self.MASK
        self.next_u32()
        self.state = (self.state + seed) & self.MASK
        self.next_u32()

    def next_u32(self):
        old = self.state
        self.state = (old * self.MULT + self.inc) & self.MASK
        xorshifted = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xorshifted >> rot) | (xorshifted << ((-rot) & 31))) & 0xFFFFFFFF

    def next_float(self):
        """Uniform float32 in [0,1) with 24 random bits."""
        return F32((self.next_u32() >> 8) * (1.0 / 16777216.0))

    def floats(self, n):
        return np.array([self.next_float() for _ in range(n)], dtype=F32)


def frame_seeds(n_frames, seed=1):
    """Per-frame `seed` uniforms (RayTracing.cs:230 uses UnityEngine.Random.value)."""
    return PCG32(seed).floats(n_frames)


# ----------------------------------------------------------------- meshes
@dataclass
class Mesh:
    positions: np.ndarray  # (V,3) f32
    normals: np.ndarray
    tangents: np.ndarray
    uvs: np.ndarray        # (V,2)
    triangles: np.ndarray  # (T,3) u32


def quad_mesh():
    """Unity built-in Quad: 1x1 in XY, normal -Z, triangles (0,3,1),(3,0,2)."""
    p = np.array([[-0.5, -0.5, 0], [0.5, -0.5, 0], [-0.5, 0.5, 0], [0.5, 0.5, 0]], F32)
    n = np.tile(np.array([0, 0, -1], F32), (4, 1))
    t = np.tile(np.array([1, 0, 0], F32), (4, 1))
    uv = np.array([[0, 0], [1, 0], [0, 1], [1, 1]], F32)
    tri = np.array([[0, 3, 1], [3, 0, 2]], np.uint32)
    return Mesh(p, n, t, uv, tri)


def uv_sphere_mesh(n_lon=64, n_lat=32, radius=1.0):
    """UV sphere, front faces outward under the reference's Moller-Trumbore
    culling convention (det > 0 for rays arriving from outside)."""
    lon = np.arange(n_lon + 1, dtype=np.float64) / n_lon * 2 * np.pi
    lat = np.arange(n_lat + 1, dtype=np.float64) / n_lat * np.pi
    st, ct = np.sin(lat)[:, None], np.cos(lat)[:, None]
    x = st * np.cos(lon)[None, :]
    z = st * np.sin(lon)[None, :]
    y = np.repeat(ct, n_lon + 1, axis=1)
    n = np.stack([x, y, z], -1).reshape(-1, 3)
    p = (n * radius).astype(F32)
    tang = np.stack([-np.sin(lon)[None, :].repeat(n_lat + 1, 0), np.zeros_like(x),
                     np.cos(lon)[None, :].repeat(n_lat + 1, 0)], -1).reshape(-1, 3)
    uv = np.stack([np.repeat(lon[None, :] / (2 * np.pi), n_lat + 1, 0),
                   np.repeat(1 - lat[:, None] / np.pi, n_lon + 1, 1)], -1).reshape(-1, 2)
    tris = []
    w = n_lon + 1
    for i in range(n_lat):
        for j in range(n_lon):
            a, b, c, d = i * w + j, i * w + j + 1, (i + 1) * w + j, (i + 1) * w + j + 1
            if i != 0:
                tris.append((a, b, c))
            if i != n_lat - 1:
                tris.append((b, d, c))
    return _ensure_outward(Mesh(p, n.astype(F32), tang.astype(F32), uv.astype(F32), np.array(tris, np.uint32)))


def box_mesh():
    """Unit cube centred at the origin, 24 vertices, 12 outward-facing triangles."""
    ps, ns, ts, uvs, tris = [], [], [], [], []
    axes = [((1, 0, 0), (0, 1, 0), (0, 0, 1)), ((-1, 0, 0), (0, 1, 0), (0, 0, -1)),
            ((0, 1, 0), (0, 0, 1), (1, 0, 0)), ((0, -1, 0), (0, 0, 1), (-1, 0, 0)),
            ((0, 0, 1), (1, 0, 0), (0, 1, 0)), ((0, 0, -1), (1, 0, 0), (0, -1, 0))]
    for n, u, v in axes:
        n, u, v = (np.array(a, np.float64) for a in (n, u, v))
        base = len(ps)
        for su, sv in ((-1, -1), (1, -1), (-1, 1), (1, 1)):
            ps.append(0.5 * (n + su * u + sv * v))
            ns.append(n)
            ts.append(u)
            uvs.append(((su + 1) / 2, (sv + 1) / 2))
        tris += [(base, base + 3, base + 1), (base + 3, base, base + 2)]
    return _ensure_outward(Mesh(np.array(ps, F32), np.array(ns, F32), np.array(ts, F32), np.array(uvs, F32),
                                np.array(tris, np.uint32)))


def cylinder_mesh(segments=20, radius=0.5, half_height=1.0):
    """A cylinder with the dimensions of Unity's built-in Cylinder (radius 0.5, height 2, 20 sides): a smooth-shaded
    side and two flat caps.  The exact vertex layout of the built-in asset is not reproduced."""
    ang = np.arange(segments + 1, dtype=np.float64) / segments * 2 * np.pi
    cx, cz = np.cos(ang), np.sin(ang)
    ps, ns, uvs, tris = [], [], [], []
    for j in range(segments + 1):                                   # side: two rings
        for y in (-half_height, half_height):
            ps.append((radius * cx[j], y, radius * cz[j]))
            ns.append((cx[j], 0.0, cz[j]))
            uvs.append((j / segments, (y + half_height) / (2 * half_height)))
    for j in range(segments):
        a = 2 * j
        tris += [(a, a + 1, a + 2), (a + 1, a + 3, a + 2)]
    for y, ny in ((-half_height, -1.0), (half_height, 1.0)):        # caps: fans round a centre vertex
        centre = len(ps)
        ps.append((0.0, y, 0.0)); ns.append((0.0, ny, 0.0)); uvs.append((0.5, 0.5))
        for j in range(segments + 1):
            ps.append((radius * cx[j], y, radius * cz[j])); ns.append((0.0, ny, 0.0)); uvs.append((0.5 + 0.5 * cx[j], 0.5 + 0.5 * cz[j]))
        for j in range(segments):
            tris.append((centre, centre + 1 + j, centre + 2 + j))
    n = len(ps)
    return _ensure_outward(Mesh(np.array(ps, F32), np.array(ns, F32), np.tile(np.array([1, 0, 0], F32), (n, 1)), np.array(uvs, F32),
                                np.array(tris, np.uint32)))


def _ensure_outward(mesh):
    """Orient every triangle so that cross(e1, e2) points along the vertex
    normals: IntersectTriangle_MT97 culls det = d . cross(e2, e1) < EPSILON, so
    such a face is hit only by rays travelling against its normal."""
    p = mesh.positions.astype(np.float64)
    t = mesh.triangles
    e1 = p[t[:, 1]] - p[t[:, 0]]
    e2 = p[t[:, 2]] - p[t[:, 0]]
    fn = np.cross(e1, e2)
    nv = mesh.normals.astype(np.float64)[t].sum(1)
    flip = (fn * nv).sum(1) < 0
    t2 = t.copy()
    t2[flip, 1], t2[flip, 2] = t[flip, 2], t[flip, 1]
    mesh.triangles = t2
    return mesh


# ----------------------------------------------------------------- transforms
def quat_rotate(q, v):
    """Rotate vectors v (...,3) by Unity quaternion q = (x,y,z,w)."""
    q = np.asarray(q, np.float64)
    u, w = q[:3], q[3]
    v = np.asarray(v, np.float64)
    return 2 * np.dot(v, u)[..., None] * u + (w * w - np.dot(u, u)) * v + 2 * w * np.cross(u, v)


def quat_from_euler_deg(x, y, z):
    """Unity Quaternion.Euler: rotate z, then x, then y (extrinsic)."""
    hx, hy, hz = (np.deg2rad(a) / 2 for a in (x, y, z))
    qx = np.array([np.sin(hx), 0, 0, np.cos(hx)])
    qy = np.array([0, np.sin(hy), 0, np.cos(hy)])
    qz = np.array([0, 0, np.sin(hz), np.cos(hz)])
    return quat_mul(qy, quat_mul(qx, qz))


def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw,
                     aw * bw - ax * bx - ay * by - az * bz])


IDENTITY_Q = (0.0, 0.0, 0.0, 1.0)


def trs_matrix(position, rotation=IDENTITY_Q, scale=(1, 1, 1)):
    """Matrix4x4.TRS as a 4x4 array (row r, column c)."""
    m = np.eye(4)
    for c in range(3):
        e = np.zeros(3)
        e[c] = scale[c]
        m[:3, c] = quat_rotate(rotation, e)
    m[:3, 3] = position
    return m


# ----------------------------------------------------------------- materials
def srgb_to_linear(c):
    """UnityEngine.Color.linear."""
    c = np.asarray(c, np.float64)
    return np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)


def material_data(color=(1, 1, 1, 1), metallic=0.0, glossiness=0.5, mode=0, emission=None,
                  material_type=0, ior=1.0, clear_coat_color=(0, 0, 0)):
    """MaterialData(Material), BVH.cs:38-76: override order type -> transparent -> emission."""
    m = np.zeros((), wire.MATERIAL)
    lin = srgb_to_linear(color[:3])
    m["albedo"] = lin.astype(F32)
    m["metallic"] = max(0.01, metallic)
    m["smoothness"] = max(0.01, glossiness)
    m["transparent"] = -1
    m["materialType"] = int(material_type)
    m["ior"] = ior
    if int(material_type) == 3:
        m["clearCoatColor"] = np.asarray(clear_coat_color, F32)
    if int(mode) == 3:
        m["transparent"] = color[3]            # alpha is not gamma-encoded
        m["materialType"] = 1
    if emission is not None:
        m["emissionColor"] = np.asarray(emission, F32)
        m["materialType"] = 2
    return m


# ----------------------------------------------------------------- scene container
@dataclass
class Primitives:
    """The buffers MeshCollector.Init uploads (MeshCollector.cs:170-200)."""
    planes: np.ndarray = field(default_factory=lambda: np.zeros(0, wire.PLANE))
    spheres: np.ndarray = field(default_factory=lambda: np.zeros(0, wire.SPHERE))
    cubes: np.ndarray = field(default_factory=lambda: np.zeros(0, wire.CUBE))
    meshes: np.ndarray = field(default_factory=lambda: np.zeros(0, wire.CUSTOM_MESH))
    vertices: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), F32))
    normals: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), F32))
    indices: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))

    @property
    def n_triangles(self):
        return int(self.meshes["indicesCount"].sum()) // 3 if len(self.meshes) else 0


@dataclass
class Scene:
    name: str
    vertices: np.ndarray          # MESH_VERTEX[V]
    indices: np.ndarray           # u32[3N]
    material_index: np.ndarray    # u32[N]
    shadow_index: np.ndarray      # i32[N,2]
    materials: np.ndarray         # MATERIAL[M]
    sphere_lights: np.ndarray = field(default_factory=lambda: np.zeros(0, wire.SPHERE_LIGHT))
    area_lights: np.ndarray = field(default_factory=lambda: np.zeros(0, wire.AREA_LIGHT))
    disc_lights: np.ndarray = field(default_factory=lambda: np.zeros(0, wire.DISC_LIGHT))
    skybox: np.ndarray = None     # f32[6,S,S,4] or None
    params: wire.RtParams = None
    notes: str = ""
    primitives: Primitives = None  # brute-force lists (useBVH = false), or None

    @property
    def n_triangles(self):
        return len(self.material_index)


class SceneBuilder:
    """Merges meshes the way BVH.InitMesh does (BVH.cs:229-271)."""

    def __init__(self):
        self.v, self.i, self.mi, self.sh, self.mats = [], [], [], [], []
        self.offset = 0

    def add(self, mesh, material, position=(0, 0, 0), rotation=IDENTITY_Q, scale=(1, 1, 1),
            cast_shadow=True, receive_shadow=True, matrix=None):
        """matrix: a 4x4 localToWorld of a transform hierarchy, used instead of position / rotation / scale."""
        if matrix is not None:
            m = np.asarray(matrix, np.float64)
            pos = mesh.positions.astype(np.float64) @ m[:3, :3].T + m[:3, 3]   # TransformPoint
            nrm = mesh.normals.astype(np.float64) @ m[:3, :3].T                # TransformVector: scaled, not renormalised
            tan = mesh.tangents.astype(np.float64) @ m[:3, :3].T
        else:
            s = np.asarray(scale, np.float64)
            pos = quat_rotate(rotation, mesh.positions.astype(np.float64) * s) + np.asarray(position, np.float64)
            nrm = quat_rotate(rotation, mesh.normals.astype(np.float64) * s)      # TransformVector: scaled
            tan = quat_rotate(rotation, mesh.tangents.astype(np.float64) * s)
        v = np.zeros(len(pos), wire.MESH_VERTEX)
        v["position"], v["normal"], v["tangent"], v["uv"] = pos.astype(F32), nrm.astype(F32), tan.astype(F32), mesh.uvs
        self.v.append(v)
        self.i.append((mesh.triangles.astype(np.uint32) + np.uint32(self.offset)).reshape(-1))
        self.offset += len(pos)
        idx = None
        for k, m in enumerate(self.mats):
            if m is material:
                idx = k
        if idx is None:
            self.mats.append(material)
            idx = len(self.mats) - 1
        nt = len(mesh.triangles)
        self.mi.append(np.full(nt, idx, np.uint32))
        self.sh.append(np.tile(np.array([[int(cast_shadow), int(receive_shadow)]], np.int32), (nt, 1)))
        return self

    def build(self, name, **kw):
        return Scene(name, np.concatenate(self.v), np.concatenate(self.i), np.concatenate(self.mi),
                     np.concatenate(self.sh), np.array(self.mats, wire.MATERIAL), **kw)


# ----------------------------------------------------------------- camera / params
def camera_to_world(position, rotation=IDENTITY_Q):
    """Camera.cameraToWorldMatrix: TRS(position, rotation, 1) * scale(1,1,-1)
    (OpenGL convention, camera looks down -Z), as 16 floats in Unity memory
    order (column-major)."""
    m = np.eye(4)
    for c in range(3):
        e = np.zeros(3)
        e[c] = 1.0
        m[:3, c] = quat_rotate(rotation, e)
    m[:3, 2] *= -1.0
    m[:3, 3] = position
    return m.T.reshape(-1).astype(F32)        # column-major


def inverse_projection(fov_deg, aspect, near=0.3, far=1000.0):
    """Camera.projectionMatrix.inverse (OpenGL-style perspective), column-major."""
    f = 1.0 / np.tan(np.deg2rad(fov_deg) / 2)
    p = np.zeros((4, 4))
    p[0, 0] = f / aspect
    p[1, 1] = f
    p[2, 2] = -(far + near) / (far - near)
    p[2, 3] = -2 * far * near / (far - near)
    p[3, 2] = -1
    return np.linalg.inv(p).T.reshape(-1).astype(F32)


def make_params(width, height, cam_pos, cam_rot=IDENTITY_Q, fov=50.0, sampling=wire.SAMPLING_MIS,
                light_dir=(0.3, -0.8, 0.5), light_radius=0.1, light_color=(1, 1, 1), sky_intensity=1.0,
                shadow_color=(0, 0, 0), shadow_intensity=1.0, sky_rotation=0.0, sky_exposure=1.0,
                lens_radius=0.0, focal_length=1.0, flags=0, seed=0.25):
    p = wire.RtParams()
    p.width, p.height = width, height
    p.camera2World[:] = camera_to_world(cam_pos, cam_rot).tolist()
    p.cameraInverseProjection[:] = inverse_projection(fov, width / height).tolist()
    p.seed = float(seed)
    ld = np.asarray(light_dir, np.float64)
    ld = ld / np.linalg.norm(ld)
    p.lightParameter[:] = [float(F32(ld[0])), float(F32(ld[1])), float(F32(ld[2])), light_radius]
    p.lightColor[:] = [*light_color, sky_intensity]
    p.shadowParameter[:] = [*shadow_color, shadow_intensity]
    p.skyboxRotation, p.skyboxExposure = sky_rotation, sky_exposure
    p.lensRadius, p.focalLength = lens_radius, focal_length
    p.flags = int(sampling) | int(flags)
    p.tileRank, p.tileCount, p.tileSize = 0, 1, 64
    return p


def area_light(position, rotation, box_size_xy, local_scale):
    """AreaLight(Transform), LightImportanceSampling.cs:27-42."""
    a = np.zeros((), wire.AREA_LIGHT)
    a["position"] = np.asarray(position, F32)
    a["normal"] = (-quat_rotate(rotation, np.array([0.0, 0, 1]))).astype(F32)
    a["up"] = quat_rotate(rotation, np.array([0.0, 1, 0])).astype(F32)
    a["size"] = (F32(box_size_xy[0]) * F32(local_scale[0]), F32(box_size_xy[1]) * F32(local_scale[1]))
    return a


def sphere_light(position, collider_radius, local_scale_x):
    s = np.zeros((), wire.SPHERE_LIGHT)
    s["position"] = np.asarray(position, F32)
    s["radius"] = F32(collider_radius) * F32(local_scale_x)
    return s


# ----------------------------------------------------------------- sky
def procedural_skybox(size=64, sun_dir=(0.4, 0.6, -0.5), sun_cos=0.995, sun_radiance=20.0):
    """Analytic HDR sky (vertical gradient + sun disc) rasterised to a cubemap in
    D3D face order +X,-X,+Y,-Y,+Z,-Z.  Stand-in for the reference's missing HDRIs."""
    sd = np.asarray(sun_dir, np.float64)
    sd /= np.linalg.norm(sd)
    faces = np.zeros((6, size, size, 4), F32)
    c = (np.arange(size) + 0.5) / size * 2 - 1
    sc, tc = np.meshgrid(c, c)          # sc along x (columns), tc along y (rows)
    one = np.ones_like(sc)
    dirs = [(one, -tc, -sc), (-one, -tc, sc), (sc, one, tc), (sc, -one, -tc), (sc, -tc, one), (-sc, -tc, -one)]
    for f, (x, y, z) in enumerate(dirs):
        d = np.stack([x, y, z], -1)
        d /= np.linalg.norm(d, axis=-1, keepdims=True)
        horizon = np.array([0.75, 0.80, 0.85])
        zenith = np.array([0.15, 0.35, 0.80])
        ground = np.array([0.25, 0.22, 0.20])
        col = np.where(d[..., 1:2] >= 0, horizon + (zenith - horizon) * (np.maximum(d[..., 1:2], 0) ** 0.5),
                       horizon + (ground - horizon) * np.minimum(1, -d[..., 1:2] * 4))
        sun = (d @ sd) > sun_cos
        col = np.where(sun[..., None], sun_radiance, col)
        faces[f, ..., :3] = col
        faces[f, ..., 3] = 1
    return faces


# ----------------------------------------------------------------- scenes
def cornell_box(width=256, height=256, sphere_segments=64, with_sky=True, sampling=wire.SAMPLING_MIS):
    """Cornell box after Assets/Scenes/Chapter6_3.unity: five quads (floor,
    ceiling, red right wall, green left wall, slightly tilted back wall), a 6x6
    emissive quad at y = 14.97 with a matching area light, a glass sphere and a
    mirror sphere.  The scene's roundBox / teapot prefabs (binary FBX) are
    replaced by two axis-aligned boxes; sphere64.fbx by a generated UV sphere."""
    grey = material_data((0.6792453, 0.6792453, 0.6792453, 1), 0, 0.5)
    red = material_data((0.9056604, 0, 0, 1), 0, 0.5)
    green = material_data((0.037291814, 0.6792453, 0, 1), 0, 0.5)
    emission = material_data((1, 1, 1, 1), 0, 0.5, emission=(1, 1, 1))
    glass = material_data((0.6792453, 0.6792453, 0.6792453, 0.20784314), 0, 0.867, mode=3, ior=1.1)
    mirror = material_data((0.6792453, 0.6792453, 0.6792453, 1), 1, 1)
    box_mat = material_data((0.8, 0.55, 0.2, 1), 0.2, 0.6)
    s = 0.7071068
    q = quad_mesh()
    sph = uv_sphere_mesh(sphere_segments, sphere_segments // 2)
    b = SceneBuilder()
    b.add(q, grey, (0, 0, 0), (s, 0, 0, s), (20, 20, 1), cast_shadow=False)            # Bottom
    b.add(q, red, (10, 7.5, 0), (0, s, 0, s), (20, 15, 1), cast_shadow=False)          # Right
    b.add(q, grey, (0, 15, 0), (-s, 0, 0, s), (20, 20, 1), cast_shadow=False)          # Top
    b.add(q, green, (-10, 7.5, 0), (0, -s, 0, s), (20, 15, 1), cast_shadow=False)      # Left
    b.add(q, grey, (0, 7.5, 10), (0.00008726013, 0, 0, 1), (20, 15, 1), cast_shadow=False)  # Back
    b.add(q, emission, (0, 14.97, 0), (-s, 0, 0, s), (6, 6, 6))                        # AreaLight quad
    b.add(sph, glass, (2.81, 1.95, -7.79), IDENTITY_Q, (1.8472177,) * 3)               # Sphere (1)
    b.add(sph, mirror, (5.608, 8.01, -3.67), IDENTITY_Q, (3, 3, 3))                    # Sphere
    bx = box_mesh()
    b.add(bx, box_mat, (-4.5, 3.0, 2.0), quat_from_euler_deg(0, 25, 0), (5, 6, 5))     # stand-in for roundBox
    b.add(bx, grey, (-2.52, 1.25, -5.0), quat_from_euler_deg(0, -20, 0), (2.5, 2.5, 2.5))  # stand-in for teapot
    al = area_light((0, 14.97, 0), (-s, 0, 0, s), (1, 1.0000002), (6, 6, 6))
    sc = b.build("cornell", area_lights=np.array([al], wire.AREA_LIGHT),
                 skybox=procedural_skybox(32) if with_sky else None,
                 notes="Chapter6_3 stand-in: generated spheres, boxes for roundBox/teapot, procedural sky")
    sc.params = make_params(width, height, (0, 5.8, -21.2), fov=50.0, sampling=sampling, light_radius=0.1,
                            sky_intensity=1.0)
    return sc


def pbr_sphere_grid(width=1920, height=1080, sphere_segments=64):
    """6x6 grid of spheres over a quad: rows are the six presets of
    Assets/Scripts/RayTracing/CustomMaterialInfo.cs:30-74, columns sweep i/5."""
    b = SceneBuilder()
    sph = uv_sphere_mesh(sphere_segments, sphere_segments)
    floor = material_data((0.5, 0.5, 0.5, 1), 0.0, 0.3)
    b.add(quad_mesh(), floor, (0, 0, 0), (0.7071068, 0, 0, 0.7071068), (60, 60, 1), cast_shadow=False)
    base = (0.8, 0.3, 0.2)
    for row in range(6):
        for i in range(6):
            t = i / 5.0
            if row == 0:
                m = material_data((*base, 1), 0.0, t)
            elif row == 1:
                m = material_data((*base, 1), 1.0, t)
            elif row == 2:
                m = material_data((*base, 0.1), 0.0, 0.95, mode=3, ior=1.0 + 0.5 * t)
            elif row == 3:
                m = material_data((*base, 1), 0.0, 0.9, material_type=3, ior=1.0 + 0.5 * t, clear_coat_color=(1, 1, 1))
            elif row == 4:
                m = material_data((*base, t), 0.0, 0.95, mode=3, ior=1.1)
            else:
                m = material_data((*base, 1), t, 0.8)
            b.add(sph, m, ((i - 2.5) * 3.0, 1.3, (row - 2.5) * 3.0), IDENTITY_Q, (1.25,) * 3)
    sl = sphere_light((0, 12, -4), 0.5, 3.0)
    sc = b.build("pbr_grid", sphere_lights=np.array([sl], wire.SPHERE_LIGHT), skybox=procedural_skybox(128),
                 notes="Chapter6_5 stand-in: procedural HDRI, generated spheres")
    sc.params = make_params(width, height, (0, 9.0, -16.0), quat_from_euler_deg(28, 0, 0), fov=50.0,
                            sky_intensity=2.0)
    return sc


def _displaced_ellipsoid(rng, n_lon, n_lat, radii, bump):
    m = uv_sphere_mesh(n_lon, n_lat)
    p = m.positions.astype(np.float64)
    ph = rng.floats(6).astype(np.float64) * 6.28
    disp = 1 + bump * (np.sin(p[:, 0] * 5 + ph[0]) * np.sin(p[:, 1] * 7 + ph[1]) * np.sin(p[:, 2] * 3 + ph[2]))
    m.positions = (p * disp[:, None] * np.asarray(radii)).astype(F32)
    n = m.normals.astype(np.float64) / np.asarray(radii)
    m.normals = (n / np.linalg.norm(n, axis=1, keepdims=True)).astype(F32)
    return _ensure_outward(m)


def car_standin(width=1920, height=1080, detail=1.0, sampling=wire.SAMPLING_MIS, dof=True):
    """Procedural ~500k-triangle 'car' (the Alfa Romeo FBX is absent from the
    reference checkout): displaced ellipsoids for body, cabin, wheels, lamps and
    trim with 18 materials spanning types 0/1/2/3, on a matte-mask ground quad,
    lit by a procedural HDRI.  Camera FOV 50 as Assets/Scenes/Chapter7_6.unity."""
    rng = PCG32(4)
    b = SceneBuilder()
    mats = []
    paint = material_data((0.7, 0.05, 0.05, 1), 0.3, 0.9, material_type=3, ior=1.5, clear_coat_color=(1, 1, 1))
    glass = material_data((0.8, 0.9, 0.9, 0.15), 0.0, 0.98, mode=3, ior=1.45)
    chrome = material_data((0.9, 0.9, 0.9, 1), 1.0, 0.95)
    rubber = material_data((0.03, 0.03, 0.03, 1), 0.0, 0.2)
    lamp = material_data((1, 1, 1, 1), 0, 0.5, emission=(4, 3.6, 3.0))
    tail = material_data((1, 0, 0, 1), 0, 0.5, emission=(2.0, 0.1, 0.1))
    mats += [paint, glass, chrome, rubber, lamp, tail]
    for k in range(12):
        c = rng.floats(3)
        mats.append(material_data((float(c[0]), float(c[1]), float(c[2]), 1), float(rng.next_float()),
                                  float(0.2 + 0.7 * rng.next_float())))
    ground = material_data((0.5, 0.5, 0.5, 1), 0, 0.5, material_type=4)
    s = 0.7071068
    S = 0.35                    # world scale: the focal plane (focalLength 2.0) passes through the car
    b.add(quad_mesh(), ground, (0, 0, 0), (s, 0, 0, s), (80 * S, 80 * S, 1), cast_shadow=False)
    d = max(0.05, detail)
    L = lambda n: max(8, int(n * np.sqrt(d)))
    sc3 = lambda r: tuple(S * x for x in r)
    b.add(_displaced_ellipsoid(rng, L(420), L(210), sc3((2.3, 0.62, 1.0)), 0.015), paint, sc3((0, 0.85, 0)))      # body
    b.add(_displaced_ellipsoid(rng, L(300), L(150), sc3((1.15, 0.5, 0.82)), 0.0), glass, sc3((-0.25, 1.38, 0)))    # cabin
    for sx in (-1.45, 1.45):
        for sz in (-0.95, 0.95):
            b.add(_displaced_ellipsoid(rng, L(180), L(90), sc3((0.42, 0.42, 0.16)), 0.03), rubber, sc3((sx, 0.42, sz)))
            b.add(_displaced_ellipsoid(rng, L(90), L(45), sc3((0.25, 0.25, 0.18)), 0.0), chrome, sc3((sx, 0.42, sz)))
    for sz in (-0.6, 0.6):
        b.add(_displaced_ellipsoid(rng, L(60), L(30), sc3((0.12, 0.1, 0.18)), 0.0), lamp, sc3((2.2, 0.85, sz)))
        b.add(_displaced_ellipsoid(rng, L(60), L(30), sc3((0.08, 0.08, 0.2)), 0.0), tail, sc3((-2.25, 0.9, sz)))
    for k in range(12):
        r = rng.floats(5).astype(np.float64)
        b.add(_displaced_ellipsoid(rng, L(70), L(35), sc3((0.1 + 0.15 * r[0], 0.05 + 0.05 * r[1], 0.1 + 0.15 * r[2])), 0.05),
              mats[6 + k], sc3(((r[3] - 0.5) * 4.2, 0.5 + 0.8 * r[4], (1.0 if k % 2 else -1.0) * (0.85 + 0.1 * r[0]))))
    sl = sphere_light(sc3((3, 6, -3)), 0.5, 2.0 * S)
    scn = b.build("car", sphere_lights=np.array([sl], wire.SPHERE_LIGHT), skybox=procedural_skybox(256),
                  notes="procedural stand-in for Car_AlfaRomeo_8C_2007 + procedural HDRI")
    flags = wire.FLAG_MATTE_MASK | ((wire.FLAG_DEPTH_OF_FIELD | wire.FLAG_DOF_PRIMARY) if dof else 0)
    focal = 2.0
    cam_rot = quat_from_euler_deg(14, -40, 0)
    target = np.array(sc3((0.1, 0.8, 0)))
    cam_pos = target - quat_rotate(cam_rot, np.array([0.0, 0.0, 1.0])) * focal
    scn.params = make_params(width, height, tuple(cam_pos), cam_rot, fov=50.0,
                             sampling=sampling, light_radius=0.2, sky_intensity=4.0, flags=flags,
                             lens_radius=0.3 if dof else 0.0, focal_length=focal if dof else 1.0,
                             shadow_intensity=0.8)
    return scn


def _surface(rec, color, metallic, glossiness):
    rec["albedo"] = srgb_to_linear(color[:3]).astype(F32)
    rec["metallic"], rec["smoothness"], rec["transparent"] = max(0.01, metallic), max(0.01, glossiness), -1
    rec["emissionColor"] = 0


def primitive_room(width=256, height=256, sampling=wire.SAMPLING_MIS, mesh_segments=12, seed=3):
    """A chapter-3/4 style scene for the brute-force path (useBVH = false): a floor and a
    back wall (planes), a row of spheres of varying metallic / smoothness, a few cubes and two
    custom meshes (a tessellated sphere under a non-uniform TRS and a tilted box), lit by one
    area light, one sphere light and the procedural sky.  Lists are in far-to-near order as
    MeshCollector.InitPrimitive sorts them."""
    cam = np.array([0.0, 2.0, -9.0])
    rng = np.random.default_rng(seed)
    planes = np.zeros(2, wire.PLANE)
    planes[0]["normal"], planes[0]["position"], planes[0]["size"] = (0, 0, -1), (0, 3, 6), (16, 8, 1)
    _surface(planes[0:1], (0.55, 0.6, 0.7, 1), 0.0, 0.3)
    planes[1]["normal"], planes[1]["position"], planes[1]["size"] = (0, 1, 0), (0, 0, 0), (16, 1, 16)
    _surface(planes[1:2], (0.7, 0.7, 0.7, 1), 0.0, 0.5)
    spheres = np.zeros(5, wire.SPHERE)
    for k in range(5):
        spheres[k]["center"], spheres[k]["radius"] = (-4 + 2 * k, 0.8, 1.5 + 0.3 * k), 0.8
        _surface(spheres[k:k + 1], (0.9, 0.3 + 0.1 * k, 0.2, 1), k / 4, 0.2 + 0.2 * k)
    cubes = np.zeros(3, wire.CUBE)
    for k in range(3):
        c = np.array([-3 + 3 * k, 0.5, -1.5])
        cubes[k]["min"], cubes[k]["max"] = c - 0.5, c + 0.5
        _surface(cubes[k:k + 1], (0.2, 0.5 + 0.2 * k, 0.9, 1), 0.1, 0.6)
    order = lambda a, key: a[np.argsort(-np.linalg.norm(a[key] - cam, axis=1), kind="stable")]
    planes, spheres = order(planes, "position"), order(spheres, "center")
    cubes = cubes[np.argsort(-np.linalg.norm((cubes["min"] + cubes["max"]) / 2 - cam, axis=1), kind="stable")]
    ball, box = uv_sphere_mesh(mesh_segments, max(mesh_segments // 2, 2)), box_mesh()
    meshes = np.zeros(2, wire.CUSTOM_MESH)
    verts = np.concatenate([ball.positions, box.positions]).astype(F32)
    norms = np.concatenate([ball.normals, box.normals]).astype(F32)
    idx = np.concatenate([ball.triangles.reshape(-1), box.triangles.reshape(-1) + len(ball.positions)]).astype(np.int32)
    m0 = trs_matrix((1.0, 2.6, 2.0), quat_from_euler_deg(20, 30, 0), (1.4, 0.7, 1.0))
    m1 = trs_matrix((-2.5, 2.2, 3.0), quat_from_euler_deg(25, 40, 10), (1.2, 1.2, 1.2))
    meshes[0]["localToWorldMatrix"], meshes[1]["localToWorldMatrix"] = m0.T.reshape(-1), m1.T.reshape(-1)
    meshes[0]["indicesOffset"], meshes[0]["indicesCount"] = 0, ball.triangles.size
    meshes[1]["indicesOffset"], meshes[1]["indicesCount"] = ball.triangles.size, box.triangles.size
    _surface(meshes[0:1], (0.9, 0.8, 0.3, 1), 0.8, 0.7)
    _surface(meshes[1:2], (0.4, 0.8, 0.4, 1), 0.0, 0.4)
    del rng
    prims = Primitives(planes, spheres, cubes, meshes, verts, norms, idx)
    sc = Scene("primitive_room", np.zeros(0, wire.MESH_VERTEX), np.zeros(0, np.uint32), np.zeros(0, np.uint32),
               np.zeros((0, 2), np.int32), np.zeros(0, wire.MATERIAL), primitives=prims)
    sc.area_lights = np.array([area_light((0, 6.5, 1.0), quat_from_euler_deg(90, 0, 0), (1, 1), (4, 4, 1))], wire.AREA_LIGHT)
    sc.sphere_lights = np.array([sphere_light((-5, 5, -2), 0.5, 1.2)], wire.SPHERE_LIGHT)
    sc.skybox = procedural_skybox(32)
    sc.params = make_params(width, height, cam, quat_from_euler_deg(5, 0, 0), fov=50.0, sampling=sampling,
                            flags=wire.FLAG_BRUTE_FORCE, sky_intensity=2.0)
    return sc


def primitive_rays(n, seed=11):
    """Rays for first-hit checks of the primitive room: origins around the camera side, aimed into the room,
    plus some from inside / behind the primitives (exercise the unconditional sphere and cube writes)."""
    rng = np.random.default_rng(seed)
    o = np.stack([rng.uniform(-7, 7, n), rng.uniform(0.2, 6, n), rng.uniform(-9, 5, n)], 1)
    target = np.stack([rng.uniform(-6, 6, n), rng.uniform(0, 4, n), rng.uniform(-2, 6, n)], 1)
    d = target - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o.astype(F32), d.astype(F32)


def triangle_soup(n_triangles, seed=5, width=256, height=256, extent=100.0, edge=0.5):
    """Random triangle soup: centres uniform in [-extent,extent]^3, vertex offsets
    uniform in [-edge,edge]^3, one diffuse material."""
    rs = np.random.Generator(np.random.PCG64(seed))
    c = rs.uniform(-extent, extent, (n_triangles, 1, 3))
    e = rs.uniform(-edge, edge, (n_triangles, 3, 3))
    p = (c + e).astype(F32).reshape(-1, 3)
    v = np.zeros(len(p), wire.MESH_VERTEX)
    v["position"] = p
    fn = np.cross(e[:, 2] - e[:, 0], e[:, 1] - e[:, 0])
    fn /= np.maximum(np.linalg.norm(fn, axis=1, keepdims=True), 1e-20)
    v["normal"] = np.repeat(fn, 3, axis=0).astype(F32)
    v["tangent"] = (1, 0, 0)
    idx = np.arange(3 * n_triangles, dtype=np.uint32)
    mats = np.array([material_data((0.7, 0.7, 0.7, 1), 0, 0.3)], wire.MATERIAL)
    sc = Scene("soup", v, idx, np.zeros(n_triangles, np.uint32), np.ones((n_triangles, 2), np.int32), mats,
               skybox=procedural_skybox(16), notes="random triangle soup")
    sc.params = make_params(width, height, (0, 0, -260), fov=50.0, sampling=wire.SAMPLING_COSINE)
    return sc


def stress_rays(n, seed=7, extent=100.0):
    """Incoherent rays: origins uniform in the soup box, cosine-hemisphere
    directions about random normals."""
    rs = np.random.Generator(np.random.PCG64(seed))
    o = rs.uniform(-extent, extent, (n, 3)).astype(F32)
    nrm = rs.normal(size=(n, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    u1, u2 = rs.uniform(size=n), rs.uniform(size=n)
    r, ph = np.sqrt(u1), 2 * np.pi * u2
    helper = np.where(np.abs(nrm[:, :1]) > 0.9, np.array([[0, 1.0, 0]]), np.array([[1.0, 0, 0]]))
    t = np.cross(nrm, helper)
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    bt = np.cross(nrm, t)
    d = t * (r * np.cos(ph))[:, None] + bt * (r * np.sin(ph))[:, None] + nrm * np.sqrt(1 - u1)[:, None]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d.astype(F32)
