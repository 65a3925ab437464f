"""Host-side mirror of the reference's C# driver classes, re-pointed at the C ABI.

The reference host is C# (Unity MonoBehaviours) and no C# toolchain exists in
this build environment, so the same surface is provided here in Python with the
reference's class, field and method names; `csharp/RtCoreNative.cs` shows the
P/Invoke form of the identical calls.  Mirrored:

  RayTracing              Assets/Scripts/RayTracing/RayTracing.cs      Start, Update, CustomRender, SetDirty, OnDisable
  BVH                     Assets/Scripts/BVH/BVH.cs:108-352            Init, InitMesh, InitMaterialData, UpdateMaterialData
  LightImportanceSampling Assets/Scripts/RayTracing/LightImportanceSampling.cs:57-107   Init
  DepthOfField            Assets/Scripts/PostProcess/DepthOfField.cs:22-44  UpdateParameter
  PostProcessStack, ColorCorrection, ColorGrading, Bloom, PostProcessBase   Assets/Scripts/PostProcess/*.cs
  SaveTexture             Assets/Scripts/SaveTexture.cs:15-90          Save (SavePNG)
  MeshCollector           Assets/Scripts/RayTracing/MeshCollector.cs:16-307   Init, InitPrimitive, InitMaterial

Unity objects (Transform, MeshRenderer, Material, Camera) are replaced by small
data classes holding just the fields those methods read.
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import scenes, wire
from .native import RtCore


# ------------------------------------------------------------------ stand-ins for Unity objects
@dataclass
class Transform:
    position: tuple = (0.0, 0.0, 0.0)
    rotation: tuple = scenes.IDENTITY_Q
    localScale: tuple = (1.0, 1.0, 1.0)
    matrix: Optional[np.ndarray] = None           # localToWorldMatrix of a transform with parents (overrides the three above)

    @property
    def forward(self):
        return scenes.quat_rotate(self.rotation, np.array([0.0, 0.0, 1.0]))

    @property
    def up(self):
        return scenes.quat_rotate(self.rotation, np.array([0.0, 1.0, 0.0]))


@dataclass(eq=False)
class Material:
    """The shader properties MaterialData(Material) reads (BVH.cs:38-76)."""
    color: tuple = (1.0, 1.0, 1.0, 1.0)          # sRGB, as in the .mat file
    _Metallic: float = 0.0
    _Glossiness: float = 0.5
    _Mode: int = 0
    _MaterialType: Optional[float] = None
    _IOR: Optional[float] = None
    _ClearCoatColor: Optional[tuple] = None
    _EMISSION: bool = False
    _EmissionColor: tuple = (0.0, 0.0, 0.0)

    def MaterialData(self):
        return scenes.material_data(self.color, self._Metallic, self._Glossiness, self._Mode,
                                    self._EmissionColor if self._EMISSION else None,
                                    0 if self._MaterialType is None else int(self._MaterialType),
                                    1.0 if self._IOR is None else self._IOR,
                                    (0, 0, 0) if self._ClearCoatColor is None else self._ClearCoatColor[:3])


@dataclass
class MeshRenderer:
    sharedMesh: scenes.Mesh
    sharedMaterial: Material
    transform: Transform = field(default_factory=Transform)
    enabled: bool = True
    activeInHierarchy: bool = True
    shadowCastingOn: bool = True
    receiveShadows: bool = True


@dataclass
class LightObject:
    """A child of sphereLightParent / areaLightParent / discLightParent."""
    transform: Transform
    colliderRadius: float = 0.5                  # SphereCollider.radius
    colliderSize: tuple = (1.0, 1.0, 0.0)        # BoxCollider.size
    active: bool = True


@dataclass
class Camera:
    transform: Transform = field(default_factory=Transform)
    fieldOfView: float = 60.0
    nearClipPlane: float = 0.3
    farClipPlane: float = 1000.0

    def cameraToWorldMatrix(self):
        return scenes.camera_to_world(self.transform.position, self.transform.rotation)

    def inverseProjectionMatrix(self, aspect):
        return scenes.inverse_projection(self.fieldOfView, aspect, self.nearClipPlane, self.farClipPlane)


# ------------------------------------------------------------------ BVH.cs
class BVH:
    def __init__(self, meshParent: List[MeshRenderer]):
        self.meshParent = meshParent
        self.core: Optional[RtCore] = None
        self.materials: List[Material] = []
        self.mesh = None
        self.onDirty = None                      # RayTracing.SetDirty (static in the reference)

    def InitMesh(self):
        """BVH.cs:201-318: merge every enabled, active renderer into one world-space
        vertex/index buffer with per-triangle material index and shadow flags."""
        mrs = [mr for mr in self.meshParent if mr.enabled and mr.activeInHierarchy]
        b = scenes.SceneBuilder()
        for mr in mrs:
            b.add(mr.sharedMesh, mr.sharedMaterial, mr.transform.position, mr.transform.rotation, mr.transform.localScale,
                  cast_shadow=mr.shadowCastingOn, receive_shadow=mr.receiveShadows, matrix=mr.transform.matrix)
        materials = list(b.mats)
        self.mesh = (np.concatenate(b.v), np.concatenate(b.i), np.concatenate(b.mi), np.concatenate(b.sh))
        return self.mesh, materials

    def InitMaterialData(self, materials):
        self.materials = materials
        datas = np.array([m.MaterialData() for m in materials], wire.MATERIAL)
        self.core.materials_upload(datas)

    def Init(self, core: RtCore):
        """BVH.cs:108-199: upload mesh + materials, build, bind (binding is implicit here)."""
        self.core = core
        (v, i, mi, sh), materials = self.InitMesh()
        core.scene_upload(v, i, mi, sh)
        self.InitMaterialData(materials)
        core.bvh_build()

    def UpdateMaterialData(self):
        """BVH.cs:334-352: re-read every material, re-upload, mark dirty."""
        materials = []
        for mr in self.meshParent:
            if mr.enabled and mr.activeInHierarchy and not any(m is mr.sharedMaterial for m in materials):
                materials.append(mr.sharedMaterial)
        datas = np.array([m.MaterialData() for m in materials], wire.MATERIAL)
        self.core.materials_upload(datas)
        if self.onDirty:
            self.onDirty()                       # RayTracing.SetDirty() (BVH.cs:351)


class CustomMaterialInfo:
    """CustomMaterialInfo.cs:6-103: the six material presets of the PBR sphere row (README pbr0..pbr5).
    `renderers` = parent.GetComponentsInChildren<Renderer>(): objects named Sphere_<k>, whose ratio is k / 5."""

    def __init__(self, bvh: "BVH", renderers: List["MeshRenderer"]):
        self.bvh = bvh
        self.materials = [r.sharedMaterial for r in renderers]                       # :17
        self.ratios = [float(r.name.split("_")[1]) / 5.0 for r in renderers]          # :18

    def OnMaterialChanged(self, index: int):                                           # :21-76
        n, ratios = len(self.materials), self.ratios
        metallic, smoothness, transparent, ior, material_type = [0.0] * n, [0.0] * n, [0.0] * n, [0.0] * n, 0
        if index == 0:
            metallic, smoothness, transparent, ior, material_type = [0.0] * n, ratios, [-1.0] * n, [1.0] * n, 0
        elif index == 1:
            metallic, smoothness, transparent, ior, material_type = [1.0] * n, ratios, [-1.0] * n, [1.0] * n, 0
        elif index == 2:
            metallic, smoothness, transparent, ior, material_type = [0.0] * n, ratios, [0.1] * n, [1.1] * n, 1
        elif index == 3:
            metallic, smoothness, transparent, ior, material_type = [0.0] * n, [1.0] * n, [0.1] * n, [1.0 + v * 0.5 for v in ratios], 1
        elif index == 4:
            metallic, smoothness, transparent, ior, material_type = [1.0] * n, [1.0] * n, [-1.0] * n, [1.0 + v * 0.5 for v in ratios], 3
        elif index == 5:
            metallic, smoothness, transparent, ior, material_type = [0.0] * n, [1.0] * n, ratios, [1.1] * n, 1
        self.ChangeMaterial(metallic, smoothness, transparent, ior, material_type)

    def ChangeMaterial(self, metallic, smoothness, transparent, ior, material_type):  # :78-102
        for i, mat in enumerate(self.materials):
            mat._Metallic = max(0.01, metallic[i])
            mat._Glossiness = max(0.01, smoothness[i])
            mat._IOR = ior[i]
            mat._MaterialType = material_type
            if material_type == 1:
                mat._Mode = 3
                mat.color = (mat.color[0], mat.color[1], mat.color[2], transparent[i])
            else:
                mat._Mode = 0
        if self.bvh.core is not None:
            self.bvh.UpdateMaterialData()


# ------------------------------------------------------------------ MeshCollector.cs
class MeshCollector:
    """MeshCollector.cs: the brute-force primitive lists of chapters 2-4 (useBVH = false).
    Each parent is a list of MeshRenderer stand-ins; a plane / sphere / cube is described
    by its transform alone (MeshCollector.cs:59-66, 84-90, 107-115), a custom mesh by
    its sharedMesh and localToWorldMatrix (:133-155)."""

    def __init__(self, cam: Camera, planeParent=None, sphereParent=None, cubeParent=None, customMeshParent=None):
        self.cam = cam
        self.planeParent, self.sphereParent = planeParent, sphereParent
        self.cubeParent, self.customMeshParent = cubeParent, customMeshParent
        self.vertices, self.normals, self.indices = [], [], []

    @staticmethod
    def InitMaterial(r, rec):                    # :16-40
        mat = r.sharedMaterial
        rec["albedo"] = scenes.srgb_to_linear(mat.color[:3]).astype(np.float32)
        rec["metallic"] = max(0.01, mat._Metallic)
        rec["smoothness"] = max(0.01, mat._Glossiness)
        rec["transparent"] = mat.color[3] if int(mat._Mode) == 3 else -1
        rec["emissionColor"] = mat._EmissionColor[:3] if mat._EMISSION else (0, 0, 0)

    def _plane(self, r, rec):                    # :59-66
        rec["normal"] = r.transform.up
        rec["position"] = r.transform.position
        rec["size"] = r.transform.localScale

    def _sphere(self, r, rec):                   # :84-90
        rec["center"] = r.transform.position
        rec["radius"] = np.float32(r.transform.localScale[0]) / np.float32(2)

    def _cube(self, r, rec):                     # :107-115
        pos = np.asarray(r.transform.position, np.float32)
        half = np.asarray(r.transform.localScale, np.float32) / np.float32(2)
        rec["min"], rec["max"] = pos - half, pos + half

    def _custom_mesh(self, r, rec):              # :133-155
        mesh = r.sharedMesh
        firstVertex = sum(len(v) for v in self.vertices)
        self.vertices.append(np.asarray(mesh.positions, np.float32))
        self.normals.append(np.asarray(mesh.normals, np.float32))
        firstIndex = sum(len(i) for i in self.indices)
        cur = np.asarray(mesh.triangles, np.int32).reshape(-1)
        self.indices.append(cur + np.int32(firstVertex))
        m = r.transform.matrix if r.transform.matrix is not None else scenes.trs_matrix(
            r.transform.position, r.transform.rotation, r.transform.localScale)
        rec["localToWorldMatrix"] = np.asarray(m, np.float32).T.reshape(-1)      # Unity memory order: column-major
        rec["indicesOffset"], rec["indicesCount"] = firstIndex, len(cur)

    def InitPrimitive(self, parent, dtype, init):           # :271-307
        rs = [r for r in (parent or []) if r.activeInHierarchy]
        cam = np.asarray(self.cam.transform.position, np.float64)
        dist = [float(np.linalg.norm(np.asarray(r.transform.position, np.float64) - cam)) for r in rs]
        order = sorted(range(len(rs)), key=lambda k: -dist[k])               # OrderByDescending is stable
        out = np.zeros(len(rs), dtype)
        for slot, k in enumerate(order):
            init(rs[k], out[slot:slot + 1])
            self.InitMaterial(rs[k], out[slot:slot + 1])
        if len(out) == 0:
            out = np.zeros(1, dtype)                                          # :299-300  one zeroed element
        return out

    def Primitives(self):
        planes = self.InitPrimitive(self.planeParent, wire.PLANE, self._plane)
        spheres = self.InitPrimitive(self.sphereParent, wire.SPHERE, self._sphere)
        cubes = self.InitPrimitive(self.cubeParent, wire.CUBE, self._cube)
        self.vertices, self.normals, self.indices = [], [], []               # :183-185
        meshes = self.InitPrimitive(self.customMeshParent, wire.CUSTOM_MESH, self._custom_mesh)
        cat = lambda parts, shape, dt: np.concatenate(parts).astype(dt) if parts else np.zeros(shape, dt)
        return scenes.Primitives(planes, spheres, cubes, meshes, cat(self.vertices, (0, 3), np.float32),
                                 cat(self.normals, (0, 3), np.float32), cat(self.indices, 0, np.int32))

    def Init(self, core: RtCore):                # :170-200
        core.primitives_upload(self.Primitives())


# ------------------------------------------------------------------ LightImportanceSampling.cs
class LightImportanceSampling:
    def __init__(self, sphereLightParent=None, areaLightParent=None, discLightParent=None):
        self.sphereLightParent = sphereLightParent or []
        self.areaLightParent = areaLightParent or []
        self.discLightParent = discLightParent or []

    def Lights(self):
        """(sphere, area, disc) light arrays in wire layout."""
        sph = [scenes.sphere_light(l.transform.position, l.colliderRadius, l.transform.localScale[0])
               for l in self.sphereLightParent if l.active]
        area = [scenes.area_light(l.transform.position, l.transform.rotation, l.colliderSize[:2], l.transform.localScale)
                for l in self.areaLightParent if l.active]
        disc = []
        for l in self.discLightParent:
            if l.active:
                d = np.zeros((), wire.DISC_LIGHT)
                d["position"] = np.asarray(l.transform.position, np.float32)
                d["normal"] = (-l.transform.forward).astype(np.float32)
                d["radius"] = np.float32(l.colliderRadius) * np.float32(l.transform.localScale[0])
                disc.append(d)
        return np.array(sph, wire.SPHERE_LIGHT), np.array(area, wire.AREA_LIGHT), np.array(disc, wire.DISC_LIGHT)

    def Init(self, core: RtCore):
        core.lights_upload(*self.Lights())


# ------------------------------------------------------------------ PostProcess/*.cs
class PostProcessBase:
    """PostProcessBase.cs:5-37.  `source` is an IMAGE_* selector of the core; the
    destination is always the core's IMAGE_POSTPROCESS (RayTracing.cs:156)."""
    enabled = True
    core: Optional[RtCore] = None

    def Init(self, core: RtCore):
        self.core = core

    def RenderImage(self, source):               # :29-32  Graphics.Blit(source, destination)
        self.core.post_blit(source)

    def UpdateParameter(self):
        pass


@dataclass
class DepthOfField(PostProcessBase):
    """DepthOfField.cs: only feeds _LensRadius/_FocalLength and the DEPTH_OF_FIELD
    keyword (:22-36); as a stage of the stack it is the base class's plain copy."""
    enabled: bool = False
    focalLength: float = 1.0
    lensRadius: float = 0.0


@dataclass
class ColorCorrection(PostProcessBase):          # ColorCorrection.cs:7-12
    brightness: float = 0.96
    saturation: float = 1.33
    contrast: float = 1.85
    enabled: bool = True

    def RenderImage(self, source):               # :22-29
        self.core.post_color_correction(source, self.brightness, self.saturation, self.contrast)


@dataclass
class ColorGrading(PostProcessBase):             # ColorGrading.cs:7
    lutTex: Optional[np.ndarray] = None          # f32[H, W, 4], row 0 = v 0
    enabled: bool = True

    def Init(self, core: RtCore):                # Start: mat.SetTexture("_LutTex", lutTex)  (:10-16)
        super().Init(core)
        if self.lutTex is not None:
            core.post_lut_upload(self.lutTex)

    def RenderImage(self, source):               # :26-29
        self.core.post_color_grading(source)


@dataclass
class Bloom(PostProcessBase):                    # Bloom.cs:7-29
    luminanceThreshold: float = 1.0
    downSampleStep: int = 7
    downSampleBlurSize: int = 5
    downSampleBlurSigma: float = 1.0
    upSampleBlurSize: int = 5
    upSampleBlurSigma: float = 1.0
    bloomIntensity: float = 1.0
    enabled: bool = True

    def Parameters(self):
        return wire.RtBloomParams(luminanceThreshold=self.luminanceThreshold, downSampleStep=self.downSampleStep,
                                  downSampleBlurSize=self.downSampleBlurSize, downSampleBlurSigma=self.downSampleBlurSigma,
                                  upSampleBlurSize=self.upSampleBlurSize, upSampleBlurSigma=self.upSampleBlurSigma,
                                  bloomIntensity=self.bloomIntensity)

    def RenderImage(self, source):               # :41-115
        self.core.post_bloom(source, self.Parameters())


class PostProcessStack:
    """PostProcessStack.cs.  The three delegates are multicast: Render hands the SAME
    source to every bound stage in binding order, each drawing over the destination
    (:29-35), so the destination ends up holding the last bound stage's output."""

    def __init__(self, colorCorrection=None, colorGrading=None, bloom=None, dof=None):
        self.colorCorrection, self.colorGrading, self.bloom, self.dof = colorCorrection, colorGrading, bloom, dof
        self.renderers, self.updates, self.inits = [], [], []

    def Init(self, core: RtCore):                # :21-27
        self.core = core
        self.Bind()
        for init in self.inits:
            init(core)

    def Render(self, source=wire.IMAGE_CONVERGED):   # :29-35
        if self.renderers:
            for render in self.renderers:
                render(source)
        else:
            self.core.post_blit(source)

    def UpdateParameter(self):                   # :37-41
        for update in self.updates:
            update()

    def Bind(self):                              # :43-72
        self.renderers, self.updates, self.inits = [], [], []
        for stage in (self.colorCorrection, self.colorGrading, self.bloom):
            if stage is not None and stage.enabled:
                self.inits.append(stage.Init)
                self.renderers.append(stage.RenderImage)
                self.updates.append(stage.UpdateParameter)
        if self.dof is not None:                 # :66 -- bound whether enabled or not
            self.inits.append(self.dof.Init)
            self.renderers.append(self.dof.RenderImage)
            self.updates.append(self.dof.UpdateParameter)


class SaveTexture:
    """SaveTexture.cs: the 8-bit sRGB image its SavePNG hands to EncodeToPNG (:23-38)."""

    def __init__(self, rayTracing):
        self.rayTracing = rayTracing

    def Pixels(self, which=wire.IMAGE_CONVERGED):
        return self.rayTracing.core.save_rgb24(which)        # u8[H, W, 3], row 0 = bottom

    def Save(self, path, which=wire.IMAGE_CONVERGED):         # :15-21 saves convergedRT
        from .ingest import png
        png.write_png(path, self.Pixels(which)[::-1])         # PNG rows run top to bottom
        return path


# ------------------------------------------------------------------ RayTracing.cs
class RayTracing:
    SamplingType = {"Uniform": 0, "Cosine": 1, "LightImportance": 2, "BSDFImportance": 3, "MultipleImportance": 4}

    def __init__(self, cam: Camera, bvh: Optional[BVH], lightImportanceSampling: LightImportanceSampling, width, height,
                 device_id=0, seed_source=None, meshCollector: Optional[MeshCollector] = None):
        self.cam, self.bvh, self.lightImportanceSampling = cam, bvh, lightImportanceSampling
        self.meshCollector = meshCollector
        self.width, self.height = width, height
        self.device_id = device_id
        # public fields of the MonoBehaviour (RayTracing.cs:20-65)
        self.skybox = None                       # skyboxMat _Tex: f32[6,S,S,4] or None
        self.skyboxRotation = 0.0
        self.skyboxExposure = 1.0
        self.mainLightForward = (0.0, -1.0, 0.0)
        self.mainLightColor = (1.0, 1.0, 1.0)
        self.skyboxIntensityMulitiply = 2.0
        self.shadowColor = (0.0, 0.0, 0.0)
        self.lightRadius = 0.1
        self.shadowIntensity = 1.0
        self.aliasing = True
        self.samplePrePixel = 0
        self.useBVH = True
        self.samplingType = "Uniform"
        self.matteMaskGO = None
        self.pause = False
        self.depthOfField = DepthOfField()
        self.postProcessStack: Optional[PostProcessStack] = None
        self.enablePostProcess = False
        self.isInitialized = False
        self.isRenderMatteMask = False
        self.core: Optional[RtCore] = None
        self._seeds = seed_source or scenes.PCG32(1)      # stands in for UnityEngine.Random.value
        self.params = None

    # RayTracing.cs:68-97
    def Start(self):
        self.core = RtCore(self.device_id)
        self.InitRT()
        self.InitShader()
        self.InitLight()
        self.InitSampling()
        if self.useBVH:                          # :84-93
            self.bvh.onDirty = self.SetDirty
            self.bvh.Init(self.core)
        else:
            self.meshCollector.Init(self.core)
        self.isInitialized = True

    def InitRT(self):                            # :172-187
        self.isRenderMatteMask = self.matteMaskGO is not None
        self.samplePrePixel = 0

    def InitShader(self):                        # :190-217
        self.core.skybox_upload(self.skybox)
        if self.enablePostProcess:               # :215-216
            self.postProcessStack.Init(self.core)

    def InitLight(self):                         # :254-282
        self.lightImportanceSampling.Init(self.core)

    def InitSampling(self):                      # :285-311
        if self.samplingType not in self.SamplingType:
            raise ValueError(self.samplingType)

    def UpdateParameters(self):                  # :219-239 (+ the Set* calls of InitShader/InitLight/InitSampling)
        self.params = self.BuildParameters()
        self.core.set_params(self.params)

    def BuildParameters(self):
        """Every uniform the frame needs as one RtParams (draws the frame's three Random.value)."""
        flags = self.SamplingType[self.samplingType]
        if self.depthOfField.enabled:
            flags |= wire.FLAG_DEPTH_OF_FIELD
        if self.isRenderMatteMask:
            flags |= wire.FLAG_MATTE_MASK
        if not self.useBVH:
            flags |= wire.FLAG_BRUTE_FORCE       # cs.DisableKeyword("BVH")
        p = wire.RtParams()
        p.width, p.height = self.width, self.height
        p.camera2World[:] = self.cam.cameraToWorldMatrix().tolist()
        p.cameraInverseProjection[:] = self.cam.inverseProjectionMatrix(self.width / self.height).tolist()
        self._seeds.next_float()                 # pixelOffset.x  (uploaded, unused by the shader)
        self._seeds.next_float()                 # pixelOffset.y
        p.seed = float(self._seeds.next_float())
        p.lightParameter[:] = [*[float(np.float32(c)) for c in self.mainLightForward], self.lightRadius]
        p.lightColor[:] = [*self.mainLightColor, self.skyboxIntensityMulitiply]
        p.shadowParameter[:] = [*self.shadowColor, self.shadowIntensity]
        p.skyboxRotation, p.skyboxExposure = self.skyboxRotation, self.skyboxExposure
        p.lensRadius, p.focalLength = self.depthOfField.lensRadius, self.depthOfField.focalLength
        p.flags = flags
        p.tileRank, p.tileCount, p.tileSize = 0, 1, 64
        return p

    # RayTracing.cs:118-164
    def CustomRender(self):
        if not self.isInitialized:
            return None                          # the reference returns silently
        self.UpdateParameters()
        if not self.pause:
            self.core.dispatch()
        if self.isRenderMatteMask and not self.pause:
            self.core.dispatch_shadowmap()
        if not self.aliasing:
            return self.core.readback(wire.IMAGE_RT)
        self.core.accumulate(self.samplePrePixel, self.isRenderMatteMask)
        if self.enablePostProcess:               # :154-158
            self.postProcessStack.Render(wire.IMAGE_CONVERGED)
        self.samplePrePixel += 1
        return None

    def convergedRT(self):
        return self.core.readback(wire.IMAGE_CONVERGED)

    def postProcessRT(self):
        return self.core.readback(wire.IMAGE_POSTPROCESS)

    def SetDirty(self):                          # :314-317
        self.samplePrePixel = 0
        if self.core:
            self.core.set_dirty()

    def OnDisable(self):                         # :319-334
        if self.core:
            self.core.close()
            self.core = None
        self.isInitialized = False
