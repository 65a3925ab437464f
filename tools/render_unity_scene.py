#!/usr/bin/env python3
"""Render a scene of a Unity project with the library and write an sRGB PNG.
Usage: render_unity_scene.py <project root> <scene.unity> <out.png> [--size WxH] [--spp N] [--scene-npz file]
                              [--post cc|bloom|grade:<lut.png>] [--save-texture]
--scene-npz: only convert (no GPU needed): write the merged arrays the library would be given.
--post: run one stage of the post-process stack on the converged image first.
--save-texture: 8-bit conversion exactly as SaveTexture.cs does it (linear value quantised before the sRGB curve)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import scenes, wire  # noqa: E402
from raytracing_amd.ingest import png, unity  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("project")
ap.add_argument("scene")
ap.add_argument("out")
ap.add_argument("--size", default="640x400")
ap.add_argument("--spp", type=int, default=64)
ap.add_argument("--scene-npz")
ap.add_argument("--post")
ap.add_argument("--save-texture", action="store_true")
args = ap.parse_args()
w, h = (int(v) for v in args.size.split("x"))
ls = unity.load_scene(args.scene, args.project, w, h)
for msg in ls.warnings:
    print("note:", msg)
rt = ls.rayTracing
if args.scene_npz:
    (v, i, mi, sh), mats = rt.bvh.InitMesh()
    sph, area, disc = rt.lightImportanceSampling.Lights()
    np.savez_compressed(args.scene_npz, vertices=v, indices=i, material_index=mi, shadow_index=sh,
                        materials=np.array([m.MaterialData() for m in mats], wire.MATERIAL),
                        sphere_lights=sph, area_lights=area, disc_lights=disc,
                        params=np.frombuffer(bytes(rt.BuildParameters()), np.uint8))
    print(f"{len(mi)} triangles, {len(mats)} materials, {len(sph)}/{len(area)}/{len(disc)} lights -> {args.scene_npz}")
    sys.exit(0)
rt.skybox = scenes.procedural_skybox(64)
rt.Start()
for _ in range(args.spp):
    rt.CustomRender()
img = rt.convergedRT()
which = wire.IMAGE_CONVERGED
if args.post:
    from raytracing_amd import host
    stage = (host.ColorCorrection() if args.post == "cc" else host.Bloom() if args.post == "bloom"
             else host.ColorGrading(lutTex=png.read_texture(args.post.split(":", 1)[1])))
    stack = host.PostProcessStack(stage if args.post == "cc" else None, stage if args.post.startswith("grade") else None,
                                  stage if args.post == "bloom" else None)
    stack.Init(rt.core)
    stack.Render(wire.IMAGE_CONVERGED)
    which = wire.IMAGE_POSTPROCESS
    img = rt.postProcessRT()
if args.save_texture:
    png.write_png(args.out, rt.core.save_rgb24(which)[::-1])
else:
    png.write_png(args.out, png.linear_to_srgb8(img[::-1, :, :3]))      # image row 0 is the bottom of the view
print(f"{args.out}: {w}x{h}, {args.spp} spp, mean radiance {img[..., :3].mean():.4f}")
rt.OnDisable()
