#!/usr/bin/env python3
"""Render the arrays written by render_unity_scene.py --scene-npz and write an sRGB PNG.
Usage: render_npz.py scene.npz out.png [spp]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import native, scenes, wire  # noqa: E402
from raytracing_amd.ingest import png  # noqa: E402

z = np.load(sys.argv[1])
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
params = wire.RtParams.from_buffer_copy(z["params"].tobytes())
sc = scenes.Scene("ingested", z["vertices"], z["indices"], z["material_index"], z["shadow_index"], z["materials"],
                  z["sphere_lights"], z["area_lights"], z["disc_lights"], scenes.procedural_skybox(64), params)
core = native.RtCore(0)
core.upload(sc)
core.bvh_build()
core.set_params(params)
core.render_frames(scenes.frame_seeds(spp, 1))
img = core.readback(wire.IMAGE_CONVERGED)
png.write_png(sys.argv[2], png.linear_to_srgb8(img[::-1, :, :3]))
print(f"{sys.argv[2]}: {params.width}x{params.height}, {spp} spp, {sc.n_triangles} triangles, mean radiance {img[..., :3].mean():.4f}, "
      f"{core.stats()['renderMs'] / spp:.2f} ms per frame")
