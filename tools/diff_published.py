#!/usr/bin/env python3
"""Map of 16x16-block differences between a render of a committed reference scene and the image the
reference's author published for it.  Usage: diff_published.py chapter6_2 lightsampling [spp] [out.png]
If the scene fixture carries the parameters of a Bloom component, that stage is applied first and the image is
converted the way SaveTexture does it; BLOOM=0 in the environment leaves the stage out."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import native, scenes, wire  # noqa: E402
from raytracing_amd.ingest import png  # noqa: E402

scene, image = sys.argv[1], sys.argv[2]
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 512
g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
z = np.load(os.path.join(g, f"unity_{scene}_scene.npz"))
want = np.load(os.path.join(g, f"reference_{image}_blocks.npy")).astype(np.float64)
p = wire.RtParams.from_buffer_copy(z["params"].tobytes())
sc = scenes.Scene(scene, z["vertices"], z["indices"], z["material_index"], z["shadow_index"], z["materials"],
                  z["sphere_lights"], z["area_lights"], z["disc_lights"],
                  None if os.environ.get("SKY", "1") == "0" else scenes.procedural_skybox(64), p)      # SKY=0: no sky light at all
core = native.RtCore(0)
core.upload(sc); core.bvh_build(); core.set_params(p)
core.render_frames(scenes.frame_seeds(spp, 1))
if "bloom" in z.files and os.environ.get("BLOOM", "1") != "0":
    core.post_bloom(wire.IMAGE_CONVERGED, wire.RtBloomParams.from_buffer_copy(z["bloom"].tobytes()))
    got = png.linear_to_srgb8(core.readback(wire.IMAGE_POSTPROCESS)[::-1, :, :3]).astype(np.float64)
else:
    got = png.linear_to_srgb8(core.readback(wire.IMAGE_CONVERGED)[::-1, :, :3]).astype(np.float64)
if len(sys.argv) > 4:
    png.write_png(sys.argv[4], got.astype(np.uint8))
k = 16
gb = got[:want.shape[0] * k, :want.shape[1] * k].reshape(want.shape[0], k, want.shape[1], k, 3).mean((1, 3))
s = (gb - want).mean(-1)
for row in s:
    print("".join("+" if v > 12 else "-" if v < -12 else chr(48 + min(int(abs(v) / 2), 9)) for v in row))
print("median |d|", np.median(np.abs(gb - want).max(-1)), "signed mean", (gb - want).mean((0, 1)).round(2))
