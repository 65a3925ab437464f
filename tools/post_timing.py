"""Device time of every post-process kernel on a 1080p (or WxH) frame, with the bytes
each one has to move, to place them against the HBM roofline.
Usage: python tools/post_timing.py [width height]"""
import json
import sys

import numpy as np

sys.path.insert(0, ".")
from raytracing_amd import native, wire  # noqa: E402
from raytracing_amd.ingest import png  # noqa: E402

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
rng = np.random.default_rng(1)
img = (rng.random((h, w, 4), np.float32) ** 3 * 4).astype(np.float32)
core = native.RtCore(0)
core.image_upload(img)
lut = np.load("tests/golden/reference_luts.npz")["Crisp_warm"]
core.post_lut_upload(png.texture_from_u8(lut))
core.enable_kernel_timing(True)
REPS = 20
for rep in range(REPS + 1):
    if rep == 1:
        core.reset_stats()
    core.post_color_correction(wire.IMAGE_USER, 0.96, 1.33, 1.85)
    core.post_color_grading(wire.IMAGE_USER)
    core.post_bloom(wire.IMAGE_USER, wire.RtBloomParams())
    core.save_rgb24(wire.IMAGE_POSTPROCESS)
px = w * h
pyr = sum((w >> (i + 1)) * (h >> (i + 1)) for i in range(7))
ups = sum((w >> (i + 1)) * (h >> (i + 1)) for i in range(6))
algo = {"k_post_color_correction": 32 * px, "k_post_color_grading": 32 * px, "k_post_threshold": 32 * px,
        "k_post_downsample": 16 * (px + pyr) + 16 * 0, "k_post_upsample": 16 * (2 * ups + ups), "k_post_bloom_compose": 32 * px + 16 * (px >> 2),
        "k_post_rgb24": 19 * px}
out = {}
for name, nbytes in algo.items():
    ms, n = core.kernel_time(name)
    per_call = ms / REPS
    out[name] = {"ms_per_frame": round(per_call, 4), "launches_per_frame": n // REPS, "algorithmic_MB": round(nbytes / 1e6, 1),
                 "GB_per_s": round(nbytes / 1e9 / (per_call / 1e3), 1) if per_call else None}
print(json.dumps({"frame": [w, h], "kernels": out, "total_ms": round(sum(v["ms_per_frame"] for v in out.values()), 3)}, indent=1))
core.close()
