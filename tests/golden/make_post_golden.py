"""Copies the reference's colour-grading look-up tables (data files, 1024x32 8-bit PNGs
under Assets/Textures/Lut, imported as linear / point-filtered / repeat-wrapped textures
per their .meta files) into one fixture:

    python tests/golden/make_post_golden.py /root/reference

-> tests/golden/reference_luts.npz : one u8[32, 1024, 3] array per table, row 0 = the
   BOTTOM row of the image file (texture coordinate v = 0), keyed by file name.
-> tests/golden/reference_graded_pixels.npz : the five published images of ONE frame graded through five of
   those tables (Assets/Outputs/bleach_bypass.png, Candlelight.png, crisp_warm.png, urban_cowboy.png,
   you_can_do_it.png; README.md "Color grading"), as the same 20 000 pixel positions of each (u8[20000, 3], below
   the strip of the table that the host draws along the top of the frame).  The ungraded frame is not published;
   what the five have in common is their source, so grading table B applied to the colours that table A maps onto
   image A must give image B.
"""
import glob
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from raytracing_amd.ingest import png  # noqa: E402


def main(reference_root):
    out = {}
    for path in sorted(glob.glob(os.path.join(reference_root, "Assets", "Textures", "Lut", "*.png"))):
        name = os.path.splitext(os.path.basename(path))[0].replace(" ", "_")
        img = png.read_png(path)
        assert img.shape == (32, 1024, 3) and img.dtype == np.uint8, (path, img.shape)
        out[name] = np.ascontiguousarray(img[::-1])
    np.savez_compressed(os.path.join(HERE, "reference_luts.npz"), **out)
    print("wrote", sorted(out))
    graded = {"Bleach_Bypass": "bleach_bypass.png", "Candlelight": "Candlelight.png", "Crisp_warm": "crisp_warm.png",
              "Urban_cowboy": "urban_cowboy.png", "You_can_do_it": "you_can_do_it.png"}
    frames = {k: png.read_png(os.path.join(reference_root, "Assets", "Outputs", f)) for k, f in graded.items()}
    shape = next(iter(frames.values())).shape
    assert all(f.shape == shape and f.dtype == np.uint8 for f in frames.values())
    skip = 40                                                  # rows of the table strip at the top of the frame
    picks = np.random.default_rng(7).choice((shape[0] - skip) * shape[1], 20000, replace=False)
    np.savez_compressed(os.path.join(HERE, "reference_graded_pixels.npz"), positions=picks.astype(np.int32), rows_skipped=np.int32(skip),
                        **{k: np.ascontiguousarray(f[skip:].reshape(-1, 3)[picks]) for k, f in frames.items()})
    print("wrote graded pixels of", sorted(frames), shape)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
