"""Copies the reference's colour-grading look-up tables (data files, 1024x32 8-bit PNGs
under Assets/Textures/Lut, imported as linear / point-filtered / repeat-wrapped textures
per their .meta files) into one fixture:

    python tests/golden/make_post_golden.py /root/reference

-> tests/golden/reference_luts.npz : one u8[32, 1024, 3] array per table, row 0 = the
   BOTTOM row of the image file (texture coordinate v = 0), keyed by file name.
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


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
