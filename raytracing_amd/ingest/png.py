"""Minimal PNG codec (8-bit RGB / RGBA / grey, non-interlaced) on zlib, for writing renders and
reading the reference's published output images (Assets/Outputs/*.png; SaveTexture.cs:23-85
writes sRGB-encoded PNGs of the converged target)."""
import struct
import zlib

import numpy as np


def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def write_png(path, image_u8):
    """image_u8: (H, W, 3|4) uint8, row 0 at the top."""
    img = np.ascontiguousarray(image_u8, np.uint8)
    h, w, c = img.shape
    raw = np.concatenate([np.zeros((h, 1), np.uint8), img.reshape(h, w * c)], 1).tobytes()     # filter 0 per row
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, {3: 2, 4: 6}[c], 0, 0, 0)) +
                _chunk(b"IDAT", zlib.compress(raw, 6)) + _chunk(b"IEND", b""))


def read_png(path):
    """-> (H, W, C) uint8."""
    data = open(path, "rb").read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG file")
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"IEND":
            break
    w, h, depth, ctype, _, _, interlace = hdr
    if depth != 8 or interlace != 0 or ctype not in (0, 2, 4, 6):
        raise ValueError("only 8-bit non-interlaced grey / RGB / RGBA PNGs are read")
    c = {0: 1, 2: 3, 4: 2, 6: 4}[ctype]
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(h, 1 + w * c)
    out = np.zeros((h, w * c), np.uint8)
    prev = np.zeros(w * c, np.int32)
    for y in range(h):
        ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        elif ft == 1:
            cur = line.copy().reshape(w, c)
            cur = (np.cumsum(cur, 0) & 255).reshape(-1)
        else:                                   # 3 (average), 4 (Paeth): sequential along the row
            cur = line.copy()
            for i in range(w * c):
                a = cur[i - c] if i >= c else 0
                b = prev[i]
                if ft == 3:
                    pred = (a + b) >> 1
                else:
                    cc = prev[i - c] if i >= c else 0
                    p = a + b - cc
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - cc)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else cc)
                cur[i] = (cur[i] + pred) & 255
        out[y] = cur
        prev = cur
    return out.reshape(h, w, c)


def linear_to_srgb8(rgb):
    """Linear float radiance -> 8-bit sRGB (what a linear-colour-space Unity player writes to an sRGB target)."""
    x = np.clip(np.asarray(rgb, np.float64), 0.0, 1.0)
    s = np.where(x <= 0.0031308, x * 12.92, 1.055 * np.power(x, 1 / 2.4) - 0.055)
    return (s * 255.0 + 0.5).astype(np.uint8)


def texture_from_u8(image_u8_bottom_up):
    """8-bit texels (row 0 = v 0) -> the f32[H, W, 4] UNORM values a sampler returns for a
    linear (sRGBTexture: 0) texture; a missing alpha channel reads 1."""
    img = np.asarray(image_u8_bottom_up, np.uint8)
    out = np.ones(img.shape[:2] + (4,), np.float32)
    out[..., :img.shape[2]] = img.astype(np.float32) / np.float32(255.0)
    return out


def read_texture(path):
    """A PNG as the texture Unity makes of it when imported linear: f32[H, W, 4], row 0 = the
    bottom row of the file (v = 0)."""
    return texture_from_u8(read_png(path)[::-1])
