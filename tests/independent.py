"""Restatements of small pieces of the reference written from its shader text alone, in numpy binary32 scalar
arithmetic, sharing no source with the library or the CPU checker (both of which include include/rt_math.h).
They are slow and only used to cross-check that header."""
import math

import numpy as np

F = np.float32


def _xorshift(s):
    """RandInt, Assets/Shaders/RayTracing/Header.cginc:66-72, on a Python int held to 32 bits."""
    s ^= (s << 13) & 0xFFFFFFFF
    s ^= s >> 17
    s ^= (s << 5) & 0xFFFFFFFF
    return s


def _ftou(x):
    """float -> uint the way Direct3D defines it: NaN and negatives give 0, too large saturates, else truncation."""
    x = float(x)
    if not x > 0.0:
        return 0
    if x >= 4294967296.0:
        return 0xFFFFFFFF
    return int(x)


def rand_sequence(px, py, seed, n):
    """SetSeed (Header.cginc:79-83) for pixel (px, py), then n x Rand (Header.cginc:86-92).  -> (values as float32,
    rngstate after each call).  sin is libm's binary64 sin of the binary32 argument, rounded to binary32."""
    pxf, pyf = F(px), F(py)
    state = _ftou(pxf * pyf)
    for _ in range(3):
        state = _xorshift(state)
    seed = F(seed)
    vals, states = [], []
    for _ in range(n):
        dot = F(pxf * F(12.9898)) + F(pyf * F(78.233))            # dot(_pixel, float2(12.9898f, 78.233f))
        arg = F(F(seed / F(100.0)) * dot)
        result = F(F(math.sin(float(arg))) * F(43758.5453))
        seed = F(seed + F(1.0))
        state = _ftou(F(F(state) + result))                       # rngstate += result: uint + float is a float sum
        state = _xorshift(state)                                  # RandInt
        q = F(F(state) / F(134217728.0))                          # float(1 << 32 - 5) = 2^27
        vals.append(F(q - F(math.floor(float(q)))))               # frac
        states.append(state)
    return np.array(vals, np.float32), np.array(states, np.uint32)
