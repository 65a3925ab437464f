#!/usr/bin/env python3
"""Builds the car stand-in (or a 10 M triangle soup: argument `soup`) three times; run under
`rocprofv3 --kernel-trace` and give the trace to tools/build_breakdown.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import native, scenes  # noqa: E402

soup = len(sys.argv) > 1 and sys.argv[1] == "soup"
sc = scenes.triangle_soup(10_000_000, seed=5, width=64, height=64) if soup else scenes.car_standin(64, 64)
core = native.RtCore(0)
core.upload(sc)
for _ in range(3):
    core.bvh_build()
    print("buildMs", round(core.stats()["buildMs"], 3), flush=True)
