"""Experiment: frame time of the car scene as is and scaled x8 (finer Morton cells -> better tree)."""
import json, sys
import numpy as np
sys.path.insert(0, ".")
from raytracing_amd import native, scenes, wire  # noqa: E402

core = native.RtCore(0)
out = []
for k in (1.0, 8.0):
    sc = scenes.car_standin(1920, 1080)
    sc.vertices["position"] *= np.float32(k)
    sc.sphere_lights["position"] *= np.float32(k)
    sc.sphere_lights["radius"] *= np.float32(k)
    p = sc.params
    for i in (12, 13, 14):
        p.camera2World[i] *= k
    p.focalLength *= k
    p.lensRadius *= k
    core.upload(sc)
    core.bvh_build()
    core.set_params(p)
    core.render_frames(scenes.frame_seeds(8, seed=1))
    core.enable_kernel_timing(True)
    core.reset_stats()
    core.set_dirty()
    core.render_frames(scenes.frame_seeds(32, seed=2))
    st = core.stats()
    sh, n = core.kernel_time("k_shadow_rays")
    ex, n2 = core.kernel_time("k_extend")
    core.enable_kernel_timing(False)
    out.append({"scale": k, "ms_per_frame": round(st["renderMs"] / 32, 3), "shadow_ms_per_frame": round(sh / 32, 3), "extend_ms_per_frame": round(ex / 32, 3),
                "build_ms": round(st["buildMs"], 3)})
print(json.dumps(out, indent=1))
core.close()
