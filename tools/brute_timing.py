"""Device time of the brute-force extend kernel (every ray against every primitive) on the
primitive room at 1080p: rays/s and ray-primitive tests/s.
Usage: python tools/brute_timing.py [mesh_segments]"""
import json
import sys

sys.path.insert(0, ".")
from raytracing_amd import native, scenes, wire  # noqa: E402

seg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
sc = scenes.primitive_room(1920, 1080, mesh_segments=seg)
pr = sc.primitives
n_prims = len(pr.planes) + len(pr.spheres) + len(pr.cubes) + pr.n_triangles
core = native.RtCore(0)
core.upload(sc)
core.set_params(sc.params)
core.render_frames(scenes.frame_seeds(8, seed=1))          # warm-up
FR = 32
core.enable_counters(True)                                  # ray count of the same frames, in a pass of its own
core.reset_stats()
core.set_dirty()
core.render_frames(scenes.frame_seeds(FR, seed=2))
rays = core.stats()["rays"]
core.enable_counters(False)
core.enable_kernel_timing(True)
core.reset_stats()
core.set_dirty()
core.render_frames(scenes.frame_seeds(FR, seed=2))
st = core.stats()
ext_ms, ext_n = core.kernel_time("k_brute_extend")
sh_ms, _ = core.kernel_time("k_shade")
# every launch covers the live paths of one bounce of the frames in flight; rays = sum of live paths
gen_ms, _ = core.kernel_time("k_generate")
print(json.dumps({"frame": [1920, 1080], "frames": FR, "primitives": n_prims, "triangles": pr.n_triangles,
                  "render_ms_per_frame": round(st["renderMs"] / FR, 3),
                  "k_brute_extend_ms_per_frame": round(ext_ms / FR, 3), "k_shade_ms_per_frame": round(sh_ms / FR, 3),
                  "k_generate_ms_per_frame": round(gen_ms / FR, 3), "extend_launches": ext_n,
                  "rays_per_frame": rays // FR, "extend_Mrays_per_s": round(rays / ext_ms / 1e3, 1),
                  "ray_primitive_tests_per_s": round(rays * n_prims / (ext_ms / 1e3), 0)}))
core.close()
