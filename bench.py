#!/usr/bin/env python3
"""bench.py -- throughput of the rtcore hot path on N MI355X of one node.

Workload (the headline metric of this project): the car scene (procedural stand-in, the real
mesh and HDRIs are absent from the reference checkout), 1920x1080, 64 samples
per pixel.  One STEP = one finished 64-spp image = 64 x { CSMain dispatch,
ShadowMap dispatch, accumulate } = 64 calls of the reference's CustomRender
(Assets/Scripts/RayTracing/RayTracing.cs:118-164), followed by gathering the
accumulated tiles on rank 0.  value = Mrays/s over the whole job, counting RAYS
WALKED: the BVHRayTrace calls (camera, bounce or shadow segment) a traversal
kernel took through the tree.  The reference makes more calls for the same frames
("rays_answered_per_step"; the rate counted on them is "value_rays_answered"):
the 16 soft-shadow rays of a ShadowMap pixel none of whose rays can meet a shadow
caster are settled without a walk.  The CPU leg walks every ray it answers.

For N > 1 the fixed frame is cut into 8x8 tiles, one of every N consecutive tiles
to each rank in an order that changes from group to group (strong scaling); the scene is broadcast once with RCCL before the timed region
and every rank builds its own (bit-identical) BVH.

Launch: python bench.py --gpus 1 [--steps K --warmup W], or under
torch.distributed.run with one rank per GPU.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from raytracing_amd import native, scenes, wire  # noqa: E402
from raytracing_amd import dist as rtdist  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E, spec
SIMDS = 256 * 4         # compute units x SIMDs
MAX_CLOCK_GHZ = 2.4     # a wave64 vector instruction occupies its SIMD for 4 clocks
L1_GATHER_PEAK_G = round(1.68 * 256 * 2.4, 1)      # measured, tools/ubench: divergent 16-byte loads that hit the L1
VALU_PEAK_GINST = SIMDS * MAX_CLOCK_GHZ / 4.0   # wave-instructions per ns over the chip = 614.4 G/s


def source_id():
    """Hash of the kernel sources: profiles/pmc_traffic.json records the one it was measured on."""
    import glob
    import hashlib
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "raytracing_amd", "csrc", "*.h*")) + glob.glob(os.path.join(ROOT, "include", "*.h"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def make_scene(args):
    if args.workload == "car":
        return scenes.car_standin(args.width, args.height, detail=args.detail)
    if args.workload == "cornell":
        return scenes.cornell_box(args.width, args.height)
    if args.workload == "pbr":
        return scenes.pbr_sphere_grid(args.width, args.height)
    raise SystemExit(f"unknown workload {args.workload}")


def path_ray_bytes(c):
    """Algorithmic bytes of camera / bounce rays: 32 B per internal box tested
    (24 B box + two child words), 28 B per leaf box tested (24 B box + sorted
    index), 36 B per triangle that passed its box (three positions), 104 B per
    ray that ends in a hit (normals 36, material index 4, flags 8, material 56)
    and 96 B of path state per ray (a 48-B record written once, read once)."""
    return 32 * c["internalVisits"] + 28 * c["leafVisits"] + 36 * c["triTests"] + 104 * c["hits"] + 96 * c["rays"]


def shadow_ray_bytes(c):
    """Algorithmic bytes of the soft-shadow rays: the same node, leaf and
    triangle terms; a hit only needs the 4-B castShadow flag; per ray 2 B of the
    pixel's 32-B record (shared by its 16 rays) and the 1-B result flag."""
    return 32 * c["internalVisits"] + 28 * c["leafVisits"] + 36 * c["triTests"] + 4 * c["hits"] + 3 * c["rays"]


COUNT_KEYS = ("rays", "internalVisits", "leafVisits", "triTests", "hits")
SHADOW_KEYS = ("shadowRays", "shadowInternalVisits", "shadowLeafVisits", "shadowTriTests", "shadowHits")


def cpu_baseline(sc, args):
    """The CPU checker (oracle/) timed on the host cores on a bounded sample of
    the same workload: the same scene and sampling at reduced resolution."""
    from oracle import oracle_py
    w, h = max(64, args.width // 4), max(36, args.height // 4)
    p = sc.params.copy()
    p.width, p.height = w, h
    p.cameraInverseProjection[:] = scenes.inverse_projection(50.0, w / h).tolist()
    p.tileRank, p.tileCount = 0, 1
    orc = oracle_py.Oracle(sc).build()
    seeds = scenes.frame_seeds(64, seed=1)
    rays, t0, frames = 0, time.time(), 0
    use_shadow = bool(p.flags & wire.FLAG_MATTE_MASK)
    while frames < len(seeds) and (frames == 0 or time.time() - t0 < 15.0):
        p.seed = float(seeds[frames])
        _, cnt = orc.render(p, counters=True)
        rays += cnt["rays"]
        if use_shadow:
            # the ShadowMap kernel as the reference runs it: 1 primary ray per pixel + 16 shadow rays per pixel whose
            # primary ray hit something, every one of them walked
            _, scnt = orc.shadowmap(p, counters=True)
            rays += scnt["rays"]
        frames += 1
    dt = time.time() - t0
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": int(oracle_py.lib().orc_num_threads()), "kind": "port",
            "ms_per_frame_1spp_at_full_resolution": round(dt / frames * (args.width * args.height) / (w * h) * 1e3, 1),
            "sample": f"{sc.name} scene, {w}x{h} (1/16 of the pixels), {frames} frame(s) of 1 spp incl. ShadowMap pass, "
                      f"{dt:.1f} s; every BVHRayTrace call counted (the checker walks each one)"}


def query_ray_bytes(c):
    """Algorithmic bytes of first-hit queries (rt_trace_rays_device): the node, leaf and triangle terms of a camera ray,
    32 B of ray record read and a 16-B hit record written per ray; no shading inputs."""
    return 32 * c["internalVisits"] + 28 * c["leafVisits"] + 36 * c["triTests"] + 48 * c["rays"]


def soup_main(args, rank, world, local_rank, device, dist, barrier):
    """--workload soup10m: the synthetic triangle soup (centres uniform in [-100, 100]^3, edge vectors uniform in
    [-0.5, 0.5]^3, one diffuse material), BVH build timed by itself, then the traversal stress: 2^24 incoherent rays
    (origins uniform in the box, cosine-hemisphere directions about random normals) resident in device memory, one
    rt_trace_rays_device call per step.  The reference cannot hold this scene (524 288-triangle cap,
    Assets/Scripts/BVH/Constants.cs:3-6).  Its tree, 2.7 GB of records, does not fit the 256 MB Infinity Cache, so
    here HBM is the yardstick: roofline.bound = "hbm".  N ranks: every rank builds the whole tree and walks 1/N of the rays."""
    import torch
    n_tri, n_rays = args.triangles, args.rays
    sc = scenes.triangle_soup(n_tri, seed=5)
    core = native.RtCore(local_rank)
    core.upload(sc)
    core.bvh_build()
    build_first_ms = core.stats()["buildMs"]
    build = []
    for _ in range(3):
        core.bvh_build()
        build.append(core.stats()["buildMs"])
    build_ms = min(build)
    for kv in args.option:
        k, v = kv.split("=")
        core.set_option(int(k), int(v))
    o, d = scenes.stress_rays(n_rays, seed=7)
    lo, hi = n_rays * rank // world, n_rays * (rank + 1) // world
    n = hi - lo
    o4 = torch.zeros((n, 4), dtype=torch.float32)
    d4 = torch.zeros((n, 4), dtype=torch.float32)
    o4[:, :3] = torch.from_numpy(o[lo:hi])
    d4[:, :3] = torch.from_numpy(d[lo:hi])
    del o, d
    o4, d4 = o4.to(device), d4.to(device)
    hits = torch.empty((n, 4), dtype=torch.float32, device=device)
    torch.cuda.synchronize()

    def step():
        core.trace_rays_device(o4.data_ptr(), d4.data_ptr(), n, hits.data_ptr())

    core.enable_counters(True)
    core.reset_stats()
    step()
    st = core.stats()
    core.enable_counters(False)
    local = {k: int(st[k]) for k in COUNT_KEYS}
    counts = torch.tensor([local[k] for k in COUNT_KEYS], dtype=torch.int64, device=device if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(counts)
    total = {k: int(v) for k, v in zip(COUNT_KEYS, counts.tolist())}
    for _ in range(args.warmup):
        step()
    barrier()
    torch.cuda.synchronize()
    core.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    core.synchronize()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    hit_share = float((hits[:, 3].view(torch.int32) != -1).float().mean().item())
    # the launch by itself, HIP events on the library's stream
    roof_steps = 0 if args.no_kernel_timing else min(args.steps, 3)
    core.reset_stats()
    core.enable_kernel_timing(roof_steps > 0)
    for _ in range(roof_steps):
        step()
    core.synchronize()
    ms_launch, n_launch = core.kernel_time("k_trace")
    # the same rays with boxes beyond the closest hit left out (RT_OPT_PRUNE_BY_DISTANCE 2: not exact for every ray)
    pruned_ms = None
    if roof_steps and not args.option:
        core.set_option(wire.OPT_PRUNE_BY_DISTANCE, 2)
        core.reset_stats()
        for _ in range(roof_steps):
            step()
        core.synchronize()
        pm, pn = core.kernel_time("k_trace")
        pruned_ms = pm / max(pn, 1)
        core.set_option(wire.OPT_PRUNE_BY_DISTANCE, 1)
    core.enable_kernel_timing(False)
    if world > 1:
        et = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(et, op=dist.ReduceOp.MAX)
        elapsed = float(et[0].item())
    workload_name = (f"{n_tri}-triangle random soup, {n_rays} incoherent rays per step (origins uniform in the box, cosine-hemisphere "
                     f"directions), first-hit queries on rays resident in device memory")
    if rank == 0:
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic_soup.json")
        measured = json.load(open(tpath)).get("k_traverse<ArraySrc>", {}) if os.path.exists(tpath) else {}
        same = measured.get("workload") == workload_name and measured.get("n_gpus") == world
        alg = query_ray_bytes(local)
        roofline = None
        if n_launch:
            dur = ms_launch / n_launch * 1e-3
            gbps = alg / dur / 1e9
            roofline = {"bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBS, 4),
                        "traffic": measured.get("hbm_bytes_per_launch") if same else None, "kernel": "k_traverse<ArraySrc>",
                        "launches": int(n_launch), "avg_launch_ms": round(ms_launch / n_launch, 4), "rays_per_launch": n,
                        "algorithmic_bytes_per_launch": int(alg), "algorithmic_bytes_per_ray": round(alg / max(local["rays"], 1), 1),
                        "note": "achieved = algorithmic bytes (32 B per box tested, 28 B per leaf box, 36 B per triangle tested, 48 B of ray and "
                                "hit record per ray; counts from the counting kernels of the same launch) / launch time by HIP events. "
                                "traffic = memory-side bytes per launch from the committed rocprofv3 --pmc passes of this command "
                                "(FETCH_SIZE x 2 + WRITE_SIZE, profiles/pmc_traffic_soup.json)"}
            if same:
                hb = measured["hbm_bytes_per_launch"] / dur / 1e9
                roofline["measured"] = {"hbm_gbps": round(hb, 1), "hbm_frac": round(hb / HBM_PEAK_GBS, 4),
                                        "traffic_over_algorithmic": round(measured["hbm_bytes_per_launch"] / alg, 3),
                                        "valu_issue_frac": round(measured.get("valu_instructions_per_launch", 0) / dur / 1e9 / VALU_PEAK_GINST, 4),
                                        "counters": {k: measured.get(k) for k in ("l2_hit_rate", "lane_utilisation", "waves_per_simd", "valu_busy",
                                                                                  "td_busy", "ta_busy")},
                                        "pmc_sources_match": measured.get("source_id") == source_id()}
        build_bytes = 436.0 * n_tri
        out = {
            "metric": "Mrays/s", "value": round(n_rays * args.steps / elapsed / 1e6, 3), "unit": "Mrays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload_name, "parallelism": f"rays split over {world} GPU(s), tree built on every GPU"},
            "hit_share": round(hit_share, 4),
            "traversal_counts_per_step": total,
            "ms_per_step_boxes_beyond_the_closest_hit_left_out": None if pruned_ms is None else round(pruned_ms, 3),
            "bvh_build": {"ms": round(build_ms, 3), "first_ms": round(build_first_ms, 3), "mtris_per_s": round(n_tri / build_ms / 1e3, 1),
                          "algorithmic_bytes_per_triangle": 436, "achieved_gbps": round(build_bytes / (build_ms * 1e-3) / 1e9, 1),
                          "frac_of_hbm_peak": round(build_bytes / (build_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                          "note": "436 B per triangle is the contract figure for the reference's tree alone; this build also makes the "
                                  "traversal tree, its rotations and the leaf records (about 2.6 kB moved per triangle)"},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle_py
            orc = oracle_py.Oracle(sc).build()
            m = 1 << 18
            oc = o4[:m, :3].cpu().numpy().copy()
            dc = d4[:m, :3].cpu().numpy().copy()
            t1 = time.time()
            orc.trace(oc, dc)
            dt = time.time() - t1
            out["cpu_baseline"] = {"value": round(m / dt / 1e6, 3), "unit": "Mrays/s", "cores": int(oracle_py.lib().orc_num_threads()), "kind": "port",
                                   "sample": f"the first {m} of the same rays through the checker's walk of the reference's tree, {dt:.1f} s "
                                             f"(its build of the {n_tri}-triangle scene not timed)"}
        print(json.dumps(out), flush=True)
    core.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="car")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--detail", type=float, default=1.0)
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL over xGMI); gloo stages the two exchanges through host "
                                                      "memory, for rehearsing the multi-rank path on one GPU")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-dropin", action="store_true", help="skip the per-frame drop-in leg")
    ap.add_argument("--plain-order", action="store_true", help="one launch after the other on one stream (for hardware-counter "
                                                               "passes, which see each kernel by itself)")
    ap.add_argument("--option", action="append", default=[], metavar="K=V", help="rt_set_option(K, V) before rendering (experiments)")
    ap.add_argument("--triangles", type=int, default=10_000_000, help="soup10m: triangles")
    ap.add_argument("--rays", type=int, default=1 << 24, help="soup10m: rays per step")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    comm_device = device if args.backend == "nccl" else torch.device("cpu")
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    def barrier():
        if dist is not None:
            dist.barrier()

    if args.workload == "soup10m":
        soup_main(args, rank, world, local_rank, device, dist, barrier)
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---- scene: generated on rank 0, broadcast over RCCL, built on every GPU
    sc = make_scene(args) if rank == 0 else None
    if world > 1:
        sc = rtdist.broadcast_scene(sc, comm_device)
    core = native.RtCore(local_rank)
    core.upload(sc)
    core.bvh_build()
    build_first_ms = core.stats()["buildMs"]         # the process's first build also pays for loading the kernels
    core.synchronize()
    t_build = time.perf_counter()
    core.bvh_build()
    build_wall_ms = (time.perf_counter() - t_build) * 1e3     # host clock round the call: a build of as many triangles reuses its buffers
    build_ms = core.stats()["buildMs"]
    params = sc.params.copy()
    # 8x8 tiles for a share of the image: the ranks' shares then differ by 2 % in time (64x64 tiles: by 20 %)
    tile_size = 64 if world == 1 else 8
    params.tileRank, params.tileCount, params.tileSize = rank, world, tile_size
    core.set_params(params)
    if args.plain_order:
        core.set_option(wire.OPT_OVERLAP_SHADOW, 0)
    for kv in args.option:
        k, v = kv.split("=")
        core.set_option(int(k), int(v))
    seeds = scenes.frame_seeds(args.spp, seed=1)
    gatherer = rtdist.TileGatherer(args.width, args.height, tile_size, comm_device)
    staging = None if comm_device == device else torch.zeros(gatherer.n_local * 4, dtype=torch.float32, device=device)
    assert gatherer.n_local == core.local_pixel_count
    # the library's slot->pixel mapping and the host-side statement of it must agree
    assert np.array_equal(rtdist.tile_pixel_map(args.width, args.height, tile_size, rank, world).astype(np.uint32), core.local_pixel_map())

    def gather_frame():
        """Accumulated tiles -> rank 0, once per finished image (device to device)."""
        if staging is None:
            core.readback_local_device(wire.IMAGE_CONVERGED, gatherer.local_buffer().data_ptr())
        else:
            core.readback_local_device(wire.IMAGE_CONVERGED, staging.data_ptr())
            gatherer.local_buffer().copy_(staging)
        return gatherer.gather()

    def step():
        core.set_dirty()
        core.render_frames(seeds)
        gather_frame()

    # ---- count pass (untimed): exact rays / box tests / triangle tests of one step, counters on: the same walk as the
    #      timed one (two levels of the tree per step, the same boxes left out), in launches of the plain order
    core.enable_counters(True)
    core.reset_stats()
    step()
    st = core.stats()
    core.enable_counters(False)
    ALL_KEYS = COUNT_KEYS + ("raysWalked",)
    counts = torch.tensor([st[k] for k in ALL_KEYS], dtype=torch.int64, device=comm_device)
    local_shadow = {k: int(st[sk]) for k, sk in zip(COUNT_KEYS, SHADOW_KEYS)}
    local_path = {k: int(st[k]) - local_shadow[k] for k in COUNT_KEYS}
    if world > 1:
        dist.all_reduce(counts)
    total = {k: int(v) for k, v in zip(ALL_KEYS, counts.tolist())}
    rays_answered, rays_walked = total["rays"], total["raysWalked"]

    for _ in range(args.warmup):
        step()

    barrier()
    torch.cuda.synchronize()
    core.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    core.synchronize()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0

    # ---- the per-frame drop-in sequence (untimed by `value`): what RayTracing.CustomRender does once per
    # OnRenderImage (RayTracing.cs:118-164) when the host keeps its own frame loop: UpdateParameters (new seed),
    # Dispatch(CSMain), Dispatch(ShadowMap), Blit(addMaterial) -- the host makes the calls one frame at a time.
    dropin_frames = 0 if args.no_dropin else min(args.spp, 32)
    dropin_ms = None
    dropin_readback_ms = {}
    if dropin_frames:
        use_shadow = bool(params.flags & wire.FLAG_MATTE_MASK)
        pinned = torch.empty((args.height, args.width, 4), dtype=torch.float32).pin_memory()
        host_frame = np.empty((args.height, args.width, 4), np.float32)
        # without a readback, with the blocking rt_readback of every frame (what Unity's ReadPixels would be), and with
        # rt_readback_async into page-locked memory, which leaves while the next frame is traced
        for readback in (None, "blocking", "async"):
            for timed in (False, True):
                core.set_dirty()
                core.synchronize()
                t1 = time.perf_counter()
                for k in range(dropin_frames):
                    params.seed = float(seeds[k])
                    core.set_params(params)
                    core.dispatch()
                    if use_shadow:
                        core.dispatch_shadowmap()
                    core.accumulate(k, use_shadow)
                    if readback == "blocking":
                        core.readback_into(wire.IMAGE_CONVERGED, host_frame)
                    elif readback == "async":
                        core.readback_async(wire.IMAGE_CONVERGED, pinned.data_ptr())
                if readback == "async":
                    core.readback_wait()
                core.synchronize()
                if timed:
                    ms = (time.perf_counter() - t1) / dropin_frames * 1e3
                    if readback is None:
                        dropin_ms = ms
                    else:
                        dropin_readback_ms[readback] = round(ms, 4)

    # ---- roofline pass (untimed by the wall clock): the same step with every launch bracketed by HIP events
    # on the library's stream.  With the events on, the library keeps its launches in the plain order (the
    # shadow rays are otherwise walked while the later bounces run, by two launches that share the chip with
    # them), so the durations are those of each kernel by itself: k_traverse<ShadowSrc> here, while the timed
    # steps above show up in a kernel trace as k_traverse<ShadowSrcOverlapped>.
    roof_steps = 0 if args.no_kernel_timing else min(args.steps, 2)
    core.reset_stats()
    core.enable_kernel_timing(roof_steps > 0)
    for _ in range(roof_steps):
        step()
    core.synchronize()
    core.enable_kernel_timing(False)
    if world > 1:
        et = torch.tensor([elapsed, dropin_ms or 0.0], dtype=torch.float64, device=comm_device)
        dist.all_reduce(et, op=dist.ReduceOp.MAX)
        elapsed = float(et[0].item())
        dropin_ms = float(et[1].item()) or None

    workload_name = (f"{sc.name} scene (procedural stand-in), {args.width}x{args.height}, {args.spp} spp per step, "
                     f"{sc.n_triangles} triangles, MIS + DoF + ShadowMap pass")
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = rays_walked * args.steps / elapsed / 1e6
        # The traversal kernel has two instantiations in a step: k_traverse<QueueSrc> (camera and bounce
        # rays, timed as "k_extend") and k_traverse<ShadowSrc> (16 shadow rays per pixel, "k_shadow_rays").
        # The roofline is quoted for whichever took more device time; the other one is reported beside it.
        # Both are bound by vector-instruction issue, not by HBM: the tree (tens of MB) is served by L2 and the
        # Infinity Cache, the memory side moves a twentieth of the bytes the rays ask for.  So:
        #   bound "valu": achieved = vector wave-instructions per launch / launch time, peak = 1024 SIMDs x 2.4 GHz / 4
        #                 clocks per wave64 instruction; `useful` = frac x the share of lanes active in those instructions.
        #   "hbm": the memory-side bytes of the same launch (FETCH_SIZE x 2 + WRITE_SIZE) / launch time against 8 TB/s.
        # Instruction and byte counts per launch come from the committed rocprofv3 --pmc passes of this same command
        # (profiles/pmc_traffic.json, which records the workload and the hash of the kernel sources it was measured
        # on); the launch time is measured live, here, with HIP events on the library's stream.
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        measured = json.load(open(tpath)) if os.path.exists(tpath) else {}
        sid = source_id()

        def roof(timer, kernel, alg_bytes, units):
            ms, n = core.kernel_time(timer)
            if not n:
                return None, 0.0
            dur = ms / n * 1e-3
            m = measured.get(kernel, {})
            same = m.get("workload") == workload_name and m.get("n_gpus") == world
            out = {"bound": "valu", "achieved": None, "peak": round(VALU_PEAK_GINST, 1), "unit": "G wave-instructions/s", "frac": None,
                   "useful": None, "traffic": None, "kernel": kernel, "launches": int(n), "avg_launch_ms": round(ms / n, 4),
                   "rays_per_launch": int(units * roof_steps / n)}
            per_launch = alg_bytes * roof_steps / n
            out["algorithmic_bytes_per_launch"] = int(per_launch)
            out["algorithmic_gbps_served_by_caches"] = round(per_launch / dur / 1e9, 1)
            if same:
                valu = m["valu_instructions_per_launch"]
                out["achieved"] = round(valu / dur / 1e9, 2)
                out["frac"] = round(valu / dur / 1e9 / VALU_PEAK_GINST, 4)
                out["useful"] = round(out["frac"] * m["lane_utilisation"], 4)
                out["traffic"] = m["hbm_bytes_per_launch"]
                gbps = m["hbm_bytes_per_launch"] / dur / 1e9
                out["hbm"] = {"bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(gbps / HBM_PEAK_GBS, 4), "traffic_over_algorithmic": round(m["hbm_bytes_per_launch"] / per_launch, 4)}
                out["counters"] = {k: m.get(k) for k in ("l2_hit_rate", "lane_utilisation", "waves_per_simd", "valu_busy",
                                                         "valu_instructions_per_launch", "kernel_clocks_per_launch")}
                if m.get("l1_lane_accesses_per_launch"):
                    acc = m["l1_lane_accesses_per_launch"] / dur / 1e9
                    out["vector_memory"] = {"bound": "l1 data return", "achieved": round(acc, 1), "peak": L1_GATHER_PEAK_G,
                                            "unit": "G lane-accesses/s", "frac": round(acc / L1_GATHER_PEAK_G, 4),
                                            "td_busy": m.get("td_busy"), "ta_busy": m.get("ta_busy"),
                                            "note": "divergent 16-byte gathers; peak = 1.68 lane-accesses per clock and CU with every access "
                                                    "an L1 hit (tools/ubench), 0.85 when they come from L2; td_busy is the share of the "
                                                    "kernel's clocks the L1 data-return units were busy"}
                out["pmc_sources_match"] = m.get("source_id") == sid
            out["note"] = ("frac = vector wave-instructions x 4 clocks / (1024 SIMDs x launch time x 2.4 GHz): the share of the chip's "
                           "vector issue slots the launch used; useful = frac x lane utilisation. Instruction and memory-side byte "
                           "counts per launch: committed rocprofv3 --pmc passes of this command (profiles/), launch time: HIP events "
                           "here, plain launch order. The HBM figure is an upper bound (FETCH_SIZE also counts Infinity-Cache hits)")
            return out, ms

        r_path, ms_path = roof("k_extend", "k_traverse<QueueSrc>", path_ray_bytes(local_path), local_path["rays"])
        r_shadow, ms_shadow = roof("k_shadow_rays", "k_traverse<ShadowSrc>", shadow_ray_bytes(local_shadow), int(st["shadowRaysWalked"]))
        roofline, other = (r_shadow, r_path) if ms_shadow >= ms_path else (r_path, r_shadow)
        out = {
            "metric": "Mrays/s", "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload_name,
                       "parallelism": ("the whole image on one GPU" if world == 1 else f"image tiles {tile_size}x{tile_size}, one of every {world} consecutive tiles to each of {world} GPUs")},
            "value_counts": "rays walked: the BVHRayTrace calls a traversal kernel took through the tree.  The reference makes more "
                            "calls for the same frames (rays_answered_per_step): the library answers the 16 soft-shadow rays of a "
                            "ShadowMap pixel that no shadow caster can reach without a walk",
            "rays_walked_per_step": rays_walked, "rays_answered_per_step": rays_answered,
            "value_rays_answered": round(rays_answered * args.steps / elapsed / 1e6, 3),
            "ms_per_frame_1spp": round(ms_per_step / args.spp, 4),
            "ms_per_image": round(ms_per_step, 3),
            "per_frame_dropin": None if dropin_ms is None else {
                "ms_per_frame_1spp": round(dropin_ms, 4), "frames": dropin_frames,
                "ms_per_frame_with_a_full_frame_readback": dropin_readback_ms,
                "sequence": "rt_set_params(seed) + rt_dispatch + rt_dispatch_shadowmap + rt_accumulate per frame (RayTracing.cs:118-164); "
                            "the library keeps four such frames in flight (RT_OPT_PIPELINE_FRAMES)"},
            "paths_per_step": args.width * args.height * args.spp,
            "bvh_build_ms": round(build_ms, 3),
            "bvh_build_wall_ms": round(build_wall_ms, 3),
            "bvh_build_first_ms": round(build_first_ms, 3),
            "bvh_build_bytes_per_triangle": {"contract": 436, "reference_tree_and_leaves": 690, "with_traversal_tree_and_rotations": 2600},
            "bvh_build_mtris_per_s": round(sc.n_triangles / max(build_ms, 1e-6) / 1e3, 2),
            "traversal_counts_per_step": total,
            "roofline": roofline,
            "roofline_other_kernel": other,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sc, args)
        print(json.dumps(out), flush=True)
    core.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
