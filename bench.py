#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric (Mrays/s, primary+secondary) on configs[1]:
synthetic 1,002,528-triangle heightfield + ground plane, primary + shadow ray casting at 1920x1080
(SURVEY 8d C2), through the rt_hip_* C ABI with every input resident in HBM.

A "step" = one full frame: camera-ray generation + closest hit, one shadow query per hit pixel, the local
illumination and the RGB f64 frame -- for this depth-1 camera all in ONE launch (k_frame_direct; deeper ray trees
run k_primary / k_shadow / k_shade per level).  `--streams S` (default 2) keeps S
frames in flight per GPU, frame k on stream k % S with its own rt_hip_scene_share handle: wave costs in
these kernels differ 50x, and the draining tail of one frame is filled by the next frame's waves.  The time
per frame with one frame in flight is reported beside it (config.ms_per_frame_one_in_flight).

Multi-GPU (one process per GPU, torchrun): C2 is a deterministic ray cast, so ranks render
independent frames (rank r renders the camera of frame step*N + r on a small orbit): the path
partitions by frame, no data-path collective, "scaling": "weak".  `--mode samples` runs the sample-split
form instead (each rank renders its own jittered sample of ONE frame, RCCL all-reduce(sum) of the
radiance, overlapped with the next step's render on a second stream).

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FP64_VALU_PEAK_TF = 78.6   # MI355X vector FP64 (spec, FMA = 2 flop)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--cells", type=int, default=708, help="heightfield cells per side (708 -> 1,002,528 triangles)")
    ap.add_argument("--leaf-refine", type=int, default=None, help="rt_hip_scene_create leaf_refine (include/rt_hip.h; default: the library's own hierarchy)")
    ap.add_argument("--mode", choices=["frames", "samples"], default="frames")
    ap.add_argument("--streams", type=int, default=2, help="frames in flight per GPU (one scene handle and HIP stream each)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N ranks on one GPU)")
    ap.add_argument("--no-profile", action="store_true", help="no per-kernel events inside the timed region (no roofline object)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of the frame timed on the CPU (0 = auto)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import ray_tracing_amd as rta
    import scenes

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        args.gpus = world
    rehearsal = args.backend != "nccl"
    if rehearsal:
        local_rank = local_rank % max(1, torch.cuda.device_count())       # every rank on the GPUs that exist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if rehearsal else dev                       # where the collectives' tensors live
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group(args.backend)
        else:
            dist.init_process_group("nccl", device_id=dev)

    # ---- scene (host: build mesh + BSP; device: upload) ----
    t0 = time.time()
    ctx = rta.Context(local_rank)
    gscene, _ = scenes.build_heightfield(args.cells, gpu=True, oracle=False, leaf_refine=args.leaf_refine)
    raw_scene = gscene.device(ctx)
    n_tris = gscene.num_objects - 1
    build_s = time.time() - t0

    def camera_for(frame):
        # frame 0 = the C2 camera of SURVEY 8d; the others swing around it by at most 1.8 degrees, so that every
        # frame is the C2 view (same cost) but no two consecutive frames or ranks render the same rays
        a = 0.001 * (frame % 32)
        c = dict(scenes.C2_CAMERA)
        vx, vy, vz = c["view"]
        ca, sa = math.cos(a), math.sin(a)
        c["view"] = (vx * ca - vy * sa, vx * sa + vy * ca, vz)
        c["eye"] = tuple(-x for x in c["view"])
        c["left"] = (-c["eye"][1], c["eye"][0], 0.0)
        return rta.Camera(**c)

    cam0 = camera_for(0)
    W, H = cam0.width, cam0.height
    npix = W * H
    rgb = [torch.zeros(npix * 3, dtype=torch.float64, device=dev) for _ in range(2)]
    prim = torch.zeros(npix, dtype=torch.int32, device=dev)
    params = rta.Params(background=(200, 200, 200))
    stream = torch.cuda.current_stream()
    sptr = C.c_void_p(stream.cuda_stream)
    # frames in flight: frame k runs on stream k % S with its own scene handle (the handle owns the per-frame
    # scratch), so the low-occupancy tail of one frame's kernels is filled by the next frame's waves
    S = max(1, args.streams)
    fl_streams = [stream] + [torch.cuda.Stream() for _ in range(S - 1)]
    fl_scenes_keep = [gscene.share(ctx) for _ in range(S - 1)]
    fl_scenes = [raw_scene] + [sh.raw for sh in fl_scenes_keep]
    fl_rgb = [[torch.zeros(npix * 3, dtype=torch.float64, device=dev) for _ in range(2)] for _ in range(S)]
    fl_prim = [prim] + [torch.zeros(npix, dtype=torch.int32, device=dev) for _ in range(S - 1)]
    comm_stream = torch.cuda.Stream() if (world > 1 and args.mode == "samples") else None

    def render(step, buf):
        frame = (step * world + rank) if args.mode == "frames" else 0
        cam = cam0 if frame == 0 else camera_for(frame)
        if args.mode == "samples":
            params.seed = step * world + rank
        k = step % S
        rta.check(rta.hip.rt_hip_render_whitted_dev(fl_scenes[k], cam.raw, params, C.c_void_p(buf.data_ptr()),
                                                    C.c_void_p(fl_prim[k].data_ptr()), C.c_void_p(fl_streams[k].cuda_stream)))
        return cam

    def one_step(step):
        buf = fl_rgb[step % S][(step // S) & 1]
        cam = render(step, buf)
        if comm_stream is not None:
            # radiance all-reduce of this step overlaps the next step's render
            comm_stream.wait_stream(fl_streams[step % S])
            with torch.cuda.stream(comm_stream):
                if rehearsal:                      # gloo: the collective runs on a host copy (N ranks rehearsed on one GPU)
                    host = buf.cpu(); dist.all_reduce(host); buf.copy_(host)
                else:
                    dist.all_reduce(buf)
        return cam

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- statistics passes (outside the timed region): per-ray work of the traversal actually performed,
    #      separately for the primary pass (same scene without its light) and the shadow pass (difference) ----
    def stats_of(scene_raw):
        pstat = rta.Params(background=(200, 200, 200), collect_stats=1)
        ctx.stats(reset=True)
        rta.check(rta.hip.rt_hip_render_whitted_dev(scene_raw, cam0.raw, pstat, C.c_void_p(rgb[0].data_ptr()), C.c_void_p(prim.data_ptr()), sptr))
        torch.cuda.synchronize()
        return ctx.stats(reset=True)

    st_all = stats_of(raw_scene)
    rays_pf, shadow_pf = C.c_uint64(0), C.c_uint64(0)
    rta.check(rta.hip.rt_hip_last_ray_count(raw_scene, C.byref(rays_pf), C.byref(shadow_pf)))
    rays_per_frame = rays_pf.value
    dark, _ = scenes.build_heightfield(args.cells, gpu=True, oracle=False, leaf_refine=args.leaf_refine, light=False)
    st_primary = stats_of(dark.device(ctx))
    del dark
    st_shadow = {k: st_all[k] - st_primary[k] for k in st_all}

    for w in range(args.warmup):
        one_step(w)
    barrier()
    rta.check(rta.hip.rt_hip_set_profiling(raw_scene, 0 if args.no_profile else 1))
    prof = rta.Profile()
    rta.check(rta.hip.rt_hip_get_profile(raw_scene, C.byref(prof), 1))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t_start = time.perf_counter()
    ev0.record(stream)
    for s in range(args.steps):
        one_step(s)
    ev1.record(stream)
    if comm_stream is not None:
        stream.wait_stream(comm_stream)
    barrier()
    elapsed = time.perf_counter() - t_start
    ev_ms = ev0.elapsed_time(ev1)
    rta.check(rta.hip.rt_hip_get_profile(raw_scene, C.byref(prof), 1))
    rta.check(rta.hip.rt_hip_set_profiling(raw_scene, 0))

    # for reference, outside the timed region: the same frames strictly one after the other
    t1 = time.perf_counter()
    n_one = min(args.steps, 10)
    for s_ in range(n_one):
        cam_ = camera_for(s_ * world + rank) if args.mode == "frames" else cam0      # held: .raw points into it
        rta.check(rta.hip.rt_hip_render_whitted_dev(raw_scene, cam_.raw, params,
                                                    C.c_void_p(fl_rgb[0][0].data_ptr()), C.c_void_p(prim.data_ptr()), sptr))
    torch.cuda.synchronize()
    ms_one_in_flight = (time.perf_counter() - t1) * 1e3 / n_one

    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        rr = torch.tensor([float(rays_per_frame)], dtype=torch.float64, device=cdev)
        dist.all_reduce(rr)
        total_rays_per_step = float(rr.item())
    else:
        total_rays_per_step = float(rays_per_frame)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed * 1e3 / args.steps
    value = total_rays_per_step * args.steps / elapsed / 1e6

    # ---- roofline of the dominant kernel ----
    kms = {rta.KERNEL_NAMES[i]: (prof.ms[i], prof.launches[i]) for i in range(12) if prof.launches[i]}
    if not kms:
        kms = {"k_shadow": (float("nan"), 1)}
    dom = max(kms, key=lambda k: kms[k][0])
    dom_ms = kms[dom][0] / kms[dom][1]
    st = st_shadow if dom == "k_shadow" else st_primary if dom == "k_primary" else st_all
    nrays = max(1, st["rays"])
    nodes_per_ray = st["node_visits"] / nrays
    prims_per_ray = st["prim_tests"] / nrays
    exact_per_ray = st["exact_tests"] / nrays
    # Algorithmic bytes of THIS traversal (DESIGN.md "Roofline"): a wave of 64 rays walks the tree as a packet, so
    # a box test's 64 B (half a 128-B child pair, or a leaf header) and a cull record's 128 B are moved once per
    # WAVE that needs them; rays in (48 B), hits out (32 B) and the 96-B exact triangles of the confirmation
    # are per ray.  The per-lane figure (what 64 independent rays would fetch) is reported beside it.
    wave_boxes_per_ray = st["wave_node_visits"] / nrays
    wave_recs_per_ray = st["wave_prim_tests"] / nrays
    # k_frame_direct generates its rays and keeps the hits in registers: its only per-ray traffic is the frame
    # (24 B colour + 4 B id per pixel = per 2 rays)
    io_bytes = 14 if dom == "k_frame_direct" else 48 + 32
    bytes_per_ray = io_bytes + 64 * wave_boxes_per_ray + 128 * wave_recs_per_ray + 96 * exact_per_ray
    lane_bytes_per_ray = io_bytes + 64 * nodes_per_ray + 128 * prims_per_ray + 96 * exact_per_ray
    flops_per_ray = 40 * nodes_per_ray + 45 * prims_per_ray + 150 * exact_per_ray
    rays_dom = (rays_per_frame - shadow_pf.value) if dom == "k_primary" else shadow_pf.value if dom == "k_shadow" else rays_per_frame
    ach_gbs = bytes_per_ray * rays_dom / (dom_ms * 1e-3) / 1e9
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "r01_traffic.json")
    if os.path.exists(tfile):
        try:
            traffic = json.load(open(tfile)).get(dom, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    tf = flops_per_ray * rays_dom / (dom_ms * 1e-3) / 1e12
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(ach_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach_gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
                "kernel_ms": round(dom_ms, 4), "rays_per_launch": int(rays_dom),
                "bytes_per_ray": round(bytes_per_ray, 1), "lane_bytes_per_ray": round(lane_bytes_per_ray, 1),
                "box_tests_per_ray": round(nodes_per_ray, 2), "cull_records_per_ray": round(prims_per_ray, 2),
                "exact_tests_per_ray": round(exact_per_ray, 3),
                "wave_box_fetches_per_ray": round(wave_boxes_per_ray, 3), "wave_record_fetches_per_ray": round(wave_recs_per_ray, 3),
                "note": "scene (0.3 GB) is L2/Infinity-Cache resident: the walk is bound by FP64 VALU issue + LDS/scalar latency, not HBM",
                "valu_fp64": {"achieved_tflops": round(tf, 2), "peak_tflops": FP64_VALU_PEAK_TF, "frac": round(tf / FP64_VALU_PEAK_TF, 4)},
                "all_kernels_ms": {k: round(v[0] / v[1], 4) for k, v in kms.items()}}

    # ---- CPU baseline: the oracle (reference semantics, exhaustive BSP walk) on the host cores ----
    cpu = None
    if not args.no_cpu_baseline:
        import oracle_bind as ob
        _, oscene = scenes.build_heightfield(args.cells, gpu=False, oracle=True)
        ocam = scenes.oracle_camera(scenes.C2_CAMERA)
        threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
        rows = args.cpu_rows or 8
        # calibrate on a few rows in the middle of the frame, then size the sample for ~15 s
        first = (H // 2) * W
        t1 = time.perf_counter()
        _, _, cnt = oscene.render_whitted(ocam, background=(200, 200, 200), first=first, count=rows * W, threads=threads, counters=True)
        dt = time.perf_counter() - t1
        if not args.cpu_rows:
            rows = int(max(8, min(H, rows * 15.0 / max(dt, 1e-3))))
            first = ((H - rows) // 2) * W
            t1 = time.perf_counter()
            _, _, cnt = oscene.render_whitted(ocam, background=(200, 200, 200), first=first, count=rows * W, threads=threads, counters=True)
            dt = time.perf_counter() - t1
        cpu_rays = int(cnt[0])
        cpu = {"value": round(cpu_rays / dt / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
               "sample": "%d centre rows of the same 1920x1080 frame (%d rays, %.1f s), oracle/rt_oracle.c OpenMP" % (rows, cpu_rays, dt),
               "node_visits_per_ray": round(float(cnt[1]) / max(1, cpu_rays), 1), "prim_tests_per_ray": round(float(cnt[2]) / max(1, cpu_rays), 1)}

    out = {"metric": "Mrays/sec (primary+secondary)", "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "C2: synthetic %d-triangle heightfield + ground plane, primary+shadow ray casting %dx%d, 1 frame per GPU per step"
                                  % (n_tris, W, H), "triangles": n_tris, "resolution": [W, H], "rays_per_frame": int(rays_per_frame),
                      "shadow_rays_per_frame": int(shadow_pf.value), "parallelism": "%s x%d" % (args.mode, world), "frames_in_flight": S, "ms_per_frame_one_in_flight": round(ms_one_in_flight, 4),
                      "leaf_refine": "library default (own hierarchy, leaves < 24)" if args.leaf_refine is None else args.leaf_refine, "scene_build_s": round(build_s, 2), "event_ms_per_step": round(ev_ms / args.steps, 4)},
           "roofline": roofline, "cpu_baseline": cpu}
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
