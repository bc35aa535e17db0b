#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric (Mrays/s, primary+secondary; s/frame at fixed spp) through the rt_hip_* C ABI with
every input resident in HBM.

Default workload = BASELINE configs[1]: the synthetic 1,002,528-triangle heightfield + ground plane,
primary + shadow ray casting at 1920x1080.  A STEP = a batch of `--frames-per-step` C2 frames (a camera swinging around
the C2 view); every frame is camera-ray generation + closest hit, one shadow query per hit pixel, the local illumination
and the reference's Vec3b picture (Camera.h:69) -- for this depth-1 camera ONE launch (k_frame_direct).  `--streams S`
keeps S frames in flight per GPU (wave costs differ 50x; the draining tail of one frame is filled by the next).

N > 1 GPUs (one process per GPU; `python bench.py --gpus N` starts its own ranks through torch.distributed.run, or is
started by it): the SAME batch of frames per step, every frame split over the ranks by interleaved 128x64-pixel
super-tiles (rt_hip_params.tile_offset / tile_stride: rank r renders super-tiles r, r+N, ...), the shards all-gathered
over RCCL (rt_hip_comm_all_gather, or torch.distributed with --comm torch) and put back into raster order on every rank
(rt_hip_tiles_to_frame_dev); the gather of one sub-batch overlaps the rendering of the next on a second stream.  Total
work per step is fixed: "scaling": "strong".  `--mode replicas` is the collective-free form (each rank renders its own
frames, "weak"), kept as a secondary figure.

`--config C3|C4|C5` times the other BASELINE configs (depth of field, Bezier scene, photon mapping) with the same JSON
shape; a short run of each is folded into the default line as config.other_configs.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_VALU_PEAK_TF = 78.6     # MI355X vector FP64 (FMA = 2 flop): 256 CUs x 4 SIMDs x 16 lanes x 2 x 2.4 GHz
N_SIMD = 1024                # 256 CUs x 4 SIMDs
N_CU = 256
CLOCK_GHZ = 2.4              # peak engine clock
# MI355X_MICROARCH.md (lines 54, 485): the SIMDs are 32 lanes wide, so a wave64 FP32 / integer / move instruction issues over 2 cycles (with a second wave
# resident), an FP64 one over 4 (78.6 TF FP64 vector = half the 157.3 TF FP32 vector peak).  The VALU roof of a kernel therefore depends on its
# instruction mix: peak = 1024 SIMDs x 2.4 GHz / (4 f64_share + 2 (1 - f64_share)) wave-instructions/s, f64_share from the per-type SQ counters.
VALU_ISSUE_PEAK_ALL_F64_G = N_SIMD * CLOCK_GHZ / 4.0      # 614.4 G wave-instructions/s if every instruction were FP64 (round 2's constant)
VALU_ISSUE_PEAK_ALL_F32_G = N_SIMD * CLOCK_GHZ / 2.0      # 1228.8 if none were
TRAFFIC_FILES = [os.path.join(ROOT, "profiles", "r03_traffic.json"), os.path.join(ROOT, "profiles", "r02_traffic.json")]


def valu_peak_g(f64_share):
    """mix-weighted VALU issue roof in G wave-instructions/s"""
    return N_SIMD * CLOCK_GHZ / (4.0 * f64_share + 2.0 * (1.0 - f64_share))


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=["C2", "C3", "C4", "C5"], default="C2")
    ap.add_argument("--frames-per-step", type=int, default=128, help="C2: frames per step (a step of ~100 ms on one GPU)")
    ap.add_argument("--sub-batch", type=int, default=16, help="C2, N>1: frames per all-gather")
    ap.add_argument("--cells", type=int, default=708, help="heightfield cells per side (708 -> 1,002,528 triangles)")
    ap.add_argument("--leaf-refine", type=int, default=None, help="rt_hip_scene_create leaf_refine (include/rt_hip.h; default: the library's own hierarchy)")
    ap.add_argument("--mode", choices=["tiles", "replicas"], default="tiles")
    ap.add_argument("--comm", choices=["abi", "torch"], default="abi", help="data-path collective: rt_hip_comm_* (RCCL behind the C ABI) or torch.distributed")
    ap.add_argument("--streams", type=int, default=0, help="frames in flight per GPU (0 = 4, or 8 from 8 ranks on: one per hardware queue)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N ranks on one GPU: collectives then run on HOST tensors)")
    ap.add_argument("--emulate-world", type=int, default=0, help="one GPU renders rank 0's share of an N-rank run, no collective (tuning aid; not a result)")
    ap.add_argument("--batch-frames", type=int, default=0, help="C2, N=1: frames per rt_hip_render_whitted_batch_dev launch (0 = one launch per frame, --streams in flight)")
    ap.add_argument("--force-batch-launch", action="store_true", help="C2, N>1: the batched launch also below 8 ranks (rehearsals of the path)")
    ap.add_argument("--no-batch-launch", action="store_true", help="C2, N>1: one launch per frame share (8 frames in flight) instead of one rt_hip_render_whitted_batch_dev launch per sub-batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (reference tree, f64 frame, other configs)")
    ap.add_argument("--others", choices=["full", "short"], default="full", help="C3 / C4 / C5 inside the default line: at BASELINE's full sizes (1024 spp, 256 spp, 10 M photons; ~10 s of GPU time together) or cut down and extrapolated")
    ap.add_argument("--spp", type=int, default=0, help="C3 / C4: samples per pixel timed per step (0 = a short default; the full config is 1024 / 256)")
    ap.add_argument("--photons", type=int, default=0, help="C5: photons per light sample and pass (0 = a short default; the full config is 400000)")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` invoked plainly: start N fresh ranks BEFORE anything here touches the GPU (a process that
    has initialised HIP must never exec), relay their output, return their exit code"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------------------------------------------------------------
class Dist:
    """rank / world, barrier, small reductions; the data-path collective of the chosen kind"""

    def __init__(self, args, torch):
        self.torch = torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.rehearsal = args.backend != "nccl"
        if self.rehearsal:
            self.local_rank %= max(1, torch.cuda.device_count())      # every rank on the GPUs that exist
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device("cuda", self.local_rank)
        self.cdev = torch.device("cpu") if self.rehearsal else self.dev
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            self.dist = dist
            if self.rehearsal:
                dist.init_process_group(args.backend)
            else:
                dist.init_process_group("nccl", device_id=self.dev)

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def reduce(self, value, op="max"):
        if self.dist is None:
            return float(value)
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self.cdev)
        self.dist.all_reduce(t, op={"max": self.dist.ReduceOp.MAX, "min": self.dist.ReduceOp.MIN, "sum": self.dist.ReduceOp.SUM}[op])
        return float(t.item())

    def finish(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


def make_abi_comm(rta, ctx, D):
    """rt_hip_comm over all ranks: rank 0's ncclUniqueId travels through torch.distributed (the launcher's rendezvous).
    Returns the handle, or None when any rank failed (then every rank falls back to torch's collective)."""
    torch = D.torch
    ident = (C.c_ubyte * 128)()
    ok = 1.0
    if D.rank == 0 and rta.hip.rt_hip_comm_unique_id(ident) != 0:
        ok = 0.0
    t = torch.tensor(list(bytes(ident)), dtype=torch.uint8, device=D.cdev)
    D.dist.broadcast(t, src=0)
    ident = (C.c_ubyte * 128)(*t.cpu().tolist())
    comm = C.c_void_p()
    if D.reduce(ok, "min") < 1.0:
        return None
    if rta.hip.rt_hip_comm_create(ctx.raw, ident, D.rank, D.world, C.byref(comm)) != 0:
        sys.stderr.write("rank %d: rt_hip_comm_create failed: %s\n" % (D.rank, rta.hip.rt_hip_last_error().decode()))
        ok = 0.0
    if D.reduce(ok, "min") < 1.0:
        return None
    return comm


def last_rays(rta, raw):
    a, b = C.c_uint64(0), C.c_uint64(0)
    rta.check(rta.hip.rt_hip_last_ray_count(raw, C.byref(a), C.byref(b)))
    return a.value, b.value


def traffic_file():
    for f in TRAFFIC_FILES:
        if os.path.exists(f):
            return f
    return TRAFFIC_FILES[0]


def read_traffic(kernel):
    """per-launch PMC figures of `kernel` from the committed rocprofv3 passes (profiles/r03_traffic.json), or {}"""
    try:
        return json.load(open(traffic_file())).get(kernel, {})
    except Exception:
        return {}


def counters_stale():
    """do the committed counters belong to the library being timed?  True / False, None when the file carries no stamp.
    (tools/collect_profiles.sh stamps the sha256 of the librt_hip.so it profiled; the same file travels to the GPU box.)"""
    import hashlib
    try:
        want = json.load(open(traffic_file())).get("_build", {}).get("librt_hip_so_sha256")
    except Exception:
        want = None
    if not want:
        return None
    lib_dir = os.path.join(ROOT, "ray_tracing_amd", "lib")
    path = os.environ.get("RT_HIP_LIB")
    path = os.path.normpath(os.path.join(lib_dir, path)) if path else os.path.join(lib_dir, "librt_hip.so")
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest() != want
    except OSError:
        return None


def issue_roofline(pmc, kernel_ms):
    """the VALU-issue roofline of one kernel from its per-launch counters and a live kernel time: achieved wave-instructions/s against the
    mix-weighted peak, and the scalar unit's share beside it (one scalar issue per CU and cycle)"""
    vi = pmc.get("SQ_INSTS_VALU")
    if not vi or not kernel_ms:
        return {}
    ach = vi / (kernel_ms * 1e-3) / 1e9
    share = pmc.get("valu_f64_share")
    out = {"achieved": round(ach, 1), "unit": "G wave-instructions/s"}
    if share is not None:
        peak = valu_peak_g(share)
        out.update({"peak": round(peak, 1), "frac": round(ach / peak, 4), "valu_f64_share": round(share, 4), "valu_other_share": round(1.0 - share, 4),
                    "peak_definition": "1024 SIMDs x 2.4 GHz / (4 cycles x f64_share + 2 cycles x the rest) per wave64 instruction (MI355X_MICROARCH.md)"})
    else:
        out.update({"peak": VALU_ISSUE_PEAK_ALL_F64_G, "frac": round(ach / VALU_ISSUE_PEAK_ALL_F64_G, 4),
                    "peak_definition": "no per-type counters for this kernel: every instruction charged 4 cycles (an upper bound of the fraction)"})
    out["frac_if_all_4_cycles"] = round(ach / VALU_ISSUE_PEAK_ALL_F64_G, 4)
    out["frac_if_all_2_cycles"] = round(ach / VALU_ISSUE_PEAK_ALL_F32_G, 4)
    sq = pmc.get("sq", {})
    salu = sq.get("SQ_INSTS_SALU")
    if salu:
        cycles = kernel_ms * 1e-3 * CLOCK_GHZ * 1e9
        out["salu_issue"] = {"frac": round(salu / N_CU / cycles, 4), "wave_insts_per_launch": salu,
                             "definition": "SQ_INSTS_SALU / 256 CUs / kernel cycles at 2.4 GHz: the scalar unit is shared by a CU's four SIMDs"}
    return out


# ---------------------------------------------------------------------------------------------------------------------
def bench_c2(args, D, torch, rta, W):
    rank, world, dev = D.rank, D.world, D.dev
    t0 = time.time()
    ctx = rta.Context(D.local_rank)
    gscene = W.build_heightfield(W.product_stage, args.cells, leaf_refine=args.leaf_refine).sides[0][0]
    raw_scene = gscene.device(ctx)
    n_tris = gscene.num_objects - 1
    build_s = time.time() - t0

    cams = [rta.Camera(**W.c2_orbit_camera(f)) for f in range(32)]
    cam0 = cams[0]
    Wd, Ht = cam0.width, cam0.height
    npix = Wd * Ht
    emulate = args.emulate_world if world == 1 else 0
    tw = emulate or world                                   # ranks the frame is split over
    tiles_mode = args.mode == "tiles" and tw > 1
    # measured with --emulate-world N (Grays/s projected, batched / per frame share): N = 2: 14.5 / 16.3, N = 4: 30.1 / 34.3, N = 8 at 4 queues: 60.6 / 48.3
    batch_launch = tiles_mode and (tw >= 8 or args.force_batch_launch) and not args.no_batch_launch
    S = args.streams or (8 if (tw >= 8 and args.no_batch_launch) else 4)   # frames in flight: one per hardware queue (main()); N = 1 measured 2: 5.63, 3: 5.87, 4: 6.11, 6: 5.81, 8: 6.00 Grays/s
    F = max(1, args.frames_per_step)
    G = max(S, min(args.sub_batch, F))
    while F % G:
        G -= 1
    stream0 = torch.cuda.current_stream()
    streams = [stream0] + [torch.cuda.Stream() for _ in range(S - 1)]
    shared_keep = [gscene.share(ctx) for _ in range(S - 1)]
    handles = [raw_scene] + [sh.raw for sh in shared_keep]
    comm_stream = torch.cuda.Stream() if tiles_mode else None
    params_full = rta.Params(background=(200, 200, 200), frame_format=1)
    params_f64 = rta.Params(background=(200, 200, 200))

    total_st, per_rank = C.c_uint32(), C.c_uint32()
    rta.check(rta.hip.rt_hip_tile_shard_info(cam0.raw, tw, C.byref(total_st), C.byref(per_rank)))
    blk = per_rank.value * 8192 * 3                         # bytes of one rank's BGR8 shard of one frame
    comm = None
    comm_kind = "none"
    if tiles_mode:
        params_tile = rta.Params(background=(200, 200, 200), frame_format=1, tile_offset=rank, tile_stride=tw)
        NSET = S if batch_launch else 2       # buffer sets = sub-batches in flight (batched launches: one per stream / hardware queue)
        send = [torch.zeros(G * blk, dtype=torch.uint8, device=dev) for _ in range(NSET)]
        recv = [torch.zeros(tw * G * blk, dtype=torch.uint8, device=dev) for _ in range(NSET)] if not emulate else None
        frames = [torch.zeros((G, npix * 3), dtype=torch.uint8, device=dev) for _ in range(NSET)]
        gather_done = [torch.cuda.Event() for _ in range(NSET)]
        if world > 1:
            if args.comm == "abi" and not D.rehearsal:
                comm = make_abi_comm(rta, ctx, D)
            comm_kind = "rt_hip_comm_all_gather (RCCL)" if comm is not None else ("torch.distributed all_gather_into_tensor (%s)" % args.backend)
    else:
        frames = [torch.zeros((max(2 * S, 2), npix * 3), dtype=torch.uint8, device=dev)]
    BF = args.batch_frames if (not tiles_mode and world == 1) else 0
    if BF > 1:
        while F % BF:
            BF -= 1
        G = BF
        frames = [torch.zeros((S, BF, npix * 3), dtype=torch.uint8, device=dev)]

    def gather(set_i):
        """all-gather sub-batch `set_i` and put its G frames into raster order, on the comm stream"""
        for st in streams:
            comm_stream.wait_stream(st)
        if emulate:
            return
        with torch.cuda.stream(comm_stream):
            if comm is not None:
                rta.check(rta.hip.rt_hip_comm_all_gather(comm, C.c_void_p(send[set_i].data_ptr()), C.c_void_p(recv[set_i].data_ptr()), G * blk,
                                                         C.c_void_p(comm_stream.cuda_stream)))
            elif D.rehearsal:                       # gloo: host tensors only
                comm_stream.synchronize()
                h_send = send[set_i].cpu(); h_recv = torch.empty(tw * G * blk, dtype=torch.uint8)
                D.dist.all_gather_into_tensor(h_recv, h_send)
                recv[set_i].copy_(h_recv)
            else:
                D.dist.all_gather_into_tensor(recv[set_i], send[set_i])
            # the G frames of the sub-batch into raster order, one launch
            rta.check(rta.hip.rt_hip_tiles_to_frames_dev(cam0.raw, tw, C.c_void_p(recv[set_i].data_ptr()), G * blk, 3, G, blk,
                                                         C.c_void_p(frames[set_i].data_ptr()), npix * 3, C.c_void_p(comm_stream.cuda_stream)))
            gather_done[set_i].record(comm_stream)

    cam_arrays = {}

    def cam_array(f0):
        """the cameras of the sub-batch that starts at frame f0, as rt_hip_render_whitted_batch_dev takes them"""
        key = f0 % 32
        if key not in cam_arrays:
            cam_arrays[key] = rta.camera_array([cams[(key + i) % 32] for i in range(G)])
        return cam_arrays[key]

    def one_step(step):
        for j in range(F):
            f = step * F + j
            k = f % S
            if tiles_mode and batch_launch:
                # the rank's tiles of a whole sub-batch in ONE launch (rt_hip_render_whitted_batch_dev): G x 1/N of a frame fills the GPU like G/N frames
                if f % G:
                    continue
                sb = f // G
                set_i, k = sb % NSET, sb % S
                streams[k].wait_event(gather_done[set_i])
                rta.check(rta.hip.rt_hip_render_whitted_batch_dev(handles[k], cam_array(f), G, params_tile, C.c_void_p(send[set_i].data_ptr()), blk, None, 0,
                                                                  C.c_void_p(streams[k].cuda_stream)))
                gather(set_i)
            elif BF > 1:
                if f % BF == 0:                # BF whole frames in one launch, the frames' copies of a super-tile side by side on one XCD
                    k = (f // BF) % S
                    rta.check(rta.hip.rt_hip_render_whitted_batch_dev(handles[k], cam_array(f), BF, params_full, C.c_void_p(frames[0][k].data_ptr()), npix * 3, None, 0,
                                                                      C.c_void_p(streams[k].cuda_stream)))
            elif tiles_mode:
                set_i, g = (f // G) & 1, f % G
                if g < S:
                    streams[k].wait_event(gather_done[set_i])      # the gather that last read this buffer set has finished
                rta.check(rta.hip.rt_hip_render_whitted_dev(handles[k], cams[f % 32].raw, params_tile, C.c_void_p(send[set_i].data_ptr() + g * blk), None,
                                                            C.c_void_p(streams[k].cuda_stream)))
                if g == G - 1:
                    gather(set_i)
            else:
                fr = (f * world + rank) if world > 1 else f        # replicas: every rank its own frames
                rta.check(rta.hip.rt_hip_render_whitted_dev(handles[k], cams[fr % 32].raw, params_full, C.c_void_p(frames[0][f % frames[0].shape[0]].data_ptr()), None,
                                                            C.c_void_p(streams[k].cuda_stream)))

    # ---- rays per frame: the census of the kernels themselves, taken for every one of the 32 cameras the timed frames cycle through ----
    cam_rays = []
    for cm in cams:
        if tiles_mode:
            rta.check(rta.hip.rt_hip_render_whitted_dev(raw_scene, cm.raw, params_tile, C.c_void_p(send[0].data_ptr()), None, None))
        else:
            rta.check(rta.hip.rt_hip_render_whitted_dev(raw_scene, cm.raw, params_full, C.c_void_p(frames[0][0].data_ptr()), None, None))
        torch.cuda.synchronize()
        cam_rays.append(float(last_rays(rta, raw_scene)[0]))
    my_rays = cam_rays[0]
    if tiles_mode and not emulate:
        cam_rays = [D.reduce(x, "sum") for x in cam_rays]
    elif tiles_mode:
        cam_rays = [x * tw for x in cam_rays]
    rays_per_frame = sum(cam_rays) / len(cam_rays)           # mean over the cameras (reported); the rate below sums the frames actually timed

    for w in range(args.warmup):
        one_step(w)
    D.barrier()
    t_start = time.perf_counter()
    for s in range(args.steps):
        one_step(s)
    if comm_stream is not None:
        stream0.wait_stream(comm_stream)
    D.barrier()
    elapsed = D.reduce(time.perf_counter() - t_start, "max")
    frames_total = F * args.steps * (world if (world > 1 and not tiles_mode) else 1)
    # outside the timed region: the last sub-batch's gathered pictures against this rank's own whole-frame renders of the same cameras, byte for byte
    # (what the exchange is for: every rank holds the finished picture)
    gathered_check = None
    if tiles_mode and not emulate and args.steps > 0:
        torch.cuda.synchronize()
        last_sb = (F * args.steps) // G - 1
        set_i = last_sb % (NSET if batch_launch else 2)
        bad = 0
        full = torch.zeros(npix * 3, dtype=torch.uint8, device=dev)
        picks = sorted(set([0, G // 2, G - 1]))
        for g_ in picks:
            rta.check(rta.hip.rt_hip_render_whitted_dev(raw_scene, cams[(last_sb * G + g_) % 32].raw, params_full, C.c_void_p(full.data_ptr()), None, None))
            torch.cuda.synchronize()
            bad += int((frames[set_i][g_] != full).sum().item())
        bad = int(D.reduce(float(bad), "sum"))
        gathered_check = {"frames_compared_per_rank": len(picks), "mismatched_bytes_all_ranks": bad,
                          "what": "the all-gathered Vec3b pictures of the last sub-batch against whole-frame renders of the same cameras on every rank"}
    if rank != 0:
        return None
    ms_per_step = elapsed * 1e3 / args.steps
    # the rays of the frames that were timed: frame f of a rank uses camera f % 32 (replicas: (f x world + rank) % 32, every camera alike in the mean)
    if world > 1 and not tiles_mode:
        rays_timed = rays_per_frame * frames_total
    else:
        rays_timed = sum(cam_rays[f % 32] for f in range(F * args.steps))
    value = rays_timed / elapsed / 1e6
    scaling = "strong" if (tiles_mode or world == 1) else "weak"

    out = {"metric": "Mrays/sec (primary+secondary)", "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
           "scaling": scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "C2: synthetic %d-triangle heightfield + ground plane, primary+shadow ray casting %dx%d; a step = %d frames (camera swinging <= 1.8 deg around "
                                  "the C2 view), each to the Vec3b picture of Camera.h:69" % (n_tris, Wd, Ht, F),
                      "triangles": n_tris, "resolution": [Wd, Ht], "frames_per_step": F, "rays_per_frame": int(rays_per_frame), "rays_per_frame_definition": "mean of the kernels' own census over the 32 cameras of the orbit; value = the timed frames' rays, camera by camera, / time",
                      "ms_per_frame": round(ms_per_step / F, 4), "frames_in_flight": S, "frames_per_launch": (BF if BF > 1 else (G if batch_launch else 1)),
                      "parallelism": ("one frame split over %d ranks by interleaved 128x64 super-tiles, %d-frame all-gather: %s; %s" % (tw, G, comm_kind, "the rank's tiles of a sub-batch in one launch (rt_hip_render_whitted_batch_dev)" if batch_launch else "one launch per frame share")) if tiles_mode
                                     else ("replicas x%d (no collective)" % world if world > 1 else "1 GPU"),
                      "leaf_refine": "library default (own hierarchy, leaves < 20; < 8 for scenes of at most 4096 objects)" if args.leaf_refine is None else args.leaf_refine,
                      "scene_build_s": round(build_s, 2)}}
    if gathered_check is not None:
        out["gathered_frame_check"] = gathered_check
    if emulate:
        out["config"]["emulated"] = "rank 0's share of a %d-rank run on one GPU, no collective: value = this rank's rate x %d (a projection, not a result)" % (tw, tw)
    if world > 1 or emulate:
        # the dominant kernel of THIS rank's share, timed live with HIP events on its stream (one launch in flight); its instruction count is
        # the full frame's SQ_INSTS_VALU (profiles/r02_traffic.json) scaled by the rank's share of the rays.  The CPU leg belongs to N = 1.
        if tiles_mode:
            sptr = C.c_void_p(stream0.cuda_stream)
            torch.cuda.synchronize()
            rta.check(rta.hip.rt_hip_set_profiling(raw_scene, 1))
            prof = rta.Profile()
            rta.check(rta.hip.rt_hip_get_profile(raw_scene, C.byref(prof), 1))
            for s_ in range(20):
                if batch_launch:       # one launch = this rank's shares of G frames: the per-share time is the launch's / G
                    if s_ < 6:
                        rta.check(rta.hip.rt_hip_render_whitted_batch_dev(raw_scene, cam_array(s_ * G), G, params_tile, C.c_void_p(send[0].data_ptr()), blk, None, 0, sptr))
                else:
                    rta.check(rta.hip.rt_hip_render_whitted_dev(raw_scene, cams[s_ % 32].raw, params_tile, C.c_void_p(send[0].data_ptr()), None, sptr))
            torch.cuda.synchronize()
            rta.check(rta.hip.rt_hip_get_profile(raw_scene, C.byref(prof), 1))
            rta.check(rta.hip.rt_hip_set_profiling(raw_scene, 0))
            kms = {rta.KERNEL_NAMES[i]: prof.ms[i] / prof.launches[i] / (G if batch_launch else 1) for i in range(12) if prof.launches[i]}
            dom = max(kms, key=lambda k: kms[k])
            pmc = read_traffic(dom)
            share = float(my_rays) / max(1.0, rays_per_frame)
            scaled = dict(pmc)
            if pmc.get("SQ_INSTS_VALU"):
                scaled["SQ_INSTS_VALU"] = pmc["SQ_INSTS_VALU"] * share
            scaled["sq"] = {k: v * share for k, v in pmc.get("sq", {}).items()}
            ir = issue_roofline(scaled, kms[dom])
            out["roofline"] = {"bound": "valu_issue", "kernel": dom, "achieved": ir.get("achieved"), "peak": ir.get("peak"), "unit": "G wave-instructions/s",
                               "frac": ir.get("frac"), "traffic": pmc.get("hbm_bytes_per_launch") * share if pmc.get("hbm_bytes_per_launch") else None,
                               "valu_f64_share": ir.get("valu_f64_share"), "salu_issue": ir.get("salu_issue"), "counters_stale": counters_stale(),
                               "kernel_ms": round(kms[dom], 4), "rays_per_launch": int(my_rays), "share_of_frame": round(share, 4),
                               "note": "rank 0's tile share of a frame, one launch in flight, HIP events on its stream; instruction and traffic counts = the one-GPU frame's "
                                       "PMC figures x this rank's share of the rays (the counters were collected on the full frame)"}
        return out

    # =========================== one GPU: roofline, secondary figures, CPU baseline ===========================
    # ---- kernel time with ONE frame in flight, HIP events on the kernel's own stream (rt_hip_set_profiling) ----
    rgb64 = torch.zeros(npix * 3, dtype=torch.float64, device=dev)
    prim = torch.zeros(npix, dtype=torch.int32, device=dev)
    sptr = C.c_void_p(stream0.cuda_stream)
    rta.check(rta.hip.rt_hip_set_profiling(raw_scene, 1))
    prof = rta.Profile()
    rta.check(rta.hip.rt_hip_get_profile(raw_scene, C.byref(prof), 1))
    n_one = 20
    t1 = time.perf_counter()
    for s_ in range(n_one):
        rta.check(rta.hip.rt_hip_render_whitted_dev(raw_scene, cams[s_ % 32].raw, params_full, C.c_void_p(frames[0][0].data_ptr()), None, sptr))
    torch.cuda.synchronize()
    ms_one_in_flight = (time.perf_counter() - t1) * 1e3 / n_one
    rta.check(rta.hip.rt_hip_get_profile(raw_scene, C.byref(prof), 1))
    rta.check(rta.hip.rt_hip_set_profiling(raw_scene, 0))
    kms = {rta.KERNEL_NAMES[i]: prof.ms[i] / prof.launches[i] for i in range(12) if prof.launches[i]}
    dom = max(kms, key=lambda k: kms[k])
    dom_ms = kms[dom]

    # ---- per-ray work of the traversal actually performed (statistics build of the same kernel) ----
    pstat = rta.Params(background=(200, 200, 200), collect_stats=1)
    ctx.stats(reset=True)
    rta.check(rta.hip.rt_hip_render_whitted_dev(raw_scene, cam0.raw, pstat, C.c_void_p(rgb64.data_ptr()), C.c_void_p(prim.data_ptr()), sptr))
    torch.cuda.synchronize()
    st = ctx.stats(reset=True)
    rays_dom, shadow_rays = last_rays(rta, raw_scene)
    nrays = max(1, st["rays"])
    nodes_per_ray, prims_per_ray, exact_per_ray = st["node_visits"] / nrays, st["prim_tests"] / nrays, st["exact_tests"] / nrays
    wave_boxes_per_ray, wave_recs_per_ray = st["wave_node_visits"] / nrays, st["wave_prim_tests"] / nrays
    # Algorithmic bytes, two definitions.  PACKET: a wave of 64 rays walks the tree together, so a
    # box test's 32 B (half a 64-B child pair of FP32 boxes; a leaf header, one box test in ten, is counted the same) and a cull record's 128 B move once per WAVE that needs
    # them; the 96-B exact triangle of a confirmation is per ray; the frame is 3 B per pixel = per two rays.  LANE: what 64
    # independent rays would fetch (SURVEY 8d's per-ray formula on this traversal's counts) -- served from LDS / SGPR
    # broadcasts, not from HBM.
    io_bytes = 1.5
    bytes_per_ray = io_bytes + 32 * wave_boxes_per_ray + 128 * wave_recs_per_ray + 96 * exact_per_ray
    lane_bytes_per_ray = io_bytes + 32 * nodes_per_ray + 128 * prims_per_ray + 96 * exact_per_ray
    flops_per_ray = 45 * prims_per_ray + 150 * exact_per_ray      # FP64 only: the box tests (40 flop each) run in FP32
    ach_gbs = bytes_per_ray * rays_dom / (dom_ms * 1e-3) / 1e9
    tf = flops_per_ray * rays_dom / (dom_ms * 1e-3) / 1e12
    pmc = read_traffic(dom)
    valu_insts = pmc.get("SQ_INSTS_VALU")                  # wave-level VALU instructions per launch (rocprofv3 --pmc pass)
    cycles = pmc.get("kernel_cycles")                      # GRBM_GUI_ACTIVE per launch / 8 XCDs
    ir = issue_roofline(pmc, dom_ms)
    # USEFUL work: SURVEY 8(d)'s algorithmic weights on this traversal's own counts -- 40 flop per box test, 85 per triangle test (cull records and
    # exact confirmations alike) -- against the FP64 vector peak.  What the issue fraction has beyond this is overhead (profiles/README.md itemises it).
    useful_flops_per_ray = 40.0 * nodes_per_ray + 85.0 * (prims_per_ray + exact_per_ray)
    useful_tf = useful_flops_per_ray * rays_dom / (dom_ms * 1e-3) / 1e12
    roofline = {"bound": "valu_issue", "kernel": dom, "achieved": ir.get("achieved"), "peak": ir.get("peak"),
                "unit": "G wave-instructions/s", "frac": ir.get("frac"),
                "valu_f64_share": ir.get("valu_f64_share"), "valu_other_share": ir.get("valu_other_share"), "peak_definition": ir.get("peak_definition"),
                "frac_if_all_4_cycles": ir.get("frac_if_all_4_cycles"), "frac_if_all_2_cycles": ir.get("frac_if_all_2_cycles"),
                "salu_issue": ir.get("salu_issue"),
                "useful": {"frac": round(useful_tf / FP64_VALU_PEAK_TF, 4), "achieved_tflops": round(useful_tf, 2), "peak_tflops": FP64_VALU_PEAK_TF,
                           "flops_per_ray": round(useful_flops_per_ray, 1),
                           "definition": "SURVEY 8(d) weights on the counts of this traversal: (40 x box tests + 85 x (cull records + confirmations)) x rays / kernel time / 78.6 TF FP64 vector"},
                "counters_stale": counters_stale(), "counters_file": os.path.relpath(traffic_file(), ROOT),
                "traffic": pmc.get("hbm_bytes_per_launch"),
                "kernel_ms": round(dom_ms, 4), "kernel_ms_source": "HIP events on the kernel's stream, one frame in flight, %d launches" % n_one,
                "rays_per_launch": int(rays_dom), "valu_wave_insts_per_launch": valu_insts,
                "valu_lane_insts_per_ray": round(valu_insts * 64.0 / rays_dom, 1) if valu_insts else None,
                "frac_at_measured_clock_all_4_cycles": round(valu_insts * 4.0 / N_SIMD / cycles, 4) if (valu_insts and cycles) else None,
                "note": "the walk is bound by instruction issue (FP32 box tests, FP64 cull and confirmation + the integer, move and select work around them), not by HBM: the "
                        "scene (0.2 GB) is L2 / Infinity-Cache resident.  Counters: rocprofv3 --pmc passes of this command kept in profiles/ (counters_stale says whether "
                        "they are of the library timed here); time: live",
                "hbm": {"achieved": round(ach_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach_gbs / HBM_PEAK_GBS, 4),
                        "bytes_per_ray": round(bytes_per_ray, 1), "bytes_definition": "packet: node / record bytes once per wave that needs them",
                        "lane_bytes_per_ray": round(lane_bytes_per_ray, 1), "lane_bytes_definition": "what 64 independent rays would fetch (served by LDS / SGPR broadcasts)"},
                "valu_fp64": {"achieved_tflops": round(tf, 2), "peak_tflops": FP64_VALU_PEAK_TF, "frac": round(tf / FP64_VALU_PEAK_TF, 4), "flops_per_ray": round(flops_per_ray, 1),
                              "definition": "the FP64 flops this implementation issues (45 per cull record, 150 per confirmation; its box tests run in FP32)"},
                "box_tests_per_ray": round(nodes_per_ray, 2), "cull_records_per_ray": round(prims_per_ray, 2), "exact_tests_per_ray": round(exact_per_ray, 3),
                "wave_box_fetches_per_ray": round(wave_boxes_per_ray, 3), "wave_record_fetches_per_ray": round(wave_recs_per_ray, 3),
                "all_kernels_ms": {k: round(v, 4) for k, v in kms.items()}}
    out["roofline"] = roofline
    out["config"]["ms_per_frame_one_in_flight"] = round(ms_one_in_flight, 4)
    out["config"]["shadow_rays_per_frame"] = int(shadow_rays)

    def timed_frames(handle_list, params, bufs, n, ids=None):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for f in range(n):
            k = f % len(handle_list)
            rta.check(rta.hip.rt_hip_render_whitted_dev(handle_list[k], cams[f % 32].raw, params, C.c_void_p(bufs[k].data_ptr()),
                                                        C.c_void_p(ids[k].data_ptr()) if ids else None, C.c_void_p(streams[k].cuda_stream)))
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n

    if not args.no_extras:
        # ---- secondary: the f64 R,G,B + prim-id frame of round 1's line; the reference's own BSP tree on the device ----
        b64 = [rgb64] + [torch.zeros(npix * 3, dtype=torch.float64, device=dev) for _ in range(S - 1)]
        bid = [prim] + [torch.zeros(npix, dtype=torch.int32, device=dev) for _ in range(S - 1)]
        dt = timed_frames(handles, params_f64, b64, 40, bid)
        out["config"]["f64_rgb_plus_ids_frame_mrays"] = round(rays_per_frame / dt / 1e6, 1)
        del b64, bid
        t_r = time.time()
        ref_scene = W.build_heightfield(W.product_stage, args.cells, leaf_refine=0xFFFFFFFF).sides[0][0]      # RT_HIP_LEAF_REFINE_NONE
        ref_raw = ref_scene.device(ctx)
        ref_share = [ref_scene.share(ctx) for _ in range(S - 1)]
        bufs8 = [frames[0][k % frames[0].shape[0]] for k in range(S)]
        timed_frames([ref_raw] + [r.raw for r in ref_share], params_full, bufs8, 4)
        dt = timed_frames([ref_raw] + [r.raw for r in ref_share], params_full, bufs8, 24)
        out["config"]["reference_tree_mrays"] = round(rays_per_frame / dt / 1e6, 1)
        out["config"]["reference_tree_note"] = "RT_HIP_LEAF_REFINE_NONE: the device walks BSPTree.h's own tree (leaves < 100 objects, 1.85 M references)"
        del ref_share, ref_scene
        # the allocation-free entry point on the same frame
        torch.cuda.synchronize(); t = time.perf_counter()
        for f in range(20):
            rta.check(rta.hip.rt_hip_render_whitted_noalloc(raw_scene, cams[f % 32].raw, params_full, C.c_void_p(frames[0][0].data_ptr()), None, sptr))
        torch.cuda.synchronize()
        out["config"]["noalloc_entry_point_ms_per_frame"] = round((time.perf_counter() - t) * 1e3 / 20, 4)

    # ---- CPU baseline + whole-frame parity: the oracle (reference semantics, exhaustive BSP walk) on the host cores ----
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import numpy as np
        import scenes
        import parity_util
        _, oscene = scenes.build_heightfield(args.cells, gpu=False, oracle=True)
        ocam = scenes.oracle_camera(W.C2_CAMERA)
        threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
        t1 = time.perf_counter()
        ref_rgb, ref_ids, cnt = oscene.render_whitted(ocam, background=(200, 200, 200), threads=threads, counters=True)
        dt = time.perf_counter() - t1
        cpu_rays = int(cnt[0])
        out["cpu_baseline"] = {"value": round(cpu_rays / dt / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
                               "sample": "the whole 1920x1080 C2 frame once (%d rays, %.1f s), oracle/rt_oracle.c with OpenMP" % (cpu_rays, dt),
                               "node_visits_per_ray": round(float(cnt[1]) / max(1, cpu_rays), 1), "prim_tests_per_ray": round(float(cnt[2]) / max(1, cpu_rays), 1)}
        rta.check(rta.hip.rt_hip_render_whitted_dev(raw_scene, cam0.raw, params_f64, C.c_void_p(rgb64.data_ptr()), C.c_void_p(prim.data_ptr()), sptr))
        torch.cuda.synchronize()
        g_rgb, g_ids = rgb64.cpu().numpy().reshape(-1, 3), prim.cpu().numpy()
        rays_np, _, _ = oscene.camera_rays(ocam)

        def t_of(idx):
            hits = gscene.first_inter_pt(ctx, rays_np[idx])
            _, pts = oscene.trace_closest(rays_np[idx], use_bsp=True)
            return hits["t"], np.linalg.norm(pts - rays_np[idx, :3], axis=1)
        out["parity"] = parity_util.frame_parity(g_rgb, g_ids, ref_rgb, ref_ids, t_of, two_class_ids=lambda i: np.where(i < 0, -1, np.minimum(i, 1)))
        out["parity"]["what"] = "all primary hit ids, shadow bits and radiance of the C2 frame, HIP vs oracle; bar: tie_fraction <= 1e-4, max_dt < 1e-6, colour < 1e-3"
        del oscene
    return out


# ---------------------------------------------------------------------------------------------------------------------
def other_config(name, args, D, torch, rta, W, ctx, short, once=False):
    """C3 / C4 / C5 on their stated geometry at 1080p.  `short`: a cut-down, extrapolated run; otherwise BASELINE's sizes (1024 spp, 256 spp,
    10 M photons per pass) unless --spp / --photons say otherwise.  once: one timed repetition (the default line's other_configs).
    N > 1: C3 splits the samples (all-reduce of the un-clamped sums), C5 the photons (all-reduce of the per-hit-point statistics), C4 the
    rows (all-gather)."""
    from ray_tracing_amd import parallel
    rank, world, dev = D.rank, D.world, D.dev
    res = {}
    if name == "C3":
        g = W.build_icosahedron(W.product_stage, material="plastic", light="area10").sides[0][0]
        cam = rta.Camera(**W.C3_CAMERA)
        n = cam.width * cam.height
        spp_total = 1024
        spp_run = args.spp or (2 if short else spp_total)
        rgb = torch.zeros(n * 3, dtype=torch.float64, device=dev)
        s0, cnt = parallel.sample_shard(spp_run, rank, world)
        p = rta.Params(background=(200, 200, 200), spp=spp_total, lens_radius=0.1, sample_first=s0, sample_count=max(cnt, 1), seed=1)
        raw = g.device(ctx)

        def step():
            if cnt:
                rta.check(rta.hip.rt_hip_render_focusing_dev(raw, cam.raw, p, C.c_void_p(rgb.data_ptr()), None))
            if world > 1:
                if D.rehearsal:
                    h = rgb.cpu(); D.dist.all_reduce(h); rgb.copy_(h)
                else:
                    D.dist.all_reduce(rgb)
        reps = 1 if (short or once or spp_run >= 256) else max(1, args.steps)
        if not (once and not short):          # (a full-size frame inside the default line is rendered once: its own first call is the timed one)
            step(); torch.cuda.synchronize()
        else:
            p1 = rta.Params(background=(200, 200, 200), spp=spp_total, lens_radius=0.1, sample_first=0, sample_count=1, seed=1)
            rta.check(rta.hip.rt_hip_render_focusing_dev(raw, cam.raw, p1, C.c_void_p(rgb.data_ptr()), None)); torch.cuda.synchronize()     # workspace allocation
        D.barrier(); t = time.perf_counter()
        for _ in range(reps):
            step()
        D.barrier()
        dt = D.reduce((time.perf_counter() - t) / reps, "max")
        rays = D.reduce(last_rays(rta, raw)[0] if cnt else 0, "sum")
        pv = C.c_uint64(0)
        if cnt:
            rta.check(rta.hip.rt_hip_last_proved_shadow_rays(raw, C.byref(pv)))
        proved = D.reduce(pv.value, "sum")
        res = {"workload": "C3: Layout3_Icosahedron, depth of field (Camera.h:171-199) %dx%d, 10x10 area light (100 shadow rays per lens sample), lens radius 0.1" % (cam.width, cam.height),
               "samples_per_pixel_timed": spp_run, "rays": int(rays), "s_per_step": round(dt, 5), "Mrays_per_s": round(rays / dt / 1e6, 1),
               "s_per_frame_at_1024spp": round(dt / spp_run * spp_total, 2), "extrapolated": spp_run != spp_total,
               "parallelism": "sample split + all-reduce(sum) of the f64 frame" if world > 1 else "1 GPU",
               # `rays` = the rays the reference makes for this frame (primary + every light sample of every hit).  Of them, the shadow rays of points
               # that area_light_all_clear proves unoccluded are counted but not traced:
               "shadow_rays_proved_clear": int(proved), "rays_traced": int(rays - proved), "Mrays_per_s_traced": round((rays - proved) / dt / 1e6, 1)}
        kernel = "k_shadow"
    elif name == "C4":
        g = W.build_paper(W.product_stage).sides[0][0]
        assert g.num_objects == 11578, g.num_objects          # plane + patch + WallLight.obj's 11,576 triangles
        cam = rta.Camera(**W.C4_CAMERA)
        assert (cam.width, cam.height) == (1920, 1080)
        n = cam.width * cam.height
        spp_total = 256
        spp_run = args.spp or (4 if short else spp_total)
        aa = rta.AaParams(oversample_times=spp_run, depth_grad_threshold=-1.0)       # every pixel re-sampled: the fixed-spp jittered render
        first, count = parallel.row_shard(cam.height, cam.width, rank, world)
        p = rta.Params(background=(30, 30, 30), seed=1, pixel_first=first, pixel_count=count)
        rgb = torch.zeros(max(count, 1) * 3, dtype=torch.float64, device=dev)
        raw = g.device(ctx)

        def step():
            if count:
                rta.check(rta.hip.rt_hip_render_antialiasing_dev(raw, cam.raw, p, aa, C.c_void_p(rgb.data_ptr()), None, None))
            if world > 1 and not D.rehearsal:
                parallel.gather_rows(rgb.view(-1, 3)[:count], cam.height, cam.width, 3)
        reps = 1 if (short or once or spp_run >= 64) else max(1, min(args.steps, 5))
        if not (once and not short):
            step(); torch.cuda.synchronize()
        else:
            aa1 = rta.AaParams(oversample_times=1, depth_grad_threshold=-1.0)
            rta.check(rta.hip.rt_hip_render_antialiasing_dev(raw, cam.raw, p, aa1, C.c_void_p(rgb.data_ptr()), None, None)); torch.cuda.synchronize()   # first-call set-up
        D.barrier(); t = time.perf_counter()
        for _ in range(reps):
            step()
        D.barrier()
        dt = D.reduce((time.perf_counter() - t) / reps, "max")
        rays = D.reduce(last_rays(rta, raw)[0] if count else 0, "sum")
        res = {"workload": "C4: Layout2_WallAndPaper (Bezier patch + WallLight.obj, %d objects) %dx%d depth 2, pixel-jittered (r0 0.007), every pixel re-sampled" % (g.num_objects, cam.width, cam.height),
               "samples_per_pixel_timed": spp_run, "rays": int(rays), "s_per_step": round(dt, 5), "Mrays_per_s": round(rays / dt / 1e6, 1),
               "s_per_frame_at_256spp": round(dt / spp_run * spp_total, 2), "extrapolated": spp_run != spp_total,
               "parallelism": "row split + all-gather" if world > 1 else "1 GPU"}
        kernel = ["k_newton_bez", "k_primary_list"]       # deferred Newton solves (the default) / the in-place kernels (RT_HIP_PARAM_DEFER=0)
    else:
        g = W.build_wine(W.product_stage, cups="layout", light="area5").sides[0][0]
        assert g.num_objects == 24
        cam = rta.Camera(**W.C5_CAMERA)
        photon_num = args.photons or (20000 if short else 400000)                   # x 25 light samples per pass (the config: 400000 -> 10 M)
        t = time.perf_counter()
        pm = rta.PhotonMap(ctx, g, cam, photon_num=photon_num, times=64, background=(10, 10, 10), seed=1)
        torch.cuda.synchronize()
        t_eye = time.perf_counter() - t
        eye_rays = last_rays(rta, g.device(ctx))[0]
        info = pm.info()
        passes = 1 if (short or once) else max(1, min(args.steps, 4))
        D.barrier(); t = time.perf_counter()
        for k in range(passes):
            if world > 1 and not D.rehearsal:
                parallel.ppm_pass_sharded(pm, k)
            else:
                pm.run_pass(k)
        D.barrier()
        dt = D.reduce((time.perf_counter() - t) / passes, "max")
        i2 = pm.info()
        segs = D.reduce(i2["photon_segments"], "sum")
        res = {"workload": "C5: Layout5_Wine (three cups, uRange 0.7/0.5/0.8, 24 objects, no tree) %dx%d depth 8, 5x5 area light, progressive photon mapping" % (cam.width, cam.height),
               "eye_pass_s": round(t_eye, 3), "eye_pass_rays": int(eye_rays), "hit_points": info["hitpoints"], "photons_per_pass_timed": info["photons_per_pass"],
               "passes_timed": passes, "s_per_step": round(dt, 4), "photon_segments_per_pass": int(segs / passes), "rays": int(segs / passes),
               "Mrays_per_s": round(segs / passes / dt / 1e6, 2), "s_per_pass_at_10M_photons": round(dt * 1e7 / info["photons_per_pass"], 3),
               "extrapolated": info["photons_per_pass"] != 10000000, "parallelism": "photon split + all-reduce of (M, tau) per hit point" if world > 1 else "1 GPU"}
        kernel = ["k_newton_rot", "k_ppm_bounce"]        # the solver's launches of eye pass and photon passes / the in-place per-bounce kernels
    if isinstance(kernel, str):
        kernel = [kernel]
    if os.environ.get("RT_HIP_PARAM_DEFER", "") == "0":
        kernel = kernel[-1:]
    pmc = {}
    for kname in kernel:
        pmc = read_traffic(kname + ":" + name) or read_traffic(kname)
        if pmc:
            kernel = kname
            break
    else:
        kernel = kernel[0]
    if pmc.get("SQ_INSTS_VALU") and pmc.get("kernel_cycles"):
        # (both figures from the same profiled run: instructions and GRBM cycles of the kernel; the mix weights the cycles per instruction)
        share = pmc.get("valu_f64_share")
        cyc = (4.0 * share + 2.0 * (1.0 - share)) if share is not None else 4.0
        res["dominant_kernel"] = kernel
        res["valu_issue_frac"] = round(pmc["SQ_INSTS_VALU"] * cyc / N_SIMD / pmc["kernel_cycles"], 4)
        res["valu_f64_share"] = round(share, 4) if share is not None else None
        res["valu_issue_frac_if_all_4_cycles"] = round(pmc["SQ_INSTS_VALU"] * 4.0 / N_SIMD / pmc["kernel_cycles"], 4)
    res["_scene"] = g
    return res


def config_checks(name, rta, ctx, g, threads):
    """parity of config `name` against the oracle on a bounded sample of the same frame, and the oracle's rate on that sample (the config's
    cpu_baseline): tests/config_parity.py, the comparisons the GPU tests assert on.  Outside every timed region."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import config_parity as cp
    import scenes
    if name == "C3":
        _, o = scenes.build_icosahedron(material="plastic", light="area10", gpu=False)
        par, cpu = cp.c3_rows(rta, ctx, g, o, rows=tuple(range(2, 1080, 8)), spp=2, seed=1, threads=threads)
    elif name == "C4":
        _, o = scenes.build_paper(gpu=False)
        par, cpu = cp.c4_rows(rta, ctx, g, o, rows=tuple(range(100, 1000, 28)), threads=threads)
    else:
        _, o = scenes.build_wine(gpu=False, cups="layout", light="area5")
        par, cpu = cp.c5_eye_rows(rta, ctx, g, o, rows=(300, 420, 540, 650, 760, 880), threads=threads)
        ppm, cpu2 = cp.c5_photon_map(rta, ctx, g, o, res=24, photon_num=1500, passes=2)
        par = {"eye_pass": par, "photon_map": ppm}
        cpu = {"rays": cpu["rays"] + cpu2["rays"], "seconds": cpu["seconds"] + cpu2["seconds"]}
    base = {"value": round(cpu["rays"] / max(cpu["seconds"], 1e-9) / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": "the parity sample of this config (parity.sample): %d rays, %.2f s, oracle/rt_oracle.c with OpenMP" % (cpu["rays"], cpu["seconds"])}
    return par, base


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))

    # an eighth of a frame fills the GPU's wave slots once and lasts as long as its slowest tile.  From 8 ranks on a rank renders its tiles of a whole
    # sub-batch in one launch (rt_hip_render_whitted_batch_dev; --emulate-world 8 at HIP's default 4 hardware queues: 60.6 Grays/s projected against 48.3
    # with a launch per frame share); only --no-batch-launch still asks for 8 frames in flight on 8 hardware queues (61.6)
    if max(int(os.environ.get("WORLD_SIZE", "1")), args.emulate_world) >= 8 and args.no_batch_launch:
        if int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) < 8:
            os.environ["GPU_MAX_HW_QUEUES"] = "8"
    import torch
    import ray_tracing_amd as rta
    from ray_tracing_amd import workloads as W
    W.set_asset_dir(os.path.join(ROOT, "tests", "golden", "assets"))      # the layouts' meshes: the repository's data fixtures

    D = Dist(args, torch)
    if D.world != args.gpus:
        args.gpus = D.world
    if args.config == "C2":
        out = bench_c2(args, D, torch, rta, W)
        if out is not None and D.world == 1 and not args.emulate_world and not args.no_extras:
            ctx = rta.Context(D.local_rank)
            threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
            others = {}
            for name in ("C3", "C4", "C5"):
                try:
                    others[name] = other_config(name, args, D, torch, rta, W, ctx, short=(args.others == "short"), once=True)
                    g = others[name].pop("_scene")
                    if not args.no_cpu_baseline:
                        others[name]["parity"], others[name]["cpu_baseline"] = config_checks(name, rta, ctx, g, threads)
                    del g
                except Exception as e:          # a secondary figure must not cost the headline line
                    others[name] = {"error": repr(e)}
            out["config"]["other_configs"] = others
    else:
        ctx = rta.Context(D.local_rank)
        res = other_config(args.config, args, D, torch, rta, W, ctx, short=False)
        g = res.pop("_scene")
        out = None
        if D.rank == 0:
            out = {"metric": "Mrays/sec (primary+secondary)", "value": res["Mrays_per_s"], "unit": "Mrays/s", "n_gpus": D.world, "steps": args.steps,
                   "warmup": args.warmup, "ms_per_step": round(res["s_per_step"] * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                   "dtype": "f64", "data": "synthetic", "config": res}
            pmc_k = res.get("dominant_kernel")
            share = res.get("valu_f64_share")
            peak = valu_peak_g(share) if share is not None else VALU_ISSUE_PEAK_ALL_F64_G
            out["roofline"] = {"bound": "valu_issue", "kernel": pmc_k, "frac": res.get("valu_issue_frac"), "peak": round(peak, 1), "unit": "G wave-instructions/s",
                               "achieved": round(res["valu_issue_frac"] * peak, 1) if res.get("valu_issue_frac") else None, "traffic": None,
                               "valu_f64_share": share, "counters_stale": counters_stale(),
                               "note": "from the rocprofv3 --pmc passes of this config kept in %s: SQ_INSTS_VALU x (4 f64_share + 2 (1 - f64_share)) cycles / 1024 SIMDs / "
                                       "the kernel's GRBM cycles in the same profiled run" % os.path.relpath(traffic_file(), ROOT)}
            if not args.no_cpu_baseline and D.world == 1:
                threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
                out["parity"], out["cpu_baseline"] = config_checks(args.config, rta, ctx, g, threads)
    if D.rank == 0 and out is not None:
        print(json.dumps(out), flush=True)
    D.finish()


if __name__ == "__main__":
    main()
