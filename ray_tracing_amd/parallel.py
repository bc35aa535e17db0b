"""Multi-GPU sharding of the render calls: one process per GPU (torch.distributed; backend "nccl" is RCCL
on ROCm, "gloo" in the CPU tests).  The scene is replicated; a frame is split

  * by image rows (deterministic renders): every rank renders its row block through the ABI's pixel shard
    (bit-identical to the single-GPU frame) and the blocks are all-gathered;
  * by samples (depth of field / stochastic renders): every rank renders a slice of the spp loop; the ABI
    returns un-clamped sums of sample/spp, which add up under all-reduce(sum); min(255, .) once at the end
    equals the reference's running clamp (Colour.h:41-47) because every term is >= 0.

  * photon mapping: by photons (ppm_pass_sharded): the per-hit-point sums of a pass add up under all-reduce.

The arithmetic of the shards lives here (pure functions, covered by tests/test_parallel_cpu.py with two
gloo ranks); the rendering itself is the rt_hip_* ABI.
"""
import ctypes as C


def row_shard(height, width, rank, world):
    """rows [r0, r1) of rank -> (pixel_first, pixel_count); 8-row granularity keeps the 8x8 tiles whole"""
    tiles = (height + 7) // 8
    per, extra = divmod(tiles, world)
    t0 = rank * per + min(rank, extra)
    t1 = t0 + per + (1 if rank < extra else 0)
    r0, r1 = min(height, t0 * 8), min(height, t1 * 8)
    return r0 * width, (r1 - r0) * width


def sample_shard(spp, rank, world):
    """samples [s0, s0+n) of rank"""
    per, extra = divmod(spp, world)
    s0 = rank * per + min(rank, extra)
    return s0, per + (1 if rank < extra else 0)


def photon_shard(photons_per_pass, rank, world):
    """photons [first, first + count) of rank (rt_hip_ppm_trace's photon range)"""
    per, extra = divmod(photons_per_pass, world)
    first = rank * per + min(rank, extra)
    return first, per + (1 if rank < extra else 0)


class _DeviceArray:
    """a raw device pointer as a 1-D float64 array for torch (``__cuda_array_interface__``): no copy"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def ppm_pass_sharded(pm, pass_index, group=None, stream=None):
    """one iteration of PhotonMappingPass (PhotonMapper.h:226-238) over the ranks of `group`: every rank holds the same
    hit points (rt_hip_ppm_begin is deterministic), traces its own photon range, the per-hit-point sums (M, tau) are
    all-reduced in place in the library's accumulator buffer, and every rank applies the same update"""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    info = pm.info()
    first, count = photon_shard(info["photons_per_pass"], rank, world)
    s = stream if stream is not None else torch.cuda.current_stream()
    sp = C.c_void_p(s.cuda_stream)
    if count:
        pm.trace(pass_index, first, count, sp)
    ptr, n = pm.accumulators()
    if n:
        acc = torch.as_tensor(_DeviceArray(ptr, n), device="cuda")
        with torch.cuda.stream(s):
            dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
    pm.update(sp)


def gather_rows(local, height, width, channels, group=None):
    """all-gather of the row blocks into the full (height*width, channels) frame on every rank"""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    counts = [row_shard(height, width, r, world)[1] for r in range(world)]
    assert local.shape[0] == counts[rank]
    mx = max(counts)
    pad = torch.zeros((mx, channels), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)


def reduce_samples(local_sum, group=None):
    """all-reduce(sum) of the per-rank sample sums, then the single clamp"""
    import torch
    import torch.distributed as dist
    dist.all_reduce(local_sum, op=dist.ReduceOp.SUM, group=group)
    return torch.clamp(local_sum, max=255.0)


def render_whitted_rows(rta, ctx, scene, camera, background=(0, 0, 0), stream=None, group=None):
    """Render_RayTracer split by rows over the ranks of `group`; returns the full frame (H*W, 3) f64 on the GPU"""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    first, count = row_shard(camera.height, camera.width, rank, world)
    local = torch.empty((count, 3), dtype=torch.float64, device="cuda")
    if count:
        p = rta.Params(background=background, pixel_first=first, pixel_count=count)
        s = stream if stream is not None else torch.cuda.current_stream()
        rta.check(rta.hip.rt_hip_render_whitted_dev(scene.device(ctx), camera.raw, p, C.c_void_p(local.data_ptr()), None,
                                                    C.c_void_p(s.cuda_stream)))
    return gather_rows(local, camera.height, camera.width, 3, group)


def render_antialiasing_rows(rta, ctx, scene, camera, aa=None, background=(0, 0, 0), seed=0, spp=0, lens_radius=0.0, stream=None, group=None):
    """Render_RayTracer_Antialiasing / Render_Focusing_Antialiasing split by rows: the edge mask needs the rows next
    to a block, which the ABI renders as a halo, so the blocks again equal the single-GPU frame"""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    first, count = row_shard(camera.height, camera.width, rank, world)
    local = torch.empty((count, 3), dtype=torch.float64, device="cuda")
    if count:
        p = rta.Params(background=background, seed=seed, spp=spp, lens_radius=lens_radius, pixel_first=first, pixel_count=count)
        s = stream if stream is not None else torch.cuda.current_stream()
        rta.check(rta.hip.rt_hip_render_antialiasing_dev(scene.device(ctx), camera.raw, p, aa if aa is not None else rta.AaParams(),
                                                         C.c_void_p(local.data_ptr()), None, C.c_void_p(s.cuda_stream)))
    return gather_rows(local, camera.height, camera.width, 3, group)


def render_focusing_samples(rta, ctx, scene, camera, radius, spp, background=(0, 0, 0), seed=0, stream=None, group=None):
    """Render_Focusing split by samples over the ranks of `group`; returns the clamped frame (H*W, 3) f64"""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    s0, n = sample_shard(spp, rank, world)
    acc = torch.zeros((camera.height * camera.width, 3), dtype=torch.float64, device="cuda")
    if n:
        p = rta.Params(background=background, seed=seed, spp=spp, lens_radius=radius, sample_first=s0, sample_count=n)
        s = stream if stream is not None else torch.cuda.current_stream()
        rta.check(rta.hip.rt_hip_render_focusing_dev(scene.device(ctx), camera.raw, p, C.c_void_p(acc.data_ptr()),
                                                     C.c_void_p(s.cuda_stream)))
    return reduce_samples(acc, group)
