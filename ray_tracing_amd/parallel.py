"""Multi-GPU sharding of the render calls: one process per GPU (torch.distributed; backend "nccl" is RCCL
on ROCm, "gloo" in the CPU tests).  The scene is replicated; a frame is split

  * by interleaved tiles (deterministic renders, the default): rank r renders the 128x64-pixel super-tiles
    r, r + N, ... through the ABI's tile shard (rt_hip_params.tile_offset / tile_stride), so every rank's share is
    spread over the whole image and costs the same; the shards are all-gathered and rt_hip_tiles_to_frame_dev puts
    them back into raster order: bit-identical to the single-GPU frame;
  * by image rows (the anti-aliasing drivers, whose stencil needs neighbouring rows): every rank renders its row
    block through the ABI's pixel shard (bit-identical to the single-GPU frame) and the blocks are all-gathered;
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


def _render_stream(stream):
    """the stream a render is enqueued on, ordered after what the current stream did to its buffers (allocation, zero fill)"""
    import torch
    cur = torch.cuda.current_stream()
    if stream is None or stream == cur:
        return cur
    stream.wait_stream(cur)
    return stream


def _join(stream):
    """the current stream (where the collectives below run) waits for the render on `stream`"""
    import torch
    cur = torch.cuda.current_stream()
    if stream != cur:
        cur.wait_stream(stream)


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
        s = _render_stream(stream)
        rta.check(rta.hip.rt_hip_render_whitted_dev(scene.device(ctx), camera.raw, p, C.c_void_p(local.data_ptr()), None,
                                                    C.c_void_p(s.cuda_stream)))
    if count:
        _join(s)
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
        s = _render_stream(stream)
        rta.check(rta.hip.rt_hip_render_antialiasing_dev(scene.device(ctx), camera.raw, p, aa if aa is not None else rta.AaParams(),
                                                         C.c_void_p(local.data_ptr()), None, C.c_void_p(s.cuda_stream)))
    if count:
        _join(s)
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
        s = _render_stream(stream)
        rta.check(rta.hip.rt_hip_render_focusing_dev(scene.device(ctx), camera.raw, p, C.c_void_p(acc.data_ptr()),
                                                     C.c_void_p(s.cuda_stream)))
    if n:
        _join(s)
    return reduce_samples(acc, group)


# ---- interleaved tile shards (rt_hip_params.tile_offset / tile_stride) ----------------------------------------------
SUPER_W, SUPER_H = 128, 64            # one super-tile: 16 x 8 tiles of 8 x 8 pixels = 128 waves


def tile_shard_counts(width, height, world):
    """(super-tiles of the image, super-tiles per rank): the arithmetic of rt_hip_tile_shard_info"""
    sx = ((width + 7) // 8 + 15) // 16
    sy = ((height + 7) // 8 + 7) // 8
    return sx * sy, (sx * sy + world - 1) // world


def tiles_of_rank(width, height, rank, world):
    """global super-tile indices rendered by `rank`"""
    total, _ = tile_shard_counts(width, height, world)
    return list(range(rank, total, world))


def tiles_to_frame_host(tiles, width, height, world):
    """numpy model of rt_hip_tiles_to_frame_dev: tiles of shape (rank, slot, 64, 128, ...) -> frame (height, width, ...)"""
    import numpy as np
    total, per = tile_shard_counts(width, height, world)
    sx = ((width + 7) // 8 + 15) // 16
    t = np.asarray(tiles)
    assert t.shape[:4] == (world, per, SUPER_H, SUPER_W)
    frame = np.zeros((height, width) + t.shape[4:], dtype=t.dtype)
    for st in range(total):
        x0, y0 = (st % sx) * SUPER_W, (st // sx) * SUPER_H
        w, h = min(SUPER_W, width - x0), min(SUPER_H, height - y0)
        if w > 0 and h > 0:
            frame[y0:y0 + h, x0:x0 + w] = t[st % world, st // world, :h, :w]
    return frame


def render_whitted_tiles(rta, ctx, scene, camera, rank, world, background=(0, 0, 0), bgr8=False, out=None, ids=None, stream=None, raw_scene=None):
    """this rank's interleaved tile shard of Render_RayTracer: (per_rank * 8192, 3) f64 R,G,B -- or uint8 B,G,R with bgr8 --
    in the ABI's tile order, zero where a super-tile sticks out of the image; enqueued on `stream`"""
    import torch
    _, per = tile_shard_counts(camera.width, camera.height, world)
    if out is None:
        out = torch.zeros((per * 8192, 3), dtype=torch.uint8 if bgr8 else torch.float64, device="cuda")
    p = rta.Params(background=background, tile_offset=rank, tile_stride=world, frame_format=1 if bgr8 else 0)
    s = _render_stream(stream)          # after the zero fill above (the current stream's), like the row and sample paths
    rta.check(rta.hip.rt_hip_render_whitted_dev(raw_scene if raw_scene is not None else scene.device(ctx), camera.raw, p, C.c_void_p(out.data_ptr()),
                                                C.c_void_p(ids.data_ptr()) if ids is not None else None, C.c_void_p(s.cuda_stream)))
    _join(s)                            # what the current stream does with `out` next (the all-gather) comes after the render
    return out


def gather_tiles(local, group=None):
    """all-gather of the equally sized tile shards: (world * per_rank * 8192, channels), rank-major"""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    full = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(full, local.contiguous(), group=group)
    return full


def tiles_to_frame(rta, camera, world, tiles, frame=None, stream=None, rank_pitch_bytes=0):
    """rt_hip_tiles_to_frame_dev on device tensors: the gathered shards back into the (H * W, channels) raster frame"""
    import torch
    ch = tiles.shape[1] if tiles.dim() > 1 else 1
    if frame is None:
        frame = torch.empty((camera.height * camera.width, ch) if ch > 1 else (camera.height * camera.width,), dtype=tiles.dtype, device=tiles.device)
    s = _render_stream(stream)          # after the all-gather that filled `tiles` on the current stream
    rta.check(rta.hip.rt_hip_tiles_to_frame_dev(camera.raw, world, C.c_void_p(tiles.data_ptr()), rank_pitch_bytes, tiles.element_size() * ch,
                                                C.c_void_p(frame.data_ptr()), C.c_void_p(s.cuda_stream)))
    _join(s)
    return frame


def render_whitted_tile_split(rta, ctx, scene, camera, background=(0, 0, 0), bgr8=False, group=None):
    """Render_RayTracer over the ranks of `group` by interleaved tiles + all-gather; the full raster frame on every rank"""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    local = render_whitted_tiles(rta, ctx, scene, camera, rank, world, background, bgr8)
    return tiles_to_frame(rta, camera, world, gather_tiles(local, group))
