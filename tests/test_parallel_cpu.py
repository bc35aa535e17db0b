"""N>1 path on CPU: two gloo ranks, the shard arithmetic and the collectives of ray_tracing_amd/parallel.py.
The per-pixel 'renderer' is a stand-in function of (pixel, sample): what is tested is that row shards
all-gather and sample shards all-reduce into exactly the single-rank frame."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _fake_pixel(p, s):
    # deterministic, non-negative, sometimes above the clamp when summed
    return np.stack([(p * 7 + s * 13) % 300, (p * 3 + s) % 41, (p + 5 * s) % 97], axis=-1).astype(np.float64)


def _worker(rank, world, port, H, W, spp, out):
    import importlib.util
    spec = importlib.util.spec_from_file_location("rt_parallel", os.path.join(ROOT, "ray_tracing_amd", "parallel.py"))
    par = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(par)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = par.row_shard(H, W, rank, world)
    local = torch.from_numpy(_fake_pixel(np.arange(first, first + count), 0))
    full = par.gather_rows(local, H, W, 3)
    s0, n = par.sample_shard(spp, rank, world)
    acc = torch.zeros((H * W, 3), dtype=torch.float64)
    for s in range(s0, s0 + n):
        acc += torch.from_numpy(_fake_pixel(np.arange(H * W), s)) / spp
    red = par.reduce_samples(acc)
    # interleaved tile shards: this rank's super-tiles in the ABI's layout (slot-major, 64 x 128 entries, zero padding)
    total, per = par.tile_shard_counts(W, H, world)
    sx = ((W + 7) // 8 + 15) // 16
    shard = np.zeros((per, 64, 128, 3))
    for slot, st in enumerate(par.tiles_of_rank(W, H, rank, world)):
        x0, y0 = (st % sx) * 128, (st // sx) * 64
        w, h = min(128, W - x0), min(64, H - y0)
        yy, xx = np.meshgrid(np.arange(y0, y0 + h), np.arange(x0, x0 + w), indexing="ij")
        shard[slot, :h, :w] = _fake_pixel(yy * W + xx, 1)
    tiles = par.gather_tiles(torch.from_numpy(shard.reshape(-1, 3)))
    frame = par.tiles_to_frame_host(tiles.numpy().reshape(world, per, 64, 128, 3), W, H, world)
    if rank == 0:
        np.save(out, np.stack([full.numpy(), red.numpy(), frame.reshape(-1, 3)]))
    dist.barrier()
    dist.destroy_process_group()


def test_row_and_sample_shards_partition():
    import importlib.util
    spec = importlib.util.spec_from_file_location("rt_parallel", os.path.join(ROOT, "ray_tracing_amd", "parallel.py"))
    par = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(par)
    for H, W, world in ((1080, 1920, 8), (7, 5, 4), (480, 640, 3), (1, 1, 2)):
        pos = 0
        for r in range(world):
            first, count = par.row_shard(H, W, r, world)
            assert first == pos and first % W == 0
            pos += count
        assert pos == H * W
    for H, W, world in ((1080, 1920, 8), (290, 409, 3), (64, 128, 2), (1, 1, 4)):
        total, per = par.tile_shard_counts(W, H, world)
        mine = [par.tiles_of_rank(W, H, r, world) for r in range(world)]
        assert sorted(sum(mine, [])) == list(range(total)) and max(len(m) for m in mine) <= per
        assert max(len(m) for m in mine) - min(len(m) for m in mine) <= 1
    assert par.tile_shard_counts(1920, 1080, 8) == (255, 32)
    for spp, world in ((1024, 8), (100, 3), (1, 4)):
        pos = 0
        for r in range(world):
            s0, n = par.sample_shard(spp, r, world)
            assert s0 == pos
            pos += n
        assert pos == spp


def test_two_rank_gather_and_reduce(tmp_path):
    H, W, spp, world = 137, 291, 5, 2            # 3 x 3 super-tiles, ragged on both sides
    out = str(tmp_path / "res.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, H, W, spp, out), nprocs=world, join=True)
    full, red, frame = np.load(out)
    assert (full == _fake_pixel(np.arange(H * W), 0)).all()            # row split: bit-identical
    assert (frame == _fake_pixel(np.arange(H * W), 1)).all()           # interleaved tile split: bit-identical
    ref = np.zeros((H * W, 3))
    for s in range(spp):                                              # the reference's running clamped sum
        ref = np.minimum(255.0, ref + _fake_pixel(np.arange(H * W), s) / spp)
    assert np.allclose(red, ref, rtol=1e-12, atol=1e-12)              # sample split: FP reassociation only
