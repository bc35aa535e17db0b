"""Worker of tests/test_gpu_tiles.py::test_two_processes_render_one_frame: one process per rank, every rank on GPU 0
(the GPU box has one GPU), collectives over gloo on HOST tensors (gloo on device tensors from a side stream with two ranks
on one device is the combination that hung in round 1, DESIGN.md section 7: it is not used anywhere).  Not collected by pytest."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.dirname(HERE), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def render_worker(rank, world, port, out_path):
    import torch
    import torch.distributed as dist
    import ray_tracing_amd as rta
    from ray_tracing_amd import parallel
    import scenes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    ctx = rta.Context(0)
    res = {}
    # ---- tile split (deterministic Whitted, depth 3): all-gather of the interleaved shards, bit-identical ----
    g, _ = scenes.build_icosahedron(material="metal", light="point_area", extra_sphere=True, oracle=False)
    cam = rta.Camera(**dict(scenes.C1_CAMERA, hor=4.1, ver=2.9, res=100, depth=3))          # 410 x 290: 4 x 5 super-tiles, ragged
    for bgr8 in (False, True):
        ids = torch.full((parallel.tile_shard_counts(cam.width, cam.height, world)[1] * 8192,), -9, dtype=torch.int32, device="cuda")
        local = parallel.render_whitted_tiles(rta, ctx, g, cam, rank, world, (200, 200, 200), bgr8=bgr8, ids=ids)
        torch.cuda.synchronize()
        full = parallel.gather_tiles(local.cpu())                         # gloo, host tensors
        frame = parallel.tiles_to_frame(rta, cam, world, full.cuda())
        idf = parallel.tiles_to_frame(rta, cam, world, parallel.gather_tiles(ids.cpu()).cuda())
        torch.cuda.synchronize()
        res["tiles_bgr8" if bgr8 else "tiles_f64"] = frame.cpu().numpy()
        res["tiles_ids"] = idf.cpu().numpy()
    # ---- sample split (Render_Focusing): all-reduce of the un-clamped sums ----
    g2, _ = scenes.build_icosahedron(material="plastic", light="area3", oracle=False)
    cam2 = rta.Camera(**dict(scenes.C1_CAMERA, hor=1.6, ver=1.2, res=50, depth=1))
    spp = 7
    s0, n = parallel.sample_shard(spp, rank, world)
    acc = torch.zeros((cam2.height * cam2.width, 3), dtype=torch.float64, device="cuda")
    p = rta.Params(background=(200, 200, 200), seed=5, spp=spp, lens_radius=0.1, sample_first=s0, sample_count=n)
    rta.check(rta.hip.rt_hip_render_focusing_dev(g2.device(ctx), cam2.raw, p, C.c_void_p(acc.data_ptr()), None))
    torch.cuda.synchronize()
    res["dof"] = parallel.reduce_samples(acc.cpu()).numpy()
    # ---- photon split (photon mapping): every rank traces its photon range, the accumulators are all-reduced ----
    g3, _ = scenes.build_icosahedron(material="plastic", light="area3", extra_sphere=True, oracle=False)
    cam3 = rta.Camera(**dict(scenes.C1_CAMERA, hor=3.2, ver=2.4, res=20, depth=2))
    pm = rta.PhotonMap(ctx, g3, cam3, photon_num=3000, times=2, seed=3)
    for t in range(2):
        first, count = parallel.photon_shard(pm.info()["photons_per_pass"], rank, world)
        pm.trace(t, first, count, None)
        ptr, nd = pm.accumulators()
        acc_dev = torch.as_tensor(parallel._DeviceArray(ptr, nd), device="cuda")
        torch.cuda.synchronize()
        host = acc_dev.cpu()
        dist.all_reduce(host)
        acc_dev.copy_(host)
        torch.cuda.synchronize()
        pm.update(None)
    torch.cuda.synchronize()
    res["ppm_hitpoints"] = pm.hitpoints()
    res["ppm_picture"] = pm.picture()
    info = pm.info()
    cnt = torch.tensor([info["photon_segments"], info["photon_hits"]], dtype=torch.int64)
    dist.all_reduce(cnt)
    res["ppm_census"] = cnt.numpy()
    if rank == 0:
        np.savez(out_path, **res)
    dist.barrier()
    dist.destroy_process_group()
