"""Interleaved tile shards (rt_hip_params.tile_offset / tile_stride), the BGR8 frame (Camera.h:69), the communicator
behind the ABI (rt_hip_comm_*) and one frame rendered by TWO processes: every form of the multi-GPU split against the
one-rank result.  Bars: tile split bit-identical, sample split <= 1e-12, photon split exact."""
import ctypes as C
import os
import socket

import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu


def _full_frame(rta, ctx, g, cam, fmt=0, background=(200, 200, 200)):
    import torch
    n = cam.width * cam.height
    rgb = torch.zeros((n, 3), dtype=torch.uint8 if fmt else torch.float64, device="cuda")
    ids = torch.zeros(n, dtype=torch.int32, device="cuda")
    p = rta.Params(background=background, frame_format=fmt)
    rta.check(rta.hip.rt_hip_render_whitted_dev(g.device(ctx), cam.raw, p, C.c_void_p(rgb.data_ptr()), C.c_void_p(ids.data_ptr()), None))
    torch.cuda.synchronize()
    return rgb.cpu().numpy(), ids.cpu().numpy()


@pytest.mark.parametrize("depth,stride", [(1, 3), (1, 8), (3, 2), (3, 5)])
def test_tile_shards_reassemble(ctx, depth, stride):
    """every rank's shard rendered in turn on the one GPU; the reassembled frame equals the one-call frame bit for bit,
    for the one-launch depth-1 frame and for the wavefront levels, as f64 R,G,B and as the Vec3b picture"""
    import torch
    import ray_tracing_amd as rta
    from ray_tracing_amd import parallel
    g, _ = scenes.build_icosahedron(material="metal", light="point_area", extra_sphere=True, oracle=False)
    cam = rta.Camera(**dict(scenes.C1_CAMERA, hor=4.1, ver=2.9, res=100, depth=depth))        # 410 x 290: ragged super-tiles
    total, per = C.c_uint32(), C.c_uint32()
    rta.check(rta.hip.rt_hip_tile_shard_info(cam.raw, stride, C.byref(total), C.byref(per)))
    assert (total.value, per.value) == parallel.tile_shard_counts(cam.width, cam.height, stride) == (4 * 5, (20 + stride - 1) // stride)
    ref, ref_ids = _full_frame(rta, ctx, g, cam)
    ref8, _ = _full_frame(rta, ctx, g, cam, fmt=1)
    assert (ref8[:, ::-1] == ref.astype(np.uint8)).all()            # (uchar)B,(uchar)G,(uchar)R: truncation of the doubles
    rays_total = 0
    for bgr8 in (False, True):
        shards, idsh = [], []
        for r in range(stride):
            ids = torch.full((per.value * 8192,), -9, dtype=torch.int32, device="cuda")
            shards.append(parallel.render_whitted_tiles(rta, ctx, g, cam, r, stride, (200, 200, 200), bgr8=bgr8, ids=ids))
            idsh.append(ids)
            if not bgr8:
                a, b = C.c_uint64(), C.c_uint64()
                rta.check(rta.hip.rt_hip_last_ray_count(g.device(ctx), C.byref(a), C.byref(b)))
                rays_total += a.value
        tiles = torch.cat(shards)
        frame = parallel.tiles_to_frame(rta, cam, stride, tiles).cpu().numpy()
        assert np.array_equal(frame, ref8 if bgr8 else ref)
        assert np.array_equal(parallel.tiles_to_frame(rta, cam, stride, torch.cat(idsh)).cpu().numpy(), ref_ids)
        # the numpy model of the layout (used by the CPU two-rank test) agrees with the device kernel
        host = parallel.tiles_to_frame_host(tiles.cpu().numpy().reshape(stride, per.value, 64, 128, 3), cam.width, cam.height, stride)
        assert np.array_equal(host.reshape(-1, 3), frame)
        # rt_hip_tiles_to_frames_dev: three frames per rank in one gather buffer (the second one is the real frame), one launch
        blk = shards[0].numel() * shards[0].element_size()
        batch = torch.zeros(stride * 3 * blk, dtype=torch.uint8, device="cuda")
        for r in range(stride):
            batch[(r * 3 + 1) * blk:(r * 3 + 2) * blk] = shards[r].view(torch.uint8).reshape(-1)
        elem = 3 if bgr8 else 24
        outs = torch.full((3, cam.width * cam.height * elem + 8), 7, dtype=torch.uint8, device="cuda")
        rta.check(rta.hip.rt_hip_tiles_to_frames_dev(cam.raw, stride, C.c_void_p(batch.data_ptr()), 3 * blk, elem, 3, blk, C.c_void_p(outs.data_ptr()),
                                                     outs.shape[1], None))
        torch.cuda.synchronize()
        got = outs[1, :cam.width * cam.height * elem].cpu().numpy()
        assert np.array_equal(got.view(np.uint8 if bgr8 else np.float64).reshape(-1, 3), frame)
        assert (outs[:, -8:] == 7).all() and (outs[0, :64] == 0).all()            # pitches respected; frame 0 was made of the zero tiles
    _full_frame(rta, ctx, g, cam)
    a, b = C.c_uint64(), C.c_uint64()
    rta.check(rta.hip.rt_hip_last_ray_count(g.device(ctx), C.byref(a), C.byref(b)))
    assert rays_total == a.value                                       # the shards' ray census adds up to the frame's


def test_tile_shard_argument_checks(ctx):
    import ray_tracing_amd as rta
    g, _ = scenes.build_icosahedron(oracle=False)
    cam = rta.Camera(**scenes.C1_CAMERA)
    bad = [rta.Params(tile_offset=2, tile_stride=2), rta.Params(tile_offset=1), rta.Params(tile_stride=2, pixel_first=0, pixel_count=640),
           rta.Params(frame_format=7)]
    for p in bad:
        assert rta.hip.rt_hip_render_whitted_dev(g.device(ctx), cam.raw, p, C.c_void_p(8), None, None) == -1
    assert rta.hip.rt_hip_render_focusing_dev(g.device(ctx), cam.raw, rta.Params(tile_stride=2), C.c_void_p(8), None) == -4


@pytest.mark.parametrize("force_rccl", [False, True])
def test_comm_one_rank(ctx, force_rccl, monkeypatch):
    """rt_hip_comm_* with one rank: the plain short cut, and (RT_HIP_COMM_FORCE_RCCL=1) the same calls through RCCL --
    the run-time binding, ncclGetUniqueId / ncclCommInitRank with the id passed by value, all-gather and all-reduce"""
    import torch
    import ray_tracing_amd as rta
    if force_rccl:
        monkeypatch.setenv("RT_HIP_COMM_FORCE_RCCL", "1")
    else:
        monkeypatch.delenv("RT_HIP_COMM_FORCE_RCCL", raising=False)
    ident = (C.c_ubyte * 128)()
    if force_rccl:
        rta.check(rta.hip.rt_hip_comm_unique_id(ident))
    comm = C.c_void_p()
    rta.check(rta.hip.rt_hip_comm_create(ctx.raw, ident if force_rccl else None, 0, 1, C.byref(comm)))
    r, w = C.c_int(-1), C.c_int(-1)
    rta.check(rta.hip.rt_hip_comm_info(comm, C.byref(r), C.byref(w)))
    assert (r.value, w.value) == (0, 1)
    src = torch.arange(100000, dtype=torch.float64, device="cuda")
    dst = torch.zeros_like(src)
    rta.check(rta.hip.rt_hip_comm_all_gather(comm, C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), src.numel() * 8, None))
    rta.check(rta.hip.rt_hip_comm_all_reduce_sum_f64(comm, C.c_void_p(dst.data_ptr()), dst.numel(), None))
    torch.cuda.synchronize()
    assert torch.equal(src, dst)
    assert rta.hip.rt_hip_comm_create(ctx.raw, None, 0, 2, C.byref(C.c_void_p())) == -1      # several ranks need the id
    rta.check(rta.hip.rt_hip_comm_destroy(comm))


def test_two_processes_render_one_frame(ctx, tmp_path):
    """two processes (both on the one GPU of the box, gloo on host tensors) render ONE frame / sample set / photon pass
    between them; compared with the one-rank result"""
    import torch
    import torch.multiprocessing as mp
    import ray_tracing_amd as rta
    from ray_tracing_amd import parallel
    import mp_workers
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "two_rank.npz")
    mp.spawn(mp_workers.render_worker, args=(2, port, out), nprocs=2, join=True)
    res = np.load(out)
    g, _ = scenes.build_icosahedron(material="metal", light="point_area", extra_sphere=True, oracle=False)
    cam = rta.Camera(**dict(scenes.C1_CAMERA, hor=4.1, ver=2.9, res=100, depth=3))
    ref, ref_ids = _full_frame(rta, ctx, g, cam)
    ref8, _ = _full_frame(rta, ctx, g, cam, fmt=1)
    assert np.array_equal(res["tiles_f64"], ref) and np.array_equal(res["tiles_bgr8"], ref8) and np.array_equal(res["tiles_ids"], ref_ids)
    g2, _ = scenes.build_icosahedron(material="plastic", light="area3", oracle=False)
    cam2 = rta.Camera(**dict(scenes.C1_CAMERA, hor=1.6, ver=1.2, res=50, depth=1))
    cam2.set_seed(5)
    cam2.render_focusing(ctx, g2, 0.1, background=(200, 200, 200), spp=7)
    err = np.abs(res["dof"] - cam2.rgb()) / np.maximum(1.0, np.abs(cam2.rgb()))
    print("sample split, two ranks: max rel err %.3g" % err.max())
    assert err.max() <= 1e-12
    g3, _ = scenes.build_icosahedron(material="plastic", light="area3", extra_sphere=True, oracle=False)
    cam3 = rta.Camera(**dict(scenes.C1_CAMERA, hor=3.2, ver=2.4, res=20, depth=2))
    pm = rta.PhotonMap(ctx, g3, cam3, photon_num=3000, times=2, seed=3)
    for t in range(2):
        pm.run_pass(t)
    torch.cuda.synchronize()
    hp = pm.hitpoints()
    assert np.array_equal(res["ppm_hitpoints"][:, :9], hp[:, :9]) and np.array_equal(res["ppm_hitpoints"][:, 10], hp[:, 10])   # photon counts N: exact
    assert np.allclose(res["ppm_hitpoints"][:, 9], hp[:, 9], rtol=1e-12, atol=0)                                                # radius R
    assert np.abs(res["ppm_picture"].astype(int) - pm.picture().astype(int)).max() <= 1
    info = pm.info()
    assert tuple(res["ppm_census"]) == (info["photon_segments"], info["photon_hits"])


@pytest.mark.parametrize("scene_kind", ["one_sample", "param_one_light", "two_lights"])
def test_statistics_kernels_give_the_same_frame(ctx, scene_kind):
    """collect_stats != 0 runs the counting instantiations of the kernels; for a depth-1 frame that is the general
    k_frame_direct (with its luminosity array) also where the plain call takes the loop-free one-light kernel"""
    import torch
    import ray_tracing_amd as rta
    if scene_kind == "param_one_light":
        g, _ = scenes.build_wine(oracle=False, light="point")
        cam = rta.Camera(**dict(scenes.C5_CAMERA, res=12, depth=1))
    else:
        g, _ = scenes.build_icosahedron(light="point_area" if scene_kind == "one_sample" else "two", oracle=False)
        cam = rta.Camera(**dict(scenes.C1_CAMERA, res=50))
    n = cam.width * cam.height
    outs = []
    for stats in (0, 1, 0):
        rgb = torch.zeros((n, 3), dtype=torch.float64, device="cuda")
        p = rta.Params(background=(200, 200, 200), collect_stats=stats)
        ctx.stats(reset=True)
        rta.check(rta.hip.rt_hip_render_whitted_dev(g.device(ctx), cam.raw, p, C.c_void_p(rgb.data_ptr()), None, None))
        torch.cuda.synchronize()
        st = ctx.stats(reset=True)
        assert (st["rays"] > 0) == bool(stats)
        outs.append(rgb.cpu().numpy())
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


def test_trace_occluded_dev_matches_host_entry(ctx):
    """rt_hip_trace_occluded_dev (device pointers, a stream, optional statistics) against the host-pointer entry point"""
    import torch
    import ray_tracing_amd as rta
    g, _ = scenes.build_icosahedron(extra_sphere=True, oracle=False)
    rng = np.random.default_rng(3)
    n = 5000
    org = rng.uniform(-2, 2, (n, 3)) + np.array([0, -1, 2.2])
    light = np.array([-10.0, 20.0, 20.0])
    rays = np.ascontiguousarray(np.concatenate([org, light - org], axis=1))
    ref = np.ascontiguousarray(np.broadcast_to(light, (n, 3)))
    want = g.occluded(ctx, rays, ref)
    d_rays, d_ref = torch.from_numpy(rays).cuda(), torch.from_numpy(ref).cuda()
    d_occ = torch.zeros(n, dtype=torch.uint8, device="cuda")
    st = torch.cuda.Stream()
    for stats in (0, 1):
        ctx.stats(reset=True)
        rta.check(rta.hip.rt_hip_trace_occluded_dev(g.device(ctx), C.c_void_p(d_rays.data_ptr()), C.c_void_p(d_ref.data_ptr()), n, 1,
                                                    C.c_void_p(d_occ.data_ptr()), stats, C.c_void_p(st.cuda_stream)))
        st.synchronize()
        assert np.array_equal(d_occ.cpu().numpy(), want) and 0 < want.sum() < n
        assert (ctx.stats()["shadow_rays"] == n) == bool(stats)


def test_repeated_frames_on_one_handle_are_identical(ctx, monkeypatch):
    """from its second depth-1 frame on a handle orders the waves by the run times they left last time (k_cost_order): only the
    schedule changes, the frames stay bit-identical, also when the camera moves in between and when the tiling changes"""
    import ray_tracing_amd as rta
    from ray_tracing_amd import parallel
    monkeypatch.setenv("RT_HIP_COST_ORDER", "2")          # these frames are smaller than the size from which the library does it by itself
    g, _ = scenes.build_icosahedron(material="plastic", light="point_area", oracle=False)
    cams = [rta.Camera(**dict(scenes.C1_CAMERA, depth=1)), rta.Camera(**dict(scenes.C1_CAMERA, depth=1, view=(-10.2, 0.3, 1.4))),
            rta.Camera(**dict(scenes.C1_CAMERA, depth=1, res=60))]
    first = []
    for cam in cams:
        cam.render_raytracer(ctx, g, background=(200, 200, 200))
        first.append((cam.rgb().copy(), cam.prim_id().copy()))
    for rep in range(3):
        for cam, (rgb, ids) in zip(cams, first):
            cam.render_raytracer(ctx, g, background=(200, 200, 200))
            assert np.array_equal(cam.rgb(), rgb) and np.array_equal(cam.prim_id(), ids)
        # the same camera twice in a row: the second frame runs on the first one's costs
        cams[0].render_raytracer(ctx, g, background=(200, 200, 200))
        assert np.array_equal(cams[0].rgb(), first[0][0])
    # rt_hip_last_wave_costs: one entry per 8x8 tile of the padded super-tile grid, every wave has been resident for some cycles
    nw = C.c_uint64()
    rta.check(rta.hip.rt_hip_last_wave_costs(g.device(ctx), None, 0, C.byref(nw)))
    assert nw.value == parallel.tile_shard_counts(cams[0].width, cams[0].height, 1)[0] * 128
    cost = np.zeros(nw.value, dtype=np.uint32)
    rta.check(rta.hip.rt_hip_last_wave_costs(g.device(ctx), C.c_void_p(cost.ctypes.data), nw.value, C.byref(nw)))
    assert (cost > 0).all() and cost.max() > 4 * np.median(cost)        # the tiles on the mesh take several times the background tiles' time
    assert rta.hip.rt_hip_last_wave_costs(g.device(ctx), C.c_void_p(cost.ctypes.data), 5, C.byref(nw)) == -1


@pytest.mark.parametrize("stride", [1, 3, 8])
def test_batched_frames_equal_single_frames(ctx, stride):
    """rt_hip_render_whitted_batch_dev: the tile shares (or whole frames, stride 1) of five cameras in one launch against five calls of
    rt_hip_render_whitted_dev, as the Vec3b picture and as doubles + ids, with one light of one sample (the loop-free kernel) and with the
    area light (luminosities through the lum array); pitches larger than a frame are respected"""
    import torch
    import ray_tracing_amd as rta
    for light in ("point", "point_area"):
        g, _ = scenes.build_icosahedron(material="plastic", light=light, extra_sphere=True, oracle=False)
        cams = [rta.Camera(**dict(scenes.C1_CAMERA, hor=4.1, ver=2.9, res=100, depth=1, view=(-10 + 0.3 * k, 0.2 * k, 1.5))) for k in range(5)]
        arr = rta.camera_array(cams)
        total, per = C.c_uint32(), C.c_uint32()
        rta.check(rta.hip.rt_hip_tile_shard_info(cams[0].raw, stride, C.byref(total), C.byref(per)))
        n_out = per.value * 8192 if stride > 1 else cams[0].width * cams[0].height
        for fmt, elem, dt in ((1, 3, torch.uint8), (0, 24, torch.float64)):
            for rank in range(min(stride, 2)):
                kw = dict(tile_offset=rank, tile_stride=stride) if stride > 1 else {}
                p = rta.Params(background=(200, 200, 200), frame_format=fmt, seed=11, **kw)
                pitch = n_out * elem + 64
                batch = torch.full((5, pitch), 9, dtype=torch.uint8, device="cuda")
                bids = torch.full((5, n_out + 16), -7, dtype=torch.int32, device="cuda")
                rta.check(rta.hip.rt_hip_render_whitted_batch_dev(g.device(ctx), arr, 5, p, C.c_void_p(batch.data_ptr()), pitch, C.c_void_p(bids.data_ptr()), n_out + 16, None))
                torch.cuda.synchronize()
                a, b = C.c_uint64(), C.c_uint64()
                rta.check(rta.hip.rt_hip_last_ray_count(g.device(ctx), C.byref(a), C.byref(b)))
                rays = 0
                for k in range(5):
                    one = torch.full((n_out * elem,), 9, dtype=torch.uint8, device="cuda")
                    ids = torch.full((n_out,), -7, dtype=torch.int32, device="cuda")
                    rta.check(rta.hip.rt_hip_render_whitted_dev(g.device(ctx), cams[k].raw, p, C.c_void_p(one.data_ptr()), C.c_void_p(ids.data_ptr()), None))
                    torch.cuda.synchronize()
                    a1, b1 = C.c_uint64(), C.c_uint64()
                    rta.check(rta.hip.rt_hip_last_ray_count(g.device(ctx), C.byref(a1), C.byref(b1)))
                    rays += a1.value
                    assert torch.equal(batch[k, :n_out * elem], one), (light, stride, fmt, rank, k)
                    assert torch.equal(bids[k, :n_out], ids)
                assert (batch[:, n_out * elem:] == 9).all() and (bids[:, n_out:] == -7).all()
                assert a.value == rays
    # refused: deeper cameras, cameras of another size
    deep = rta.Camera(**dict(scenes.C1_CAMERA, hor=4.1, ver=2.9, res=100, depth=2))
    arr2 = rta.camera_array([cams[0], deep])
    p = rta.Params(background=(200, 200, 200))
    assert rta.hip.rt_hip_render_whitted_batch_dev(g.device(ctx), arr2, 2, p, C.c_void_p(8), 1 << 30, None, 0, None) == -4      # RT_HIP_EUNSUPPORTED
    other = rta.Camera(**dict(scenes.C1_CAMERA, hor=4.0, ver=2.9, res=100, depth=1))
    arr3 = rta.camera_array([cams[0], other])
    assert rta.hip.rt_hip_render_whitted_batch_dev(g.device(ctx), arr3, 2, p, C.c_void_p(8), 1 << 30, None, 0, None) == -1      # RT_HIP_EINVAL
