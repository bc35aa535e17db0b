"""rt_hip_render_whitted_noalloc (include/rt_hip.h): the allocation-free variant of rt_hip_render_whitted_dev.
Called through the C ABI and compared bit for bit with rt_hip_render_whitted_dev (same kernels), on two calls in a
row, the second on a smaller frame (the fixed per-GPU scratch array keeps the first call's contents).  Both kernel
forms are covered: one light of one sample (no luminosity array) and an area light of 2x2 samples + a parallel
light (luminosities in the scratch array).  Sizes beyond the array are refused with a clear error."""
import ctypes as C

import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu


def _render(rta, ctx, scene, cam_args, fn, seed=3):
    import torch
    cam = rta.Camera(**cam_args)
    n = cam.width * cam.height
    rgb = torch.full((n * 3,), -1.0, dtype=torch.float64, device="cuda")
    ids = torch.full((n,), -7, dtype=torch.int32, device="cuda")
    p = rta.Params(background=(200, 200, 200), seed=seed)
    st = torch.cuda.current_stream()
    rc = fn(scene.device(ctx), cam.raw, p, C.c_void_p(rgb.data_ptr()), C.c_void_p(ids.data_ptr()), C.c_void_p(st.cuda_stream))
    torch.cuda.synchronize()
    rays, shadow = C.c_uint64(0), C.c_uint64(0)
    if rc == 0:
        rta.check(rta.hip.rt_hip_last_ray_count(scene.device(ctx), C.byref(rays), C.byref(shadow)))
    return rc, rgb.cpu().numpy(), ids.cpu().numpy(), (rays.value, shadow.value)


@pytest.mark.parametrize("light", ["point_area", "two"])
def test_noalloc_matches_dev_twice(ctx, light):
    import ray_tracing_amd as rta
    g, _ = scenes.build_icosahedron(light=light, gpu=True, oracle=False)
    big = dict(scenes.C1_CAMERA)                                   # 640 x 480
    small = dict(scenes.C1_CAMERA, hor=3.3, ver=2.1, res=70)       # 231 x 147: ragged 8x8 tiles, fewer waves than the call before
    for cam_args in (big, small):
        rc0, rgb0, ids0, cnt0 = _render(rta, ctx, g, cam_args, rta.hip.rt_hip_render_whitted_dev)
        rc1, rgb1, ids1, cnt1 = _render(rta, ctx, g, cam_args, rta.hip.rt_hip_render_whitted_noalloc)
        assert rc0 == 0 and rc1 == 0, rta.hip.rt_hip_last_error()
        assert (ids0 != -7).all() and (ids1 == ids0).all()
        assert np.array_equal(rgb0, rgb1)                          # same kernels, same arithmetic: bit-identical
        assert cnt0 == cnt1 and cnt1[0] > 0


def test_noalloc_refuses_what_does_not_fit(ctx):
    import ray_tracing_amd as rta
    g, _ = scenes.build_icosahedron(light="two", gpu=True, oracle=False)
    # 2 lights x 8 B per padded pixel: 4000 x 3000 needs ~190 MB of luminosities, the array holds 32 MiB
    huge = dict(scenes.C1_CAMERA, hor=40.0, ver=30.0)
    cam = rta.Camera(**huge)
    p = rta.Params(background=(0, 0, 0))
    rc = rta.hip.rt_hip_render_whitted_noalloc(g.device(ctx), cam.raw, p, C.c_void_p(8), None, None)   # refused before the pointer is used
    assert rc == -1 and b"bytes of scratch" in rta.hip.rt_hip_last_error()
    deep = rta.Camera(**dict(scenes.C1_CAMERA, depth=3))
    rc = rta.hip.rt_hip_render_whitted_noalloc(g.device(ctx), deep.raw, p, C.c_void_p(8), None, None)
    assert rc == -4 and b"depth" in rta.hip.rt_hip_last_error()


def test_noalloc_call_replays_from_a_hip_graph(ctx):
    """what the entry point is for: recorded into a HIP graph (stream capture forbids hipMalloc) and replayed, bit-identical to the direct call"""
    import torch
    import ray_tracing_amd as rta
    g, _ = scenes.build_icosahedron(light="point_area", gpu=True, oracle=False)
    cam = rta.Camera(**scenes.C1_CAMERA)
    n = cam.width * cam.height
    p = rta.Params(background=(200, 200, 200), seed=3)
    ref = torch.zeros(n * 3, dtype=torch.float64, device="cuda"); ref_ids = torch.zeros(n, dtype=torch.int32, device="cuda")
    rta.check(rta.hip.rt_hip_render_whitted_dev(g.device(ctx), cam.raw, p, C.c_void_p(ref.data_ptr()), C.c_void_p(ref_ids.data_ptr()), None))
    torch.cuda.synchronize()
    rgb = torch.full((n * 3,), -1.0, dtype=torch.float64, device="cuda"); ids = torch.full((n,), -7, dtype=torch.int32, device="cuda")
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.graph(graph, stream=side):
        rc = rta.hip.rt_hip_render_whitted_noalloc(g.device(ctx), cam.raw, p, C.c_void_p(rgb.data_ptr()), C.c_void_p(ids.data_ptr()),
                                                   C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rta.hip.rt_hip_last_error()
    assert (ids == -7).all()                       # recording runs nothing
    for rep in range(3):
        rgb.fill_(-1.0); ids.fill_(-7)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(ids, ref_ids) and torch.equal(rgb, ref)
