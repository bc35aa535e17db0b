"""The collective paths of ray_tracing_amd/parallel.py on the real backend (RCCL through torch.distributed's "nccl"),
one rank: the device-pointer wrapping, the in-place all-reduce of the library's accumulator buffer and the all-gather of
row blocks run as they do on N GPUs; the N > 1 shard arithmetic is covered on CPU (tests/test_parallel_cpu.py) and by the
in-process shard tests of test_gpu_parity.py."""
import os
import socket

import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rccl_group():
    import torch
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    yield dist
    dist.destroy_process_group()


def test_sharded_photon_pass_equals_plain_pass(ctx, rccl_group):
    import torch
    import ray_tracing_amd as rta
    from ray_tracing_amd import parallel
    g, _ = scenes.build_icosahedron(material="plastic", light="area3", extra_sphere=True, oracle=False)
    cam = rta.Camera(**dict(scenes.C1_CAMERA, hor=3.2, ver=2.4, res=20, depth=2))
    a = rta.PhotonMap(ctx, g, cam, photon_num=3000, times=2, seed=3)
    b = rta.PhotonMap(ctx, g, cam, photon_num=3000, times=2, seed=3)
    for t in range(2):
        a.run_pass(t)
        parallel.ppm_pass_sharded(b, t)
    torch.cuda.synchronize()
    assert (a.hitpoints() == b.hitpoints()).all()
    assert (a.picture() == b.picture()).all()
    assert parallel.photon_shard(10, 0, 3) == (0, 4) and parallel.photon_shard(10, 2, 3) == (7, 3)


def test_row_sharded_renders_on_rccl(ctx, rccl_group):
    import ray_tracing_amd as rta
    from ray_tracing_amd import parallel
    g, _ = scenes.build_icosahedron(material="metal", light="point_area", extra_sphere=True, oracle=False)
    cam = rta.Camera(**dict(scenes.C1_CAMERA, hor=3.2, ver=2.4, res=30, depth=2))
    full = parallel.render_whitted_rows(rta, ctx, g, cam, background=(200, 200, 200)).cpu().numpy()
    cam.render_raytracer(ctx, g, background=(200, 200, 200))
    assert (full == cam.rgb()).all()
    aa = parallel.render_antialiasing_rows(rta, ctx, g, cam, aa=rta.AaParams(oversample_times=4), background=(200, 200, 200), seed=1).cpu().numpy()
    cam.set_seed(1)
    cam.render_raytracer_antialiasing(ctx, g, background=(200, 200, 200), oversample_times=4)
    assert (aa == cam.rgb()).all()
    dof = parallel.render_focusing_samples(rta, ctx, g, cam, 0.1, 6, background=(200, 200, 200), seed=1).cpu().numpy()
    cam.render_focusing(ctx, g, 0.1, background=(200, 200, 200), spp=6)
    assert np.allclose(dof, cam.rgb(), rtol=1e-12, atol=1e-9)
