"""The Newton solves as kernels of their own (ray_tracing_amd/csrc/rt_defer.inc: phase 1 walk + global (ray, object) queue, persistent state-machine
solver, fold) against the in-place form (PairQueue inside the walk kernels, RT_HIP_PARAM_DEFER=0): the same library, the same scenes, the two
forms must agree bit for bit -- ids, colours, the anti-aliasing driver's picture, the photon map's hit points.  Both forms are compared with
the oracle by the other GPU tests (whichever is the default); this file pins them to each other, on scenes with a tree (Bezier patch + mesh)
and without one (three surfaces of revolution), with point, parallel and area lights, depth 1 to 8."""
import os

import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu


class _Form:
    def __init__(self, defer):
        self.v = "1" if defer else "0"

    def __enter__(self):
        self.old = os.environ.get("RT_HIP_PARAM_DEFER")
        os.environ["RT_HIP_PARAM_DEFER"] = self.v           # read by the library at every render call

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("RT_HIP_PARAM_DEFER", None)
        else:
            os.environ["RT_HIP_PARAM_DEFER"] = self.old


def _frame(rta, ctx, g, cam, **pk):
    n = cam.width * cam.height
    rgb = np.zeros((n, 3)); ids = np.zeros(n, dtype=np.int32)
    rta.check(rta.hip.rt_hip_render_whitted(g.device(ctx), cam.raw, rta.Params(**pk), rgb.ctypes.data, ids.ctypes.data))
    return rgb, ids


@pytest.mark.parametrize("depth", [1, 2, 4])
def test_patch_scene_with_tree_both_forms_identical(ctx, depth):
    import ray_tracing_amd as rta
    g, _ = scenes.build_paper(oracle=False)
    c = dict(scenes.C4_CAMERA); c["res"] = 64; c["depth"] = depth
    cam = rta.Camera(**c)
    with _Form(False):
        a_rgb, a_ids = _frame(rta, ctx, g, cam, background=(30, 30, 30))
    with _Form(True):
        b_rgb, b_ids = _frame(rta, ctx, g, cam, background=(30, 30, 30))
    patch = int((a_ids == 1).sum())
    assert patch > 2000                                     # the patch fills a good part of the frame
    assert np.array_equal(a_ids, b_ids)
    assert np.array_equal(a_rgb, b_rgb)


@pytest.mark.parametrize("light", ["point", "area5"])
def test_wine_scene_without_tree_both_forms_identical(ctx, light):
    import ray_tracing_amd as rta
    g, _ = scenes.build_wine(cups="layout", light=light, oracle=False)
    c = dict(scenes.C5_CAMERA); c["res"] = 40; c["depth"] = 8
    cam = rta.Camera(**c)
    with _Form(False):
        a_rgb, a_ids = _frame(rta, ctx, g, cam, background=(10, 10, 10), seed=3)
    with _Form(True):
        b_rgb, b_ids = _frame(rta, ctx, g, cam, background=(10, 10, 10), seed=3)
    assert int((a_ids >= 21).sum()) > 1000 or int(np.unique(a_ids).size) > 5      # cups are in view
    assert np.array_equal(a_ids, b_ids)
    assert np.array_equal(a_rgb, b_rgb)


def test_photon_map_hit_points_both_forms_identical(ctx):
    import ray_tracing_amd as rta
    g, _ = scenes.build_wine(cups="layout", light="area5", oracle=False)
    c = dict(scenes.C5_CAMERA); c["res"] = 24; c["depth"] = 8
    cam = rta.Camera(**c)
    out = []
    for defer in (False, True):
        with _Form(defer):
            pm = rta.PhotonMap(ctx, g, cam, photon_num=1500, times=2, background=(10, 10, 10), seed=5)
            for k in range(2):
                pm.run_pass(k)
            out.append((pm.hitpoints(), pm.picture()))
            del pm
    assert out[0][0].shape == out[1][0].shape and out[0][0].shape[0] > 10000
    assert np.array_equal(out[0][0], out[1][0])             # positions, normals, weights, R, N of every hit point
    assert np.array_equal(out[0][1], out[1][1])             # the picture after two passes
