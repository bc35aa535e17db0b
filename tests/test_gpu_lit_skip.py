"""The all-clear test in front of an area light's sample loop (rt_device.hpp: area_light_all_clear -- a wave whose points can be proved
unoccluded from the point and the light's rectangle alone adds up the samples without making a ray) against the plain loop
(RT_HIP_LIT_SKIP=0 when the scene is uploaded) and against the oracle: the pictures must be the same bit for bit between the two forms, and
within north_star's 1e-3 of the oracle's.  Scenes: the C3 layout (floor, wall, solid, 10x10 light); the same under a ceiling that shadows
everything (an infinite plane that DOES occlude: nothing may be skipped); a low light whose rays graze the floor; spheres and a circle next
to the mesh; two lights."""
import ctypes as C
import os

import numpy as np
import pytest

import parity_log as pl
import scenes

pytestmark = pytest.mark.gpu


class _Skip:
    def __init__(self, on):
        self.v = "1" if on else "0"

    def __enter__(self):
        self.old = os.environ.get("RT_HIP_LIT_SKIP")
        os.environ["RT_HIP_LIT_SKIP"] = self.v               # read by the library when a scene is uploaded

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("RT_HIP_LIT_SKIP", None)
        else:
            os.environ["RT_HIP_LIT_SKIP"] = self.old


def _frame(rta, ctx, g, cam, **pk):
    n = cam.width * cam.height
    rgb = np.zeros((n, 3)); ids = np.zeros(n, dtype=np.int32)
    rta.check(rta.hip.rt_hip_render_whitted(g.device(ctx), cam.raw, rta.Params(**pk), rgb.ctypes.data, ids.ctypes.data))
    return rgb, ids


def _custom(light_pos, light_len, sn, ceiling=False, extras=False, oracle=True):
    P = scenes._Pair(0.5, oracle=oracle)
    plastic = P.mat("plastic")
    P.call("add_plane", (0, 0, 0), (0, 0, 1), mat=plastic, colour=(255, 255, 255))
    P.call("add_plane", (10, 0, 0), (-1, 0, 0), mat=plastic, colour=(0, 0, 0))
    if ceiling:
        P.call("add_plane", (0, 0, 6), (0, 0, -1), mat=plastic, colour=(90, 90, 90))
    P.call("read_obj_triangle", scenes.ICOSA_OBJ, (0, -1, 0), 0.0015, mat=plastic, colour=(241, 46, 24))
    if extras:
        P.call("add_sphere", (0.3, 0.6, 0.5), 0.5, mat=plastic, colour=(161, 143, 81))
        P.call("add_circle", (0.5, 1.8, 0.4), (0.2, -0.5, 1), 0.6, mat=plastic, colour=(20, 90, 200))
    P.call("construct_bsp")
    P.call("add_area_light", light_pos, 1.0, light_len, sn)
    return P.g, P.o


def _both_forms_and_oracle(ctx, build, tag, res=48, seed=7, want_dark=None):
    import ray_tracing_amd as rta
    c = dict(scenes.C3_CAMERA); c["res"] = res
    cam = rta.Camera(**c)
    out = []
    proved = []
    o = None
    for on in (False, True):
        with _Skip(on):
            g, oo = build(oracle=not on)                     # the oracle's scene once
            o = oo or o
            out.append(_frame(rta, ctx, g, cam, background=(30, 30, 30), seed=seed))
            pv, tot, sh = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
            rta.check(rta.hip.rt_hip_last_proved_shadow_rays(g.device(ctx), C.byref(pv)))
            rta.check(rta.hip.rt_hip_last_ray_count(g.device(ctx), C.byref(tot), C.byref(sh)))
            proved.append((pv.value, sh.value))
    (a_rgb, a_ids), (b_rgb, b_ids) = out
    assert proved[0][0] == 0 and proved[0][1] == proved[1][1]      # switched off: nothing is proved; the census of shadow rays is the same
    assert np.array_equal(a_ids, b_ids)
    assert np.array_equal(a_rgb, b_rgb)                      # with and without the all-clear test: the same bits
    o_rgb, o_ids = scenes.oracle_render_whitted(o, c, background=(30, 30, 30), seed=seed)
    off = int((o_ids != b_ids).sum())
    same = o_ids == b_ids
    err = np.abs(o_rgb[same] - b_rgb[same]).max(axis=1) / 255.0
    pl.allow(tag + ": id_mismatches", off, b_ids.size, bar=1e-4, cause="t-ties")
    pl.value(tag + ": max_colour_err", float(err.max()), 1e-3)
    if want_dark is not None:
        lum = b_rgb.sum(axis=1)
        assert (np.median(lum) < 3 * 140) == want_dark
    return b_rgb, proved[1][0] / max(1, proved[1][1])


def test_c3_layout_ten_by_ten_light(ctx):
    rgb, frac = _both_forms_and_oracle(ctx, lambda oracle: scenes.build_icosahedron(material="plastic", light="area10", oracle=oracle), "lit-skip C3 layout")
    assert len(np.unique(rgb.sum(axis=1))) > 50              # a penumbra, not two flat tones
    assert frac > 0.5                                        # most of the floor and the wall is proved clear


def test_ceiling_between_the_scene_and_the_light(ctx):
    # the plane z = 6 faces down between every point and the light at z = 20: all shadow rays are occluded by an INFINITE object
    _, frac = _both_forms_and_oracle(ctx, lambda oracle: _custom((-10, 20, 20), 10.0, 6, ceiling=True, oracle=oracle), "lit-skip ceiling", want_dark=True)
    assert frac == 0.0                                       # under the ceiling no point is clear


def test_low_light_grazing_the_floor(ctx):
    _both_forms_and_oracle(ctx, lambda oracle: _custom((-6, 9, 0.4), 3.0, 5, oracle=oracle), "lit-skip grazing light")


def test_light_inside_the_scene_box_height(ctx):
    # the light's rectangle at the solid's own height, next to it: shafts in every direction, spheres and a circle in the tree
    _both_forms_and_oracle(ctx, lambda oracle: _custom((-3, 2, 0.8), 2.0, 4, extras=True, oracle=oracle), "lit-skip light beside the solid")


def test_two_lights(ctx):
    _both_forms_and_oracle(ctx, lambda oracle: scenes.build_icosahedron(material="plastic", light="two", oracle=oracle), "lit-skip two lights")


def test_depth_of_field_and_a_deeper_ray_tree_both_forms_identical(ctx):
    """the per-level kernels (k_shadow) instead of the one-launch depth-1 frame: lens samples, and a metal solid at depth 3"""
    import ray_tracing_amd as rta
    c = dict(scenes.C3_CAMERA); c["res"] = 40
    c3 = dict(c); c3["depth"] = 3
    out = []
    for on in (False, True):
        with _Skip(on):
            g, _ = scenes.build_icosahedron(material="metal", light="area10", oracle=False)
            cam = rta.Camera(**c)
            n = cam.width * cam.height
            dof = np.zeros((n, 3))
            rta.check(rta.hip.rt_hip_render_focusing(g.device(ctx), cam.raw, rta.Params(background=(200, 200, 200), seed=5, spp=6, lens_radius=0.1), dof.ctypes.data))
            deep = _frame(rta, ctx, g, rta.Camera(**c3), background=(200, 200, 200), seed=5)
            out.append((dof, deep[0], deep[1]))
    assert out[0][0].std() > 1.0
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)
