"""Textures and bump textures (Object.h:79-92, 186-196 and the per-type UV code) and RotationalSurfaceToMesh
(Scene.h:168-266) through the C ABI against the oracle.  Texel indices come from atan2 / asin / sin / cos and a
truncation to int: where the device's libm and the host's differ in the last bit a hit on a texel boundary can read the
neighbouring texel, so a fraction <= 5e-4 of the pixels (reported) may differ; all others agree within 1e-3 relative."""
import numpy as np
import pytest

import parity_log as pl
import scenes
from test_gpu_parity import _ppm_compare

pytestmark = pytest.mark.gpu


def _rel_err(a, b):
    return np.abs(a - b) / np.maximum(1.0, np.abs(b))


TEXEL_PIXELS = 11          # 2 x + 1 of the count observed on the GPU box (profiles/r03_parity_counts.json: 5 of 110,592)


def _check_image(rgb, rgb_ref, what):
    err = _rel_err(rgb, rgb_ref).max(axis=1)
    bad = err >= 1e-3
    print("%s: max rel err on agreeing pixels %.3g, texel-boundary pixels %d of %d" % (what, err[~bad].max(), int(bad.sum()), bad.size))
    pl.allow("texel_boundary_pixels: " + what, bad.sum(), bad.size, bar_count=TEXEL_PIXELS, cause="a hit within an ulp of a texel boundary (atan2 / asin of the device library vs glibc) reads the neighbouring texel")


@pytest.mark.parametrize("bsp,light,depth", [(True, "point_area", 3), (False, "area", 2)])
def test_textured_scene_image_parity(ctx, bsp, light, depth):
    import ray_tracing_amd as rta
    g, o = scenes.build_textured(bsp=bsp, light=light)
    cam_args = dict(scenes.TEX_CAMERA, depth=depth)
    cam = rta.Camera(**cam_args)
    cam.set_seed(7)
    cam.render_raytracer(ctx, g, background=(20, 30, 40))
    rgb_ref, ids_ref = scenes.oracle_render_whitted(o, cam_args, background=(20, 30, 40), seed=7)
    assert (cam.prim_id() == ids_ref).all()
    assert len(np.unique(ids_ref)) > 60          # planes, spheres, patch, cup, both meshes, triangle, circle are all in view
    _check_image(cam.rgb(), rgb_ref, "textured scene (bsp=%s, %s, depth %d)" % (bsp, light, depth))
    bad = _rel_err(cam.rgb(), rgb_ref).max(axis=1) >= 1e-3
    print("  objects under the texel-boundary pixels:", dict(zip(*np.unique(ids_ref[bad], return_counts=True))))
    # textures change the picture: the same scene without them differs on most pixels
    assert (np.abs(rgb_ref - rgb_ref.mean(axis=0)).max(axis=1) > 1).mean() > 0.5


TEX_AA_MASK, TEX_AA_PIXELS = 3, 9           # 2 x + 1 of the counts observed on the GPU box (profiles/r03_parity_counts.json: 0 and 4 of 27,648)


def test_textured_scene_antialiasing(ctx):
    """the colour maps of the anti-aliasing driver see the texture edges (Camera.h:147-149)"""
    import ray_tracing_amd as rta
    g, o = scenes.build_textured()
    cam_args = dict(scenes.TEX_CAMERA, res=20, depth=2)
    cam = rta.Camera(**cam_args)
    cam.set_seed(2)
    cam.render_raytracer_antialiasing(ctx, g, background=(20, 30, 40), oversample_times=6)
    rgb_ref, mask_ref = o.render_antialiasing(scenes.oracle_camera(cam_args), oversample_times=6, background=(20, 30, 40), seed=2)
    mask = cam.edge_mask()
    agree = mask == mask_ref
    print("textured aa: %d masked pixels of %d, mask mismatches %d" % (int(mask_ref.sum()), mask.size, int((~agree).sum())))
    assert mask_ref.mean() > 0.2
    pl.allow("mask_mismatches", (~agree).sum(), mask.size, bar_count=TEX_AA_MASK, cause="texel-boundary pixels change the colour maps' Laplacian")
    err = _rel_err(cam.rgb(), rgb_ref).max(axis=1)[agree]
    # a jittered sample on a texel boundary moves a masked pixel's mean by 1 / oversample_times of a texel difference
    pl.allow("pixels_off_by_more_than_1e-3", (err >= 1e-3).sum(), err.size, bar_count=TEX_AA_PIXELS, cause="jittered samples on texel boundaries")


def test_textured_photon_map(ctx):
    """photon colours pick up the texture at every bounce (PhotonMapper.h:296, 312, 324); hit-point normals are
    the bump-mapped ones (PhotonMapper.h:178)"""
    g, o = scenes.build_textured(bsp=True, light="area")
    cam_args = dict(scenes.TEX_CAMERA, res=12, depth=3)
    _ppm_compare(ctx, g, o, cam_args, 3000, 2, "textured ppm", 3000, texel_flips=5 / 9890.0, picture_flips=49)
    # (observed on the GPU box: 2 of 9,890 eye-pass pixels on texel boundaries; 24 picture pixels whose photons picked up a neighbouring texel)


def test_texture_argument_errors(ctx):
    import ray_tracing_amd as rta
    g = rta.Scene(0.1)
    pl = g.add_plane((0, 0, 0), (0, 0, 1), rta.material("plastic"), (255, 255, 255))
    t = g.add_texture_image(scenes.synth_texture(8, 8, 1))
    with pytest.raises(Exception):
        g.add_texture(pl, t + 1)
    with pytest.raises(Exception):
        g.add_texture(pl + 5, t)
    g.add_bump_texture(pl, t, 0.0)                      # bumpTextureScale 0 divides by zero in the reference (Object.h:756)
    g.add_point_light((0, 0, 5), 1.0)
    with pytest.raises(Exception):
        g.device(ctx)
