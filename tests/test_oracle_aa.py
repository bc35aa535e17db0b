"""CPU checks of the oracle's anti-aliasing driver (Camera.h:83-169): the Laplacian with cv::filter2D's default
border is compared with an independent implementation (scipy.ndimage, mode 'mirror' = BORDER_REFLECT_101), and
the driver's two branches are checked against plain Render_RayTracer.  OpenCV itself is not in the image: the
border rule is taken from its documentation (parity unpinned by any reference output)."""
import numpy as np
from scipy import ndimage

import scenes


def _depth_map(o, cam_args):
    rays, w, h = o.camera_rays(scenes.oracle_camera(cam_args))
    ids, pts = o.trace_closest(rays, use_bsp=False, record=True)
    return ids, np.linalg.norm(pts - rays[:, :3], axis=1).reshape(h, w), w, h


def test_depth_mask_matches_scipy_laplace():
    _, o = scenes.build_icosahedron(material="plastic", light="point", gpu=False, extra_sphere=True)
    cam_args = dict(scenes.C1_CAMERA, hor=3.2, ver=2.4, res=20, depth=1)
    ids, depth, w, h = _depth_map(o, cam_args)
    assert (ids >= 0).all()
    # only the depth criterion can fire: the other thresholds are out of reach
    _, mask = o.render_antialiasing(scenes.oracle_camera(cam_args), oversample_times=1, thresholds=(0.8, 1e30, 1e30, 1e30))
    lap = ndimage.laplace(depth, mode="mirror")
    exp = np.abs(lap) > 0.8
    near = np.abs(np.abs(lap) - 0.8) < 1e-9          # summation order may differ within rounding
    assert ((mask.reshape(h, w) == exp) | near).all()
    assert exp.sum() > 50 and exp[0].any() | exp[-1].any() | exp[:, 0].any() | exp[:, -1].any()   # the border rule is exercised


def test_unmasked_pixels_are_plain_raytracer_and_masked_are_means():
    _, o = scenes.build_icosahedron(material="metal", light="point_area", gpu=False, extra_sphere=True)
    cam_args = dict(scenes.C1_CAMERA, hor=1.6, ver=1.2, res=30, depth=2)
    rgb, mask = o.render_antialiasing(scenes.oracle_camera(cam_args), oversample_times=4, background=(200, 200, 200), seed=2)
    plain, _ = scenes.oracle_render_whitted(o, cam_args, background=(200, 200, 200), seed=2)
    assert 0 < mask.sum() < mask.size
    assert (rgb[mask == 0] == plain[mask == 0]).all()
    assert (rgb[mask == 1] != plain[mask == 1]).any()
    assert rgb.min() >= 0 and rgb.max() <= 255
    # one sample, zero jitter radius: the masked pixels fall back to the plain colour too
    rgb1, mask1 = o.render_antialiasing(scenes.oracle_camera(cam_args), oversample_times=1, r0=0.0, background=(200, 200, 200), seed=2)
    assert (mask1 == mask).all()
    assert np.allclose(rgb1, plain, rtol=0, atol=1e-9)


def test_focusing_variant_ignores_colour_maps():
    _, o = scenes.build_icosahedron(material="plastic", light="point_area", gpu=False)
    cam_args = dict(scenes.C1_CAMERA, hor=1.6, ver=1.2, res=20, depth=1)
    oc = scenes.oracle_camera(cam_args)
    _, m_rt = o.render_antialiasing(oc, oversample_times=1, thresholds=(1e30, 1e30, 1e30, 1.0))
    _, m_f = o.render_antialiasing(oc, focusing=True, radius=0.05, spp=1, oversample_times=1, thresholds=(1e30, 1e30, 1e30, 1.0))
    assert m_rt.sum() > 0 and m_f.sum() == 0          # Camera.h:258-262 has no colour criterion
