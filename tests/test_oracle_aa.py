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


def test_oracle_ppm_and_godray_antialiasing_drivers():
    """the oracle's restatement of Camera.h:315-423 and 450-537 against what those lines imply: the mask is the one of the
    ray-traced driver (same maps), unmasked pixels equal the plain drivers' pixels, a masked pixel holds the hit points of
    all its jittered rays, and its photon sums are divided by oversampleTimes"""
    import oracle_bind as ob
    _, o = scenes.build_icosahedron(material="plastic", light="point_area", extra_sphere=True, gpu=False)
    cam_args = dict(scenes.C1_CAMERA, hor=3.2, ver=2.4, res=12, depth=2)
    ocam = scenes.oracle_camera(cam_args)
    os_ = 4
    _, mask_rt = o.render_antialiasing(ocam, oversample_times=os_, background=(10, 10, 10), seed=3)
    aa = ob.OraclePhotonMap(o, ocam, photon_num=4000, times=1, background=(10, 10, 10), seed=3, aa=dict(oversample_times=os_))
    plain = ob.OraclePhotonMap(o, ocam, photon_num=4000, times=1, background=(10, 10, 10), seed=3)
    mask = aa.edge_mask()
    assert (mask == mask_rt).all() and 0 < mask.sum() < mask.size
    keep = mask == 0
    assert (aa.picture()[keep] == plain.picture()[keep]).all()                   # picture_ori of unmasked pixels: the same single ray
    ha, hp = aa.hitpoints(), plain.hitpoints()
    per_pixel_a = np.bincount(ha[:, 0].astype(np.int64), minlength=mask.size)
    per_pixel_p = np.bincount(hp[:, 0].astype(np.int64), minlength=mask.size)
    assert (per_pixel_a[keep] == per_pixel_p[keep]).all() and per_pixel_a[mask == 1].sum() > 2 * per_pixel_p[mask == 1].sum()
    assert ((ha[:, 1].astype(np.int64) >> 12) < os_).all()
    aa.run_pass(0); plain.run_pass(0)
    assert (aa.picture()[keep] == plain.picture()[keep]).all()                   # same hit points, same photons, sums not divided
    _, og = scenes.build_room(gpu=False)
    gr = ob.OracleGodRay(og, (155, 155, 50), seed=9)
    gr.add_area_light_source_xz(6.0, (-3.0, 0.0), (9.0, 10.0), 0.8, (1.0, -2.0, -1.0))
    gr.photon_mapping_pass()
    rc = dict(scenes.ROOM_CAMERA, res=8, depth=1)
    rgb_aa, m = gr.render_antialiasing(scenes.oracle_camera(rc), oversample_times=3, background=(20, 20, 30))
    rgb_plain, _ = gr.render(scenes.oracle_camera(rc), background=(20, 20, 30))
    assert 0 < m.sum() < m.size and (rgb_aa[m == 0] == rgb_plain[m == 0]).all() and np.abs(rgb_aa[m == 1] - rgb_plain[m == 1]).max() > 0
