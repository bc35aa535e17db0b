"""Adaptive anti-aliasing drivers (Camera.h:83-169, 201-280) through the C ABI against the oracle.
Tolerances: edge masks identical except where a Laplacian lies within rounding of its threshold (fraction
<= 1e-3, reported); colours within 1e-3 relative on the pixels whose masks agree."""
import numpy as np
import pytest

import parity_log as pl
import scenes

pytestmark = pytest.mark.gpu


def _rel_err(a, b):
    return np.abs(a - b) / np.maximum(1.0, np.abs(b))


MASK_FLIPS = 1             # 2 x + 1 of the count observed on the GPU box (profiles/r03_parity_counts.json: 0 everywhere)


def _compare(cam, rgb_ref, mask_ref, what, min_edges):
    rgb, mask = cam.rgb(), cam.edge_mask()
    agree = mask == mask_ref
    print("%s: %d masked pixels of %d (oracle %d), mask mismatches %d" % (what, int(mask.sum()), mask.size, int(mask_ref.sum()), int((~agree).sum())))
    assert mask_ref.sum() >= min_edges and (mask_ref == 0).sum() >= min_edges      # both branches are exercised
    pl.allow("mask_mismatches: " + what, (~agree).sum(), mask.size, bar_count=MASK_FLIPS, cause="a Laplacian within rounding of its threshold")
    err = _rel_err(rgb, rgb_ref)[agree]
    print("%s: max rel err %.3g" % (what, err.max()))
    pl.value("max_rel_colour_err: " + what, err.max(), 1e-3)


@pytest.mark.parametrize("material,light,depth", [("metal", "point_area", 2), ("plastic", "area3", 1)])
def test_raytracer_antialiasing_parity(ctx, material, light, depth):
    import ray_tracing_amd as rta
    g, o = scenes.build_icosahedron(material=material, light=light, extra_sphere=True)
    cam_args = dict(scenes.C1_CAMERA, hor=3.2, ver=2.4, res=40, depth=depth)
    cam = rta.Camera(**cam_args)
    cam.set_seed(21)
    cam.render_raytracer_antialiasing(ctx, g, background=(200, 200, 200), oversample_times=12)
    rgb_ref, mask_ref = o.render_antialiasing(scenes.oracle_camera(cam_args), oversample_times=12, background=(200, 200, 200), seed=21)
    _compare(cam, rgb_ref, mask_ref, "aa %s/%s" % (material, light), 200)
    # unmasked pixels are the plain Render_RayTracer frame
    plain = rta.Camera(**cam_args)
    plain.set_seed(21)
    plain.render_raytracer(ctx, g, background=(200, 200, 200))
    keep = cam.edge_mask() == 0
    assert (cam.rgb()[keep] == plain.rgb()[keep]).all()


def test_focusing_antialiasing_parity(ctx):
    import ray_tracing_amd as rta
    g, o = scenes.build_icosahedron(material="plastic", light="point_area", extra_sphere=True)
    cam_args = dict(scenes.C1_CAMERA, hor=1.6, ver=1.2, res=40, depth=1)
    cam = rta.Camera(**cam_args)
    cam.set_seed(9)
    cam.render_focusing_antialiasing(ctx, g, 0.1, background=(200, 200, 200), spp=5, oversample_times=6)
    rgb_ref, mask_ref = o.render_antialiasing(scenes.oracle_camera(cam_args), focusing=True, radius=0.1, spp=5, oversample_times=6,
                                              background=(200, 200, 200), seed=9)
    _compare(cam, rgb_ref, mask_ref, "focusing aa", 100)


def test_antialiasing_background_pixels(ctx):
    """primary rays that leave the scene: depth = Dist(inftyPt, viewPt), zero normal, type -1 (documented
    stand-in for the reference's NULL dereference) -- silhouettes against the background are edges"""
    import ray_tracing_amd as rta
    g, o = scenes.build_icosahedron(material="plastic", light="point", planes=False)
    cam_args = dict(scenes.C1_CAMERA, hor=3.2, ver=2.4, res=30, depth=1)
    cam = rta.Camera(**cam_args)
    cam.render_raytracer_antialiasing(ctx, g, background=(10, 20, 30), oversample_times=8)
    rgb_ref, mask_ref = o.render_antialiasing(scenes.oracle_camera(cam_args), oversample_times=8, background=(10, 20, 30))
    _compare(cam, rgb_ref, mask_ref, "aa with background", 50)


def test_antialiasing_parametric_scene(ctx):
    """Bezier patch in front of a wall (C4 geometry): the normal maps need the (u, v) of the primary hits"""
    import ray_tracing_amd as rta
    g, o = scenes.build_paper()
    cam_args = dict(scenes.C4_CAMERA, hor=3.0, ver=1.7, res=24, depth=2)
    cam = rta.Camera(**cam_args)
    cam.set_seed(3)
    cam.render_raytracer_antialiasing(ctx, g, oversample_times=6)
    rgb_ref, mask_ref = o.render_antialiasing(scenes.oracle_camera(cam_args), oversample_times=6, seed=3)
    _compare(cam, rgb_ref, mask_ref, "aa paper", 20)


def test_antialiasing_pixel_shards_reassemble(ctx):
    """row blocks rendered separately (with the halo rows the stencil needs) are bit-identical to the full frame"""
    import ray_tracing_amd as rta
    g, _ = scenes.build_icosahedron(material="metal", light="point_area", extra_sphere=True)
    cam = rta.Camera(**dict(scenes.C1_CAMERA, hor=3.2, ver=2.4, res=40, depth=2))
    n, W = cam.width * cam.height, cam.width
    aa = rta.AaParams(oversample_times=5)
    full = np.zeros((n, 3)); fmask = np.zeros(n, dtype=np.uint8)
    p = rta.Params(background=(200, 200, 200), seed=4)
    rta.check(rta.hip.rt_hip_render_antialiasing(g.device(ctx), cam.raw, p, aa, full.ctypes.data, fmask.ctypes.data))
    edges = rta.C.c_uint64()
    rta.check(rta.hip.rt_hip_last_aa_edges(g.device(ctx), rta.C.byref(edges)))
    assert edges.value == fmask.sum() > 0
    parts, masks = [], []
    cuts = [0, 8 * W, 40 * W + 17, n]        # whole rows and a cut inside a row
    for a, b in zip(cuts[:-1], cuts[1:]):
        q = rta.Params(background=(200, 200, 200), seed=4, pixel_first=a, pixel_count=b - a)
        out = np.zeros((b - a, 3)); m = np.zeros(b - a, dtype=np.uint8)
        rta.check(rta.hip.rt_hip_render_antialiasing(g.device(ctx), cam.raw, q, aa, out.ctypes.data, m.ctypes.data))
        parts.append(out); masks.append(m)
    assert (np.concatenate(masks) == fmask).all()
    assert (np.concatenate(parts) == full).all()
    # sample shards are refused
    bad = rta.Params(sample_first=0, sample_count=1)
    assert rta.hip.rt_hip_render_antialiasing(g.device(ctx), cam.raw, bad, aa, full.ctypes.data, None) == -4


def test_antialiasing_shards_of_whole_super_tile_rows(ctx):
    """a shard of exactly 64 rows that starts at v_lo > 0: with its halo rows the G-buffer spans one more row of 64-row super-tiles than
    the shard itself (the G pass must size its level 0 and its launch from its OWN rows)"""
    import ray_tracing_amd as rta
    g, _ = scenes.build_icosahedron(material="metal", light="point_area", extra_sphere=True)
    cam = rta.Camera(**dict(scenes.C1_CAMERA, hor=3.6, ver=4.8, res=40, depth=2))
    n, W = cam.width * cam.height, cam.width
    assert cam.height == 192
    aa = rta.AaParams(oversample_times=3)
    full = np.zeros((n, 3)); fmask = np.zeros(n, dtype=np.uint8)
    p = rta.Params(background=(200, 200, 200), seed=4)
    rta.check(rta.hip.rt_hip_render_antialiasing(g.device(ctx), cam.raw, p, aa, full.ctypes.data, fmask.ctypes.data))
    assert fmask.sum() > 0
    for a, b in ((64 * W, 128 * W), (128 * W, 192 * W), (0, 64 * W), (64 * W, 192 * W)):
        q = rta.Params(background=(200, 200, 200), seed=4, pixel_first=a, pixel_count=b - a)
        out = np.full((b - a, 3), -1.0); m = np.full(b - a, 7, dtype=np.uint8)
        rta.check(rta.hip.rt_hip_render_antialiasing(g.device(ctx), cam.raw, q, aa, out.ctypes.data, m.ctypes.data))
        assert (m == fmask[a:b]).all(), (a // W, b // W)
        assert (out == full[a:b]).all(), (a // W, b // W)


def test_resample_every_pixel(ctx):
    """a negative threshold is always exceeded: the driver becomes the fixed-spp jittered render (BASELINE config C4)"""
    import ray_tracing_amd as rta
    g, o = scenes.build_paper()
    cam_args = dict(scenes.C4_CAMERA, hor=3.0, ver=1.7, res=16, depth=2)
    cam = rta.Camera(**cam_args)
    n = cam.width * cam.height
    rgb = np.zeros((n, 3)); mask = np.zeros(n, dtype=np.uint8)
    aa = rta.AaParams(oversample_times=8, depth_grad_threshold=-1.0)
    rta.check(rta.hip.rt_hip_render_antialiasing(g.device(ctx), cam.raw, rta.Params(seed=6), aa, rgb.ctypes.data, mask.ctypes.data))
    assert mask.all()
    rgb_ref, mask_ref = o.render_antialiasing(scenes.oracle_camera(cam_args), oversample_times=8, thresholds=(-1.0, 0.8, 0.5, 80.0), seed=6)
    assert mask_ref.all()
    assert _rel_err(rgb, rgb_ref).max() < 1e-3


def test_no_edges_and_argument_errors(ctx):
    """thresholds out of reach: nothing is re-sampled and the frame is the plain render; bad arguments are refused"""
    import ray_tracing_amd as rta
    g, _ = scenes.build_icosahedron(material="plastic", light="point", oracle=False)
    cam = rta.Camera(**dict(scenes.C1_CAMERA, hor=1.6, ver=1.2, res=20, depth=1))
    n = cam.width * cam.height
    rgb = np.zeros((n, 3)); mask = np.ones(n, dtype=np.uint8)
    aa = rta.AaParams(depth_grad_threshold=1e30, norm_grad_threshold=1e30, type_grad_threshold=1e30, colour_grad_threshold=1e30)
    rta.check(rta.hip.rt_hip_render_antialiasing(g.device(ctx), cam.raw, rta.Params(), aa, rgb.ctypes.data, mask.ctypes.data))
    assert not mask.any()
    cam.render_raytracer(ctx, g)
    assert (rgb == cam.rgb()).all()
    e = rta.C.c_uint64(7)
    rta.check(rta.hip.rt_hip_last_aa_edges(g.device(ctx), rta.C.byref(e)))
    assert e.value == 0
    # NULL aa = the reference's constants; NULL frame, a wrong struct size and an absurd sample count are refused
    rta.check(rta.hip.rt_hip_render_antialiasing(g.device(ctx), cam.raw, rta.Params(), None, rgb.ctypes.data, None))
    assert rta.hip.rt_hip_render_antialiasing(g.device(ctx), cam.raw, rta.Params(), aa, None, None) == -1
    bad = rta.AaParams(); bad.struct_size = 8
    assert rta.hip.rt_hip_render_antialiasing(g.device(ctx), cam.raw, rta.Params(), bad, rgb.ctypes.data, None) == -1
    huge = rta.AaParams(oversample_times=1 << 20, focusing=1)
    assert rta.hip.rt_hip_render_antialiasing(g.device(ctx), cam.raw, rta.Params(spp=1 << 10), huge, rgb.ctypes.data, None) == -1


def test_photon_mapper_antialiasing_parity(ctx):
    """CCamera::Render_PhotonMapper_Antialiasing (Camera.h:315-423): same mask as the ray-traced driver; a masked pixel's
    oversampleTimes jittered rays all deposit hit points on it; the passes add into the CV_32SC3 picture and a masked
    pixel's sums are divided by oversampleTimes at the end"""
    import oracle_bind as ob
    import ray_tracing_amd as rta
    g, o = scenes.build_icosahedron(material="plastic", light="point_area", extra_sphere=True)
    cam_args = dict(scenes.C1_CAMERA, hor=3.2, ver=2.4, res=25, depth=3)
    os_ = 5
    gm = rta.PhotonMap(ctx, g, rta.Camera(**cam_args), photon_num=20000, times=2, background=(10, 10, 10), seed=6, aa=rta.AaParams(oversample_times=os_))
    om = ob.OraclePhotonMap(o, scenes.oracle_camera(cam_args), photon_num=20000, times=2, background=(10, 10, 10), seed=6, aa=dict(oversample_times=os_))
    mg, mo = gm.edge_mask(), om.edge_mask()
    assert mo.sum() >= 200 and (mo == 0).sum() >= 200
    pl.allow("mask_mismatches", (mg != mo).sum(), mo.size, bar_count=MASK_FLIPS, cause="a Laplacian within rounding of its threshold")
    gh, oh = gm.hitpoints(), om.hitpoints()
    print("ppm aa: %d masked pixels of %d, hit points %d vs %d" % (int(mo.sum()), mo.size, len(gh), len(oh)))
    assert len(gh) == len(oh) and (gh[:, :2] == oh[:, :2]).all()                 # same (pixel, (ray << 12) | path) keys
    assert (np.bincount(oh[:, 0].astype(np.int64), minlength=mo.size)[mo == 1] > 1).mean() > 0.5      # masked pixels hold several rays' hit points
    assert np.abs(gh[:, 2:9] - oh[:, 2:9]).max() < 1e-6
    d0 = np.abs(gm.picture() - om.picture()).max(axis=1)
    pl.allow("eye pass: pixels off by more than one", (d0 > 1).sum(), d0.size, bar_count=0)      # picture_ori: truncation of colours that agree to ~1e-13
    for t in range(2):
        gm.run_pass(t); om.run_pass(t)
        gh, oh = gm.hitpoints(), om.hitpoints()
        dN = np.abs(gh[:, 10] - oh[:, 10])
        print("ppm aa pass %d: N total %.1f vs %.1f, hit points with different N %d" % (t, gh[:, 10].sum(), oh[:, 10].sum(), int((dN > 1e-9).sum())))
        pl.allow("pass %d: hit points with different N" % t, (dN > 1e-9).sum(), len(dN), bar=1e-3, cause="photons on the rim of a gather disc")
    pg, po = gm.picture(), om.picture()
    d = np.abs(pg - po).max(axis=1)
    print("ppm aa: picture max |diff| %d, pixels differing %d of %d; masked pixels brighter than the eye pass: %d" % (
        d.max(), int((d > 0).sum()), d.size, int(((po.sum(axis=1) > 0) & (mo == 1)).sum())))
    pl.allow("picture: pixels off by more than 2", (d > 2).sum(), d.size, bar_count=0)
    pl.allow("picture: pixels differing", (d > 0).sum(), d.size, bar=1e-3, cause="round(sum / oversampleTimes) and Vec3b(tau) truncation of values within rounding of a boundary")
    info, oc = gm.info(), om.counters()
    assert abs(info["photon_hits"] - oc["photon_hits"]) <= 1e-4 * oc["photon_hits"]
    # the plain driver on the same scene has one ray per pixel: fewer hit points
    plain = rta.PhotonMap(ctx, g, rta.Camera(**cam_args), photon_num=100, times=1, background=(10, 10, 10), seed=6)
    assert plain.info()["hitpoints"] < gm.info()["hitpoints"] and plain.edge_mask().sum() == 0


def test_godray_antialiasing_parity(ctx):
    """CCamera::Render_RayTracer_GodRay_Antialiasing (Camera.h:450-537): mask from the plain RayTracer's maps, every ray of
    the final picture through CGodRay::RayTracingPass"""
    import oracle_bind as ob
    import ray_tracing_amd as rta
    g, o = scenes.build_room()
    gg = rta.GodRay(ctx, g, (155, 155, 50), seed=9)
    og = ob.OracleGodRay(o, (155, 155, 50), seed=9)
    for x in (gg, og):
        x.add_area_light_source_xz(6.0, (-3.0, 0.0), (9.0, 10.0), 0.4, (1.0, -2.0, -1.0))
        x.photon_mapping_pass()
    cam_args = dict(scenes.ROOM_CAMERA, res=16, depth=2)
    rgb, mask = gg.render_antialiasing(rta.Camera(**cam_args), aa=rta.AaParams(oversample_times=6), background=(20, 20, 30))
    rgb_ref, mask_ref = og.render_antialiasing(scenes.oracle_camera(cam_args), oversample_times=6, background=(20, 20, 30))
    agree = mask == mask_ref
    err = _rel_err(rgb, rgb_ref)[agree]
    print("godray aa: %d masked pixels of %d, mask mismatches %d, max rel err %.3g" % (int(mask_ref.sum()), mask.size, int((~agree).sum()), err.max()))
    assert mask_ref.sum() >= 100 and (mask_ref == 0).sum() >= 100
    pl.allow("mask_mismatches", (~agree).sum(), mask.size, bar_count=MASK_FLIPS, cause="a Laplacian within rounding of its threshold")
    pl.value("max_rel_colour_err", err.max(), 1e-3)
    # unmasked pixels are the plain god-ray frame; the light shafts are in it (it differs from the shaft-less render)
    plain, _ = gg.render(rta.Camera(**cam_args), background=(20, 20, 30))
    keep = mask == 0
    assert (rgb[keep] == plain[keep]).all()
    cam = rta.Camera(**cam_args); cam.set_seed(9)
    cam.render_raytracer(ctx, g, background=(20, 20, 30))
    assert np.abs(plain - cam.rgb()).max() > 1.0
