"""The BASELINE configs on their STATED geometry at their stated image size, strips of the full frame against the oracle
(SURVEY 8d): C3 = Layout3_Icosahedron depth of field with the 10x10 area light at 1920x1080; C4 = Layout2_WallAndPaper
with WallLight.obj (11,578 objects) at 1920x1080 depth 2; C5 = Layout5_Wine with its three cups (uRange 0.7 / 0.5 / 0.8),
5x5 area light, depth 8: eye-pass rows at 1920x1080, and a photon-mapping run of the same scene at reduced resolution and
photon count (the oracle's photon tracing is too slow for 10 M photons).  Mismatch counts are printed."""
import ctypes as C

import numpy as np
import pytest

import scenes
from parity_util import rel_err

pytestmark = pytest.mark.gpu


def _rows(rta, ctx, g, cam, rows, fn="whitted", **pk):
    """product colours (and ids) of whole image rows through the ABI's pixel shard"""
    out = {}
    w = cam.width
    for r in rows:
        p = rta.Params(pixel_first=r * w, pixel_count=w, **pk)
        rgb = np.zeros((w, 3)); ids = np.zeros(w, dtype=np.int32)
        if fn == "whitted":
            rta.check(rta.hip.rt_hip_render_whitted(g.device(ctx), cam.raw, p, rgb.ctypes.data, ids.ctypes.data))
        else:
            rta.check(rta.hip.rt_hip_render_focusing(g.device(ctx), cam.raw, p, rgb.ctypes.data))
        out[r] = (rgb, ids)
    return out


def test_c3_depth_of_field_full_size_rows(ctx):
    import ray_tracing_amd as rta
    g, o = scenes.build_icosahedron(material="plastic", light="area10")
    cam = rta.Camera(**scenes.C3_CAMERA)
    assert (cam.width, cam.height) == (1920, 1080)
    ocam = scenes.oracle_camera(scenes.C3_CAMERA)
    spp = 3
    rows = (200, 700, 1000)
    got = _rows(rta, ctx, g, cam, rows, fn="focusing", background=(200, 200, 200), seed=7, spp=spp, lens_radius=0.1)
    worst = 0.0
    for r in rows:
        ref = o.render_focusing(ocam, 0.1, spp, background=(200, 200, 200), seed=7, first=r * cam.width, count=cam.width)
        e = rel_err(np.minimum(255.0, got[r][0]), ref)
        worst = max(worst, e.max())
        # a lens sample whose camera ray ties between two objects may differ: such pixels are counted, not hidden
        assert (e.max(axis=1) > 1e-3).mean() <= 1e-3
    print("C3 1920x1080, 10x10 light, %d spp, rows %s: max rel err %.3g" % (spp, rows, worst))


def test_c4_wall_light_full_size_rows(ctx):
    import ray_tracing_amd as rta
    g, o = scenes.build_paper()
    assert g.num_objects == o.num_objects == 11578                 # plane + patch + 11,576 triangles
    assert g.tree_stats() == o.tree_stats()
    cam = rta.Camera(**scenes.C4_CAMERA)
    assert (cam.width, cam.height) == (1920, 1080)
    ocam = scenes.oracle_camera(scenes.C4_CAMERA)
    rows = (150, 400, 640, 900)
    got = _rows(rta, ctx, g, cam, rows, background=(30, 30, 30))
    n_bad = n_all = n_patch = n_mesh = 0
    worst = 0.0
    for r in rows:
        ref_rgb, ref_ids = o.render_whitted(ocam, background=(30, 30, 30), first=r * cam.width, count=cam.width)
        rgb, ids = got[r]
        same = ids == ref_ids
        n_bad += int((~same).sum()); n_all += len(ids); n_patch += int((ref_ids == 1).sum()); n_mesh += int((ref_ids >= 2).sum())
        worst = max(worst, rel_err(rgb[same], ref_rgb[same]).max())
    print("C4 1920x1080 on WallLight.obj, rows %s: %d / %d id mismatches, %d patch and %d mesh pixels, max rel err %.3g" % (rows, n_bad, n_all, n_patch, n_mesh, worst))
    assert n_patch > 500 and n_mesh > 300
    assert n_bad / n_all <= 1e-4 and worst < 1e-3


def test_c5_wine_layout_eye_rows_and_photon_pass(ctx):
    import ray_tracing_amd as rta
    import oracle_bind as ob
    g, o = scenes.build_wine(cups="layout", light="area5")
    assert g.num_objects == o.num_objects == 24
    cam = rta.Camera(**scenes.C5_CAMERA)
    assert (cam.width, cam.height) == (1920, 1080)
    ocam = scenes.oracle_camera(scenes.C5_CAMERA)
    rows = (420, 650)
    got = _rows(rta, ctx, g, cam, rows, background=(10, 10, 10), seed=2)
    n_bad = n_all = n_off = 0
    for r in rows:
        ref_rgb, ref_ids = o.render_whitted(ocam, background=(10, 10, 10), seed=2, first=r * cam.width, count=cam.width)
        rgb, ids = got[r]
        same = ids == ref_ids
        e = rel_err(rgb[same], ref_rgb[same]).max(axis=1)
        n_bad += int((~same).sum()); n_all += len(ids); n_off += int((e > 1e-3).sum())
        assert (ref_ids >= 3).sum() > 100                          # the row crosses the cups
    # pixels whose ray tree meets two Newton roots 2 pi k apart (Object.h:864, 870 commented out) may take the other branch
    print("C5 eye pass 1920x1080 depth 8, rows %s: %d / %d primary id mismatches, %d pixels off by > 1e-3" % (rows, n_bad, n_all, n_off))
    assert n_bad / n_all <= 1e-4 and n_off / n_all <= 2e-3
    # photon mapping on the same scene (three cups, 5x5 light): reduced frame and photon count, two passes
    small = dict(scenes.C5_CAMERA, res=24)                         # 192 x 108
    pm = rta.PhotonMap(ctx, g, rta.Camera(**small), photon_num=1500, times=2, background=(10, 10, 10), seed=4)
    om = ob.OraclePhotonMap(o, scenes.oracle_camera(small), photon_num=1500, times=2, background=(10, 10, 10), seed=4)
    for t in range(2):
        pm.run_pass(t); om.run_pass(t)
    hp, hr = pm.hitpoints(), om.hitpoints()
    assert pm.info()["photons_per_pass"] == 1500 * 25
    # depth 8 through three glass cups: a ray tree whose Newton iteration lands on the other of two roots 2 pi k apart
    # (Object.h:864, 870 commented out) or whose weight sits on the 0.1 cut may branch differently; such hit points are
    # counted, the rest must agree
    kp = hp[:, 0].astype(np.int64) * 65536 + hp[:, 1].astype(np.int64)
    kr = hr[:, 0].astype(np.int64) * 65536 + hr[:, 1].astype(np.int64)
    common, ip, ir = np.intersect1d(kp, kr, return_indices=True)
    only = len(kp) + len(kr) - 2 * len(common)
    n_same = int((np.abs(hp[ip, 10] - hr[ir, 10]) <= 1e-9).sum())
    pos = np.abs(hp[ip, 2:5] - hr[ir, 2:5]).max(axis=1)
    dpic = np.abs(pm.picture() - om.picture()).max(axis=1)
    print("C5 photon map 192x108 depth 8, 37,500 photons x 2 passes: hit points %d vs %d (%d not shared), N equal on %d of %d shared, "
          "positions off by > 1e-6 on %d, picture pixels off by > 1: %d of %d" % (len(hp), len(hr), only, n_same, len(common), int((pos > 1e-6).sum()), int((dpic > 1).sum()), len(dpic)))
    assert len(common) > 15000 and only / len(common) <= 2e-3
    assert n_same / len(common) >= 1 - 2e-3 and (pos > 1e-6).mean() <= 2e-3
    assert (dpic > 2).mean() <= 2e-3
