"""HIP path against the oracle on bounded samples of BASELINE's configs C3 / C4 / C5 at their stated geometry and image size.  One place
for the comparisons so that the GPU tests (tests/test_gpu_configs.py: asserts + parity_log) and bench.py (the `parity` object and the
`cpu_baseline` of every config in the JSON line, outside the timed region) count the same things the same way.

Each function returns (parity dict, cpu dict): parity = what differs, counted; cpu = the oracle's rays and seconds on this sample."""
import time

import numpy as np

import scenes
from parity_util import rel_err


def _rows_gpu(rta, ctx, g, cam, rows, fn="whitted", **pk):
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


def c3_rows(rta, ctx, g, o, rows=(200, 700, 1000), spp=3, seed=7, threads=0):
    """C3: depth of field, 10x10 area light, 1920x1080: whole rows at `spp` lens samples (the oracle needs ~0.1 s per row and sample)"""
    cam = rta.Camera(**scenes.C3_CAMERA)
    assert (cam.width, cam.height) == (1920, 1080)
    ocam = scenes.oracle_camera(scenes.C3_CAMERA)
    got = _rows_gpu(rta, ctx, g, cam, rows, fn="focusing", background=(200, 200, 200), seed=seed, spp=spp, lens_radius=0.1)
    worst, n_off, n_all, dt = 0.0, 0, 0, 0.0
    for r in rows:
        t = time.perf_counter()
        ref = o.render_focusing(ocam, 0.1, spp, background=(200, 200, 200), seed=seed, first=r * cam.width, count=cam.width, threads=threads)
        dt += time.perf_counter() - t
        e = rel_err(np.minimum(255.0, got[r][0]), ref).max(axis=1)
        n_off += int((e > 1e-3).sum()); n_all += len(e)
        worst = max(worst, float(e[e <= 1e-3].max()) if (e <= 1e-3).any() else 0.0)
    parity = {"sample": "rows %s of the 1920x1080 frame x %d lens samples (of 1024), 100 shadow rays each" % (list(rows), spp), "pixels": n_all,
              "pixels_off_by_more_than_1e-3": n_off, "max_rel_colour_err_elsewhere": worst,
              "bar": "colours within 1e-3 relative; a lens sample whose camera ray ties between two objects may differ: counted"}
    return parity, {"rays": n_all * spp * 101, "seconds": dt}


def c4_rows(rta, ctx, g, o, rows=(150, 400, 640, 900), threads=0):
    """C4: WallAndPaper with WallLight.obj, depth 2, 1920x1080: whole rows, one un-jittered sample"""
    cam = rta.Camera(**scenes.C4_CAMERA)
    assert (cam.width, cam.height) == (1920, 1080)
    ocam = scenes.oracle_camera(scenes.C4_CAMERA)
    got = _rows_gpu(rta, ctx, g, cam, rows, background=(30, 30, 30))
    n_bad = n_all = n_patch = n_mesh = n_off = 0
    worst, dt, rays = 0.0, 0.0, 0
    for r in rows:
        t = time.perf_counter()
        res = o.render_whitted(ocam, background=(30, 30, 30), first=r * cam.width, count=cam.width, counters=True, threads=threads)
        dt += time.perf_counter() - t
        ref_rgb, ref_ids, cnt = res
        rays += int(cnt[0])
        rgb, ids = got[r]
        same = ids == ref_ids
        e = rel_err(rgb[same], ref_rgb[same]).max(axis=1)
        n_bad += int((~same).sum()); n_all += len(ids); n_patch += int((ref_ids == 1).sum()); n_mesh += int((ref_ids >= 2).sum())
        n_off += int((e > 1e-3).sum())
        worst = max(worst, float(e.max()) if e.size else 0.0)
    parity = {"sample": "rows %s of the 1920x1080 frame, 1 sample (of 256), depth 2" % (list(rows),), "pixels": n_all, "id_mismatches": n_bad,
              "patch_pixels": n_patch, "mesh_pixels": n_mesh, "pixels_off_by_more_than_1e-3": n_off, "max_rel_colour_err": worst,
              "bar": "ids: ties <= 1e-4 of the rays; colours within 1e-3 relative"}
    return parity, {"rays": rays, "seconds": dt}


def c5_eye_rows(rta, ctx, g, o, rows=(420, 650), seed=2, threads=0):
    """C5: the Whitted part of the eye pass (depth 8 through three glass cups, 25 shadow rays per hit), whole rows of the 1920x1080 frame"""
    cam = rta.Camera(**scenes.C5_CAMERA)
    assert (cam.width, cam.height) == (1920, 1080)
    ocam = scenes.oracle_camera(scenes.C5_CAMERA)
    got = _rows_gpu(rta, ctx, g, cam, rows, background=(10, 10, 10), seed=seed)
    n_bad = n_all = n_off = n_cup = 0
    dt, rays = 0.0, 0
    for r in rows:
        t = time.perf_counter()
        ref_rgb, ref_ids, cnt = o.render_whitted(ocam, background=(10, 10, 10), seed=seed, first=r * cam.width, count=cam.width, counters=True, threads=threads)
        dt += time.perf_counter() - t
        rays += int(cnt[0])
        rgb, ids = got[r]
        same = ids == ref_ids
        e = rel_err(rgb[same], ref_rgb[same]).max(axis=1)
        n_bad += int((~same).sum()); n_all += len(ids); n_off += int((e > 1e-3).sum()); n_cup += int((ref_ids >= 3).sum())
    parity = {"sample": "rows %s of the 1920x1080 eye pass, depth 8" % (list(rows),), "pixels": n_all, "cup_pixels": n_cup, "id_mismatches": n_bad,
              "pixels_off_by_more_than_1e-3": n_off,
              "bar": "ids: ties <= 1e-4; colours within 1e-3 relative; a deep ray whose Newton iteration sits on its convergence threshold may branch differently: counted"}
    return parity, {"rays": rays, "seconds": dt}


def c5_photon_map(rta, ctx, g, o, res=24, photon_num=1500, passes=2, seed=4):
    """C5 photon mapping on the same scene at a reduced frame (192x108 for res 24) and photon count: the oracle's photon tracing is
    too slow for 10 M photons.  Hit points by (pixel, ray-tree path); N after the passes; the picture."""
    import oracle_bind as ob
    small = dict(scenes.C5_CAMERA, res=res)
    pm = rta.PhotonMap(ctx, g, rta.Camera(**small), photon_num=photon_num, times=passes, background=(10, 10, 10), seed=seed)
    om = ob.OraclePhotonMap(o, scenes.oracle_camera(small), photon_num=photon_num, times=passes, background=(10, 10, 10), seed=seed)
    dt = 0.0
    for t in range(passes):
        pm.run_pass(t)
        t0 = time.perf_counter(); om.run_pass(t); dt += time.perf_counter() - t0
    hp, hr = pm.hitpoints(), om.hitpoints()
    kp = hp[:, 0].astype(np.int64) * 65536 + hp[:, 1].astype(np.int64)
    kr = hr[:, 0].astype(np.int64) * 65536 + hr[:, 1].astype(np.int64)
    common, ip, ir = np.intersect1d(kp, kr, return_indices=True)
    only = len(kp) + len(kr) - 2 * len(common)
    n_diff_N = int((np.abs(hp[ip, 10] - hr[ir, 10]) > 1e-9).sum())
    pos = np.abs(hp[ip, 2:5] - hr[ir, 2:5]).max(axis=1)
    dpic = np.abs(pm.picture() - om.picture()).max(axis=1)
    oc = om.counters()
    parity = {"sample": "%dx%d frame, depth 8, %d photons x %d passes (the config: 1920x1080, 10 M x 64)" % (8 * res, int(4.5 * res), pm.info()["photons_per_pass"], passes),
              "hit_points": int(len(hp)), "hit_points_oracle": int(len(hr)), "hit_points_not_shared": int(only), "hit_points_shared": int(len(common)),
              "shared_with_different_N": n_diff_N, "shared_with_position_off_by_more_than_1e-6": int((pos > 1e-6).sum()),
              "picture_pixels": int(len(dpic)), "picture_pixels_off_by_more_than_2": int((dpic > 2).sum()),
              "photon_segments": int(pm.info()["photon_segments"]), "photon_segments_oracle": int(oc["photon_segments"]),
              "bar": "stochastic images within 1e-3 per pixel on the same RNG stream; hit points that differ (a deep ray tree branching differently) are counted"}
    return parity, {"rays": int(oc["photon_segments"]), "seconds": dt}
