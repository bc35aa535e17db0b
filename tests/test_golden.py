"""Golden vectors (tests/golden/golden_r01.npz, made by tests/golden/make_golden.py): the oracle must keep reproducing them
(CPU), and the HIP path must reproduce them through the C ABI (GPU)."""
import os
import zlib

import numpy as np
import pytest

import scenes

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_r01.npz"))


def _c1_ids():
    return np.frombuffer(zlib.decompress(G["c1_ids_zlib"].tobytes()), dtype=np.int16).astype(np.int32)


def test_golden_c1_ids_carry_the_survey_hash():
    ids = _c1_ids()
    assert ids.size == 640 * 480 and (ids >= 2).sum() == 11914 and (ids < 0).sum() == 0
    assert scenes.fnv_hash_ids(ids) == "83db7884e5084a37"


def test_oracle_reproduces_golden():
    import oracle_bind as ob
    _, o = scenes.build_icosahedron(material="plastic", light="point_area", gpu=False)
    rays, W, H = ob.OracleScene(0.0).camera_rays(scenes.oracle_camera(scenes.C1_CAMERA))
    ids, pts = o.trace_closest(rays, use_bsp=True)
    assert (ids == _c1_ids()).all()
    grid = G["c1_grid"]
    assert np.abs(np.linalg.norm(pts[grid] - rays[grid, :3], axis=1) - G["c1_grid_t"]).max() < 1e-12
    _, o4 = scenes.build_paper(gpu=False, mesh="icosahedron")
    rays4, _, _ = ob.OracleScene(0.0).camera_rays(scenes.oracle_camera(dict(scenes.C4_CAMERA, res=10)))
    ids4, pts4, uv4 = o4.trace_closest(rays4, use_bsp=True, want_uv=True)
    assert (ids4 == G["c4_ids"]).all() and np.abs(uv4 - G["c4_uv"]).max() < 1e-12
    _, op = scenes.build_icosahedron(material="plastic", light="point_area", extra_sphere=True, gpu=False)
    cp = dict(scenes.C1_CAMERA, hor=3.2, ver=2.4, res=15, depth=2)
    pm = ob.OraclePhotonMap(op, scenes.oracle_camera(cp), photon_num=5000, times=2, background=(10, 10, 10), seed=5)
    for t in range(2):
        pm.run_pass(t)
        assert np.abs(pm.hitpoints()[:, [0, 1, 9, 10]] - G["ppm_pass%d_key_R_N" % t]).max() < 1e-12
    assert (pm.picture().astype(np.uint8) == G["ppm_picture"]).all()


GB = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_r01b.npz"))
AA_CAM = dict(scenes.C1_CAMERA, hor=3.2, ver=2.4, res=20, depth=2)
TEX_CAM = dict(scenes.TEX_CAMERA, res=10, depth=2)


def test_oracle_reproduces_golden_aa_textures_mesh():
    _, o = scenes.build_icosahedron(material="metal", light="point_area", extra_sphere=True, gpu=False)
    rgb, mask = o.render_antialiasing(scenes.oracle_camera(AA_CAM), oversample_times=8, background=(200, 200, 200), seed=3)
    assert (mask == GB["aa_mask"]).all() and np.abs(rgb - GB["aa_rgb"]).max() < 1e-9
    rgbf, maskf = o.render_antialiasing(scenes.oracle_camera(AA_CAM), focusing=True, radius=0.1, spp=3, oversample_times=4, background=(200, 200, 200), seed=3)
    assert (maskf == GB["aa_focus_mask"]).all() and np.abs(rgbf - GB["aa_focus_rgb"]).max() < 1e-9
    g, ot = scenes.build_textured()
    trgb, tids = scenes.oracle_render_whitted(ot, TEX_CAM, background=(20, 30, 40), seed=1)
    assert (tids == GB["tex_ids"]).all() and np.abs(trgb - GB["tex_rgb"]).max() < 1e-9
    tg = g.triangles()                                       # host mirror's RotationalSurfaceToMesh: the golden vertices, bit for bit
    assert (tg["v"].reshape(-1, 9) == GB["mesh_v"]).all()
    has = tg["has_vn"] != 0
    assert (tg["vn"].reshape(-1, 9)[has] == GB["mesh_vn"][has]).all()


@pytest.mark.gpu
def test_gpu_reproduces_golden_aa_textures(ctx):
    import ray_tracing_amd as rta
    g, _ = scenes.build_icosahedron(material="metal", light="point_area", extra_sphere=True, oracle=False)
    cam = rta.Camera(**AA_CAM)
    cam.set_seed(3)
    cam.render_raytracer_antialiasing(ctx, g, background=(200, 200, 200), oversample_times=8)
    assert (cam.edge_mask() == GB["aa_mask"]).mean() >= 1 - 1e-3
    ok = cam.edge_mask() == GB["aa_mask"]
    assert (np.abs(cam.rgb() - GB["aa_rgb"]) / np.maximum(1.0, np.abs(GB["aa_rgb"])))[ok].max() < 1e-3
    cam.render_focusing_antialiasing(ctx, g, 0.1, background=(200, 200, 200), spp=3, oversample_times=4)
    ok = cam.edge_mask() == GB["aa_focus_mask"]
    assert ok.mean() >= 1 - 1e-3
    assert (np.abs(cam.rgb() - GB["aa_focus_rgb"]) / np.maximum(1.0, np.abs(GB["aa_focus_rgb"])))[ok].max() < 1e-3
    gt, _ = scenes.build_textured(oracle=False)
    camt = rta.Camera(**TEX_CAM)
    camt.set_seed(1)
    camt.render_raytracer(ctx, gt, background=(20, 30, 40))
    assert (camt.prim_id() == GB["tex_ids"]).all()
    err = (np.abs(camt.rgb() - GB["tex_rgb"]) / np.maximum(1.0, np.abs(GB["tex_rgb"]))).max(axis=1)
    assert (err >= 1e-3).mean() <= 1e-3                      # texel-boundary pixels


@pytest.mark.gpu
def test_gpu_reproduces_golden(ctx):
    import oracle_bind as ob
    import ray_tracing_amd as rta
    # C1 ids and t on the grid; Whitted depth 3 colours on the grid
    g, _ = scenes.build_icosahedron(material="plastic", light="point_area", oracle=False)
    rays, W, H = ob.OracleScene(0.0).camera_rays(scenes.oracle_camera(scenes.C1_CAMERA))
    hits = g.first_inter_pt(ctx, rays, use_tree=True)
    assert (hits["prim_id"] == _c1_ids()).all()
    grid = G["c1_grid"]
    assert np.abs(hits["t"][grid] - G["c1_grid_t"]).max() < 1e-6
    g3, _ = scenes.build_icosahedron(material="metal", light="point_area", oracle=False)
    cam = rta.Camera(**dict(scenes.C1_CAMERA, depth=3))
    cam.render_raytracer(ctx, g3, background=(200, 200, 200))
    ref = G["c1_metal_depth3_grid_rgb"]
    assert (np.abs(cam.rgb()[grid] - ref) / np.maximum(1.0, np.abs(ref))).max() < 1e-3
    # C4 patch roots
    g4, _ = scenes.build_paper(oracle=False, mesh="icosahedron")
    rays4, _, _ = ob.OracleScene(0.0).camera_rays(scenes.oracle_camera(dict(scenes.C4_CAMERA, res=10)))
    h4 = g4.first_inter_pt(ctx, rays4, use_tree=True)
    assert (h4["prim_id"] == G["c4_ids"]).all()
    hit = G["c4_ids"] >= 0
    assert np.abs(h4["t"][hit] - G["c4_t"][hit]).max() < 1e-6
    par = G["c4_ids"] == 1
    assert np.abs(h4["u"][par] - G["c4_uv"][par, 0]).max() < 1e-6 and np.abs(h4["v"][par] - G["c4_uv"][par, 1]).max() < 1e-6
    # C5 cup, recorded and unrecorded
    g5, _ = scenes.build_wine(oracle=False)
    rays5, _, _ = ob.OracleScene(0.0).camera_rays(scenes.oracle_camera(dict(scenes.C5_CAMERA, res=6)))
    for rec in (1, 0):
        h5 = g5.first_inter_pt(ctx, rays5, use_tree=False, record=bool(rec))
        ids5 = G["c5_ids_rec%d" % rec]
        assert (h5["prim_id"] != ids5).mean() <= 1e-4
        ok = (ids5 >= 0) & (h5["prim_id"] == ids5)
        assert np.abs(h5["t"][ok] - G["c5_t_rec%d" % rec][ok]).max() < 1e-6
    # photon map
    gp, _ = scenes.build_icosahedron(material="plastic", light="point_area", extra_sphere=True, oracle=False)
    camp = rta.Camera(**dict(scenes.C1_CAMERA, hor=3.2, ver=2.4, res=15, depth=2))
    pm = rta.PhotonMap(ctx, gp, camp, photon_num=5000, times=2, background=(10, 10, 10), seed=5)
    for t in range(2):
        pm.run_pass(t)
        ref = G["ppm_pass%d_key_R_N" % t]
        got = pm.hitpoints()[:, [0, 1, 9, 10]]
        assert got.shape == ref.shape and (got[:, :2] == ref[:, :2]).all()
        assert (np.abs(got[:, 3] - ref[:, 3]) > 1e-9).mean() < 1e-3
    d = np.abs(pm.picture() - G["ppm_picture"].astype(np.float64))
    assert d.max() <= 2 and (d > 0).mean() < 5e-3
