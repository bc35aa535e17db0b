"""Regenerates the golden vectors in this directory from the oracle (oracle/rt_oracle.c).  The reference cannot be built
here, so these vectors freeze the ORACLE's answers: the C1 numbers they contain are the ones SURVEY.md
records from the reference itself (hit-id hash, tree statistics); the rest is "parity unpinned" data that guards against
regressions of the restatement and lets the GPU tests run against fixed numbers as well as against the live oracle.

    python tests/golden/make_golden.py        (a few seconds, CPU only)
"""
import os, sys, zlib
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np
import scenes, oracle_bind as ob


def main():
    out = {}
    # C1: full hit-id image (int16) + t and RGB on the 8-pixel grid
    _, o = scenes.build_icosahedron(material="plastic", light="point_area", gpu=False)
    cam_args = dict(scenes.C1_CAMERA)
    rays, W, H = ob.OracleScene(0.0).camera_rays(scenes.oracle_camera(cam_args))
    ids, pts = o.trace_closest(rays, use_bsp=True)
    out["c1_ids_zlib"] = np.frombuffer(zlib.compress(ids.astype(np.int16).tobytes(), 9), dtype=np.uint8)
    grid = (np.arange(H)[::8, None] * W + np.arange(W)[None, ::8]).reshape(-1)
    out["c1_grid"] = grid.astype(np.int32)
    out["c1_grid_t"] = np.linalg.norm(pts[grid] - rays[grid, :3], axis=1)
    _, o3 = scenes.build_icosahedron(material="metal", light="point_area", gpu=False)
    rgb, _ = scenes.oracle_render_whitted(o3, dict(cam_args, depth=3), background=(200, 200, 200), seed=0)
    out["c1_metal_depth3_grid_rgb"] = rgb[grid]
    # C4: patch roots on a 60 x 34 ray grid
    _, o4 = scenes.build_paper(gpu=False, mesh="icosahedron")
    c4 = dict(scenes.C4_CAMERA, res=10)
    rays4, W4, H4 = ob.OracleScene(0.0).camera_rays(scenes.oracle_camera(c4))
    ids4, pts4, uv4 = o4.trace_closest(rays4, use_bsp=True, want_uv=True)
    out["c4_ids"] = ids4.astype(np.int32); out["c4_t"] = np.linalg.norm(pts4 - rays4[:, :3], axis=1); out["c4_uv"] = uv4
    # C5: cup (surfaces of revolution), no tree, 48 x 27 rays, recorded and unrecorded variants
    _, o5 = scenes.build_wine(gpu=False)
    c5 = dict(scenes.C5_CAMERA, res=6)
    rays5, W5, H5 = ob.OracleScene(0.0).camera_rays(scenes.oracle_camera(c5))
    for rec in (1, 0):
        i5, p5, u5 = o5.trace_closest(rays5, use_bsp=False, record=bool(rec), want_uv=True)
        out["c5_ids_rec%d" % rec] = i5.astype(np.int32); out["c5_t_rec%d" % rec] = np.linalg.norm(p5 - rays5[:, :3], axis=1); out["c5_uv_rec%d" % rec] = u5
    # tiny photon map: per hit point (pixel, path, R, N) after each of two passes, and the picture
    _, op = scenes.build_icosahedron(material="plastic", light="point_area", extra_sphere=True, gpu=False)
    cp = dict(scenes.C1_CAMERA, hor=3.2, ver=2.4, res=15, depth=2)
    pm = ob.OraclePhotonMap(op, scenes.oracle_camera(cp), photon_num=5000, times=2, background=(10, 10, 10), seed=5)
    for t in range(2):
        pm.run_pass(t)
        h = pm.hitpoints()
        out["ppm_pass%d_key_R_N" % t] = h[:, [0, 1, 9, 10]]
    out["ppm_picture"] = pm.picture().astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, "golden_r01.npz"), **out)
    print("wrote golden_r01.npz:", {k: v.shape for k, v in out.items()})
    more = golden_b()
    np.savez_compressed(os.path.join(HERE, "golden_r01b.npz"), **more)
    print("wrote golden_r01b.npz:", {k: v.shape for k, v in more.items()})


AA_CAM = dict(scenes.C1_CAMERA, hor=3.2, ver=2.4, res=20, depth=2)
TEX_CAM = dict(scenes.TEX_CAMERA, res=10, depth=2)


def golden_b():
    """anti-aliasing drivers, textures, RotationalSurfaceToMesh"""
    out = {}
    _, o = scenes.build_icosahedron(material="metal", light="point_area", extra_sphere=True, gpu=False)
    rgb, mask = o.render_antialiasing(scenes.oracle_camera(AA_CAM), oversample_times=8, background=(200, 200, 200), seed=3)
    out["aa_rgb"] = rgb; out["aa_mask"] = mask
    rgbf, maskf = o.render_antialiasing(scenes.oracle_camera(AA_CAM), focusing=True, radius=0.1, spp=3, oversample_times=4, background=(200, 200, 200), seed=3)
    out["aa_focus_rgb"] = rgbf; out["aa_focus_mask"] = maskf
    _, ot = scenes.build_textured(gpu=False)
    trgb, tids = scenes.oracle_render_whitted(ot, TEX_CAM, background=(20, 30, 40), seed=1)
    out["tex_rgb"] = trgb; out["tex_ids"] = tids.astype(np.int32)
    tris = [t for t in (ot.get_triangle(i) for i in range(ot.num_objects)) if t is not None]
    out["mesh_v"] = np.array([t[0] for t in tris]); out["mesh_vn"] = np.array([t[1] for t in tris])
    return out


R02_PPM_CAM = dict(scenes.C1_CAMERA, hor=3.2, ver=2.4, res=20, depth=3)
R02_ROOM_CAM = dict(scenes.ROOM_CAMERA, res=12, depth=2)


def golden_c():
    """round 2: the configs on their stated geometry (C2's libm-free mesh, C4 with WallLight.obj, C5 with the layout's three cups) and the
    two drivers Layout5_Wine / Layout6_Room call (Render_PhotonMapper_Antialiasing, Render_RayTracer_GodRay_Antialiasing)"""
    out = {}
    # C2: the bench's mesh and camera; hash of the full 1080p hit-id image + ids / t on the 16-pixel grid
    _, o2 = scenes.build_heightfield(708, gpu=False)
    rays, W, H = ob.OracleScene(0.0).camera_rays(scenes.oracle_camera(scenes.C2_CAMERA))
    ids, pts = o2.trace_closest(rays, use_bsp=True)
    out["c2_hash"] = np.frombuffer(scenes.fnv_hash_ids(ids).encode(), dtype=np.uint8)
    out["c2_hits"] = np.array([(ids >= 0).sum(), (ids >= 1).sum()], dtype=np.int64)
    grid = (np.arange(H)[::16, None] * W + np.arange(W)[None, ::16]).reshape(-1)
    out["c2_grid"] = grid.astype(np.int32); out["c2_grid_ids"] = ids[grid].astype(np.int32)
    out["c2_grid_t"] = np.linalg.norm(pts[grid] - rays[grid, :3], axis=1)
    # C4 with the real mesh (11,578 objects), 192 x 108 rays
    _, o4 = scenes.build_paper(gpu=False)
    rays4, _, _ = ob.OracleScene(0.0).camera_rays(scenes.oracle_camera(dict(scenes.C4_CAMERA, res=32)))
    ids4, pts4, uv4 = o4.trace_closest(rays4, use_bsp=True, want_uv=True)
    out["c4_ids"] = ids4.astype(np.int32); out["c4_t"] = np.linalg.norm(pts4 - rays4[:, :3], axis=1); out["c4_uv"] = uv4
    # C5 with the layout's three cups (uRange 0.7 / 0.5 / 0.8), 96 x 54 rays, no tree
    _, o5 = scenes.build_wine(gpu=False, cups="layout", light="area5")
    rays5, _, _ = ob.OracleScene(0.0).camera_rays(scenes.oracle_camera(dict(scenes.C5_CAMERA, res=12)))
    i5, p5, u5 = o5.trace_closest(rays5, use_bsp=False, record=True, want_uv=True)
    out["c5_ids"] = i5.astype(np.int32); out["c5_t"] = np.linalg.norm(p5 - rays5[:, :3], axis=1)
    # Render_PhotonMapper_Antialiasing: mask, hit-point keys, N and R after each pass, picture
    _, op = scenes.build_icosahedron(material="plastic", light="point_area", extra_sphere=True, gpu=False)
    pm = ob.OraclePhotonMap(op, scenes.oracle_camera(R02_PPM_CAM), photon_num=8000, times=2, background=(10, 10, 10), seed=6, aa=dict(oversample_times=4))
    out["ppm_aa_mask"] = pm.edge_mask()
    for t in range(2):
        pm.run_pass(t)
        out["ppm_aa_pass%d_key_R_N" % t] = pm.hitpoints()[:, [0, 1, 9, 10]]
    out["ppm_aa_picture"] = pm.picture().astype(np.uint8)
    # Render_RayTracer_GodRay_Antialiasing on the synthetic room
    _, orr = scenes.build_room(gpu=False)
    og = ob.OracleGodRay(orr, (155, 155, 50), seed=9)
    og.add_area_light_source_xz(6.0, (-3.0, 0.0), (9.0, 10.0), 0.4, (1.0, -2.0, -1.0))
    og.photon_mapping_pass()
    rgb, mask = og.render_antialiasing(scenes.oracle_camera(R02_ROOM_CAM), oversample_times=4, background=(20, 20, 30))
    out["godray_aa_rgb"] = rgb; out["godray_aa_mask"] = mask
    return out


if __name__ == "__main__":
    main()
    c = golden_c()
    np.savez_compressed(os.path.join(HERE, "golden_r02.npz"), **c)
    print("wrote golden_r02.npz:", {k: v.shape for k, v in c.items()})
