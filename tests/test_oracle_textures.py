"""CPU checks of the oracle's texture / bump-texture code (Object.h:79-92, 186-196, 280-319, 725-771) against closed
forms worked out here in numpy, and of RotationalSurfaceToMesh (Scene.h:168-266): host mirror == oracle, bit for bit.
The reference holds no output for these paths (parity unpinned); its JPEG assets are replaced by synthetic arrays."""
import numpy as np

import oracle_bind as ob
import scenes

PI = 3.1415926535      # Geometry.h:13


def _plane_scene(tex=None, tex_scale=1.0, bump=None, bump_scale=1.0, n=(0, 0, 1)):
    o = ob.OracleScene(0.1)
    pl = o.add_plane((0, 0, 0), n, ob.material("plastic"), (255, 255, 255))
    if tex is not None:
        o.add_texture(pl, o.add_texture_image(tex), tex_scale)
    if bump is not None:
        o.add_texture(pl, o.add_texture_image(np.full((2, 2, 3), 255, np.uint8)), 1.0) if tex is None else None
        o.add_bump_texture(pl, o.add_texture_image(bump), bump_scale)
    o.add_point_light((3, -4, 20), 0.5)          # factors stay below 1: no clamp at 255
    return o


CAM = dict(view=(-6, 1, 5), eye=(1, 0, -0.7), left=(0, -1, 0), dist=5, hor=4, ver=3, res=16, depth=1)


def test_plane_texture_lookup_matches_closed_form():
    tex = scenes.synth_texture(37, 53, 11)
    white, _ = scenes.oracle_render_whitted(_plane_scene(), CAM)
    col, ids = scenes.oracle_render_whitted(_plane_scene(tex, 6.5), CAM)
    o = _plane_scene()
    rays, w, h = o.camera_rays(scenes.oracle_camera(CAM))
    _, pts = o.trace_closest(rays)
    hit = ids == 0
    assert hit.sum() > 500
    # Object.h:725-742 with n = (0,0,1): u0 = (1,0,0), v0 = n ^ u0 = (0,1,0); C int() and % semantics
    up = np.trunc(pts[:, 0] * 6.5).astype(np.int64); vp = np.trunc(pts[:, 1] * 6.5).astype(np.int64)
    up = np.fmod(up, 37); vp = np.fmod(vp, 53)
    up = (up + 37) % 37; vp = (vp + 53) % 53
    texel = tex[up, vp][:, ::-1].astype(np.float64)             # B,G,R -> R,G,B
    exp = white * texel / 255.0
    assert np.allclose(col[hit], exp[hit], rtol=1e-12, atol=1e-9)


def test_plane_bump_normal_matches_closed_form():
    bump = np.zeros((4, 4, 3), np.uint8); bump[..., 0] = 255; bump[..., 1] = 255; bump[..., 2] = 0     # B, G, R
    o = _plane_scene(bump=bump, bump_scale=2.0)
    col, ids = scenes.oracle_render_whitted(o, CAM)
    rays, w, h = o.camera_rays(scenes.oracle_camera(CAM))
    _, pts = o.trace_closest(rays)
    # Object.h:88-90: bitangent (n ^ u0 = (0,1,0)) * (0/127.5 - 1) + tangent (1,0,0) * 1 + N * 1
    norm = np.array([1.0, -1.0, 1.0]) / np.sqrt(3.0)
    light = np.array([3.0, -4.0, 20.0])
    ldir = pts - light; ldir /= np.linalg.norm(ldir, axis=1, keepdims=True)
    nn = np.where((rays[:, 3:] @ norm > 1e-8)[:, None], -norm, norm)
    diffuse = 0.6 * np.maximum(0.0, -(ldir * nn).sum(axis=1) * 0.5)       # Scene.h:110, plastic: diff 0.6, spec 0
    exp = np.minimum(255.0, 255.0 * (0.1 * 1.0 + diffuse))                 # ambient 0.1 * material ambient 1
    hit = ids == 0
    assert np.allclose(col[hit, 0], exp[hit], rtol=1e-12, atol=1e-9)


def test_sphere_texture_lookup_matches_closed_form():
    tex = scenes.synth_texture(30, 45, 12)
    def scene(t):
        o = ob.OracleScene(0.1)
        s = o.add_sphere((1, 0.5, 2), 1.5, ob.material("plastic"), (255, 255, 255))
        if t is not None:
            o.add_texture(s, o.add_texture_image(t))
        o.add_point_light((-8, -4, 9), 0.5)
        return o
    cam = dict(view=(-6, 0.5, 2), eye=(1, 0, 0), left=(0, -1, 0), dist=5, hor=3, ver=3, res=20, depth=1)
    white, ids = scenes.oracle_render_whitted(scene(None), cam)
    col, _ = scenes.oracle_render_whitted(scene(tex), cam)
    o = scene(None)
    rays, w, h = o.camera_rays(scenes.oracle_camera(cam))
    _, pts = o.trace_closest(rays)
    q = pts - np.array([1, 0.5, 2.0])
    hit = ids == 0
    v0 = (np.arctan2(q[:, 1], q[:, 0]) + PI) / (2 * PI)
    u0 = (np.arcsin(np.clip(q[:, 2] / 1.5, -1, 1)) + PI / 2) / PI
    row = np.clip(np.trunc(u0 * 30).astype(np.int64), 0, 29); colm = np.clip(np.trunc(v0 * 45).astype(np.int64), 0, 44)
    exp = white * tex[row, colm][:, ::-1] / 255.0
    bad = ~np.isclose(col[hit], exp[hit], rtol=1e-12, atol=1e-9).all(axis=1)
    assert hit.sum() > 500 and bad.mean() < 2e-3           # texel-boundary flips between libm and numpy only


def test_rotational_surface_to_mesh_host_equals_oracle():
    import ray_tracing_amd as rta
    g, o = scenes.build_textured(bsp=False)
    tg = g.triangles()
    tris = [t for t in (o.get_triangle(i) for i in range(o.num_objects)) if t is not None]
    assert len(tg) == len(tris) == 5 * 15 * 2 + 4 * 9 * 2 + 1
    for k, (v, vn) in enumerate(tris):
        assert (tg["v"][k].reshape(9) == v).all(), k
        if tg["has_vn"][k]:
            assert (tg["vn"][k].reshape(9) == vn).all(), k
    assert g.num_objects == o.num_objects and rta is not None
