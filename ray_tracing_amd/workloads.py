"""Synthetic workloads of the BASELINE configs and of the layouts they come from: geometry
generators, cameras and scene builders.  bench.py builds its scenes with these; the tests build the same scenes on the
product AND on their checker through tests/scenes.py.

A builder adds objects, in a fixed order (primitive id = position), to a Stage: one or more scene objects that share
the builder interface of ray_tracing_amd.Scene (add_plane, add_triangles, read_obj_triangle, construct_bsp, ...).  The
product uses a one-sided stage (product_stage); this module knows no other scene type.
"""
import os

import numpy as np

# The meshes of the reference's layouts (ObjFiles/Icosahedron_Triangle.obj; ObjFiles/WallLight.obj, 11,576 faces, Layout.h:125) are data the
# CALLER brings: the package ships none and reaches into no other directory.  bench.py and the tests point this at the repository's fixtures
# (tests/golden/assets) with set_asset_dir(); a user points it at the reference's ObjFiles/.
_ASSET_DIR = os.environ.get("RT_ASSET_DIR")


def set_asset_dir(path):
    """the directory that holds Icosahedron_Triangle.obj and WallLight.obj for the builders below"""
    global _ASSET_DIR
    _ASSET_DIR = path


def asset(name):
    if not _ASSET_DIR:
        raise RuntimeError("workloads: no asset directory (set_asset_dir(<dir with %s>) or RT_ASSET_DIR): the layouts' meshes are the caller's data" % name)
    return os.path.join(_ASSET_DIR, name)


def icosa_obj():
    return asset("Icosahedron_Triangle.obj")


def wall_light_obj():
    return asset("WallLight.obj")


class Stage:
    """forwards scene-building calls to every side; a side = (scene object, material lookup by preset name)"""

    def __init__(self, sides):
        self.sides = list(sides)

    def mat(self, name):
        return tuple(lookup(name) for _, lookup in self.sides)

    def call(self, fn, *args, mat=None, colour=None, **kw):
        out = None
        for k, (side, _) in enumerate(self.sides):
            a = list(args)
            if mat is not None:
                a += [mat[k], colour]
            out = getattr(side, fn)(*a, **kw)
        return out

    def each(self, fn):
        for side, _ in self.sides:
            fn(side)


def product_stage(ambient):
    import ray_tracing_amd as rta
    return Stage([(rta.Scene(ambient), rta.material)])


def build_icosahedron(make_stage, material="plastic", light="point_area", bsp=True, leaf_refine=None,
                      extra_sphere=False, planes=True):
    """C1: Layout3_Icosahedron (Layout.h:136-160) without textures.
    light: 'point_area' = AddAreaLight((-10,20,20),1,0,1) (deterministic), 'area10' = Layout.h:155,
           'point' / 'parallel' use the brute-force shadow rules of Light.h."""
    P = make_stage(0.5)
    plastic = P.mat("plastic")
    if planes:      # planes=False: primary rays can leave the scene
        P.call("add_plane", (0, 0, 0), (0, 0, 1), mat=plastic, colour=(255, 255, 255))
        P.call("add_plane", (10, 0, 0), (-1, 0, 0), mat=plastic, colour=(0, 0, 0))
    P.call("read_obj_triangle", icosa_obj(), (0, -1, 0), 0.0015, mat=P.mat(material), colour=(241, 46, 24))
    if extra_sphere:
        P.call("add_sphere", (0.3, 0.6, 0.5), 0.5, mat=P.mat("glass"), colour=(161, 143, 81))
        P.call("add_sphere", (-1.0, -2.2, 0.35), 0.35, mat=P.mat("mirror"), colour=(255, 255, 255))
        P.call("add_circle", (0.5, 1.8, 0.4), (0.2, -0.5, 1), 0.6, mat=P.mat("porcelain"), colour=(20, 90, 200))
    if bsp:
        P.call("construct_bsp")
    if light == "point_area":
        P.call("add_area_light", (-10, 20, 20), 1.0, 0.0, 1)
    elif light == "area10":
        P.call("add_area_light", (-10, 20, 20), 1.0, 10.0, 10)
    elif light == "area3":
        P.call("add_area_light", (-10, 20, 20), 1.0, 10.0, 3)
    elif light == "point":
        P.call("add_point_light", (-10, 20, 20), 1.0)
    elif light == "parallel":
        P.call("add_parallel_light", (1, -1, -1), 1.0)
    elif light == "two":
        P.call("add_area_light", (-10, 20, 20), 0.6, 4.0, 2)
        P.call("add_parallel_light", (1, 1, -1), 0.5)
    if leaf_refine is not None:
        P.each(lambda s: s.set_leaf_refine(leaf_refine) if hasattr(s, "set_leaf_refine") else None)
    return P


C3_CAMERA = dict(view=(-10, 0, 1.5), eye=(1, 0, 0), left=(0, -1, 0), dist=10, hor=6, ver=3.375, res=320, depth=1)     # Layout.h:157 re-framed to 1080p
C1_CAMERA = dict(view=(-10, 0, 1.5), eye=(1, 0, 0), left=(0, -1, 0), dist=10, hor=6.4, ver=4.8, res=100, depth=1)


def _wave(t):
    """a period-1 stand-in for sin(2 pi t) built from + - x and floor only (two parabola arcs), so that the mesh is the
    same bit for bit on every host, in the container and on the device: SURVEY 8d asks for exactly that"""
    f = t - np.floor(t)
    return np.where(f < 0.5, 16.0 * f * (0.5 - f), -16.0 * (f - 0.5) * (1.0 - f))


def heightfield_heights(X, Y, libm_free=True):
    """three octaves, amplitude ~0.9.  libm_free=False: the surveyor's probe function with sin / cos
    (host libm), kept because SURVEY's recorded tree statistics of the reference belong to that mesh."""
    if not libm_free:
        return 0.6 * np.sin(1.7 * X) * np.cos(2.3 * Y) + 0.25 * np.sin(5.1 * X + 1.3) * np.sin(4.3 * Y + 0.7) + 0.05 * np.sin(23 * X) * np.cos(19 * Y)
    k = 0.15915494309189535          # 1 / (2 pi): a constant, not a libm call
    S = lambda a: _wave(a * k)
    Cc = lambda a: _wave(a * k + 0.25)
    return 0.6 * S(1.7 * X) * Cc(2.3 * Y) + 0.25 * S(5.1 * X + 1.3) * S(4.3 * Y + 0.7) + 0.05 * S(23 * X) * Cc(19 * Y)


def heightfield_triangles(n_cells=708, half=5.0, libm_free=True):
    """C2 mesh: heightfield on [-half,half]^2, n_cells per side, 2 triangles per cell, order i-major,
    j-minor, (P(i,j),P(i+1,j),P(i+1,j+1)) then (P(i,j),P(i+1,j+1),P(i,j+1)); 708 cells -> 1,002,528 triangles."""
    idx = np.arange(n_cells + 1, dtype=np.float64)
    xs = -half + (2 * half) * idx / n_cells
    X, Y = np.meshgrid(xs, xs, indexing="ij")
    Z = heightfield_heights(X, Y, libm_free)
    Pnt = np.stack([X, Y, Z], axis=-1)
    p00 = Pnt[:-1, :-1]
    p10 = Pnt[1:, :-1]
    p11 = Pnt[1:, 1:]
    p01 = Pnt[:-1, 1:]
    t1 = np.stack([p00, p10, p11], axis=2)
    t2 = np.stack([p00, p11, p01], axis=2)
    tris = np.stack([t1, t2], axis=2)          # (n, n, 2, 3, 3)
    return np.ascontiguousarray(tris.reshape(-1, 9))


C2_CAMERA = dict(view=(-12, -9, 7), eye=(12, 9, -7), left=(9, -12, 0), dist=20, hor=19.2, ver=10.8, res=100, depth=1)


def c2_orbit_camera(frame):
    """frame 0 = the C2 camera of SURVEY 8d; the others swing around it by at most 1.8 degrees, so that every frame is
    the C2 view (same cost) but no two consecutive frames render the same rays (+ - x / and a 3-term series: no libm)"""
    a = 0.001 * (frame % 32)
    ca, sa = 1.0 - a * a / 2.0, a - a * a * a / 6.0
    c = dict(C2_CAMERA)
    vx, vy, vz = c["view"]
    c["view"] = (vx * ca - vy * sa, vx * sa + vy * ca, vz)
    c["eye"] = tuple(-x for x in c["view"])
    c["left"] = (c["eye"][1], -c["eye"][0], 0.0)               # frame 0: exactly C2_CAMERA
    return c


def build_heightfield(make_stage, n_cells=708, leaf_refine=None, max_obj_num=0, light=True, libm_free=True):
    """C2: ground plane z=-1 (id 0) + heightfield, plastic, point-like area light."""
    P = make_stage(0.5)
    plastic = P.mat("plastic")
    P.call("add_plane", (0, 0, -1), (0, 0, 1), mat=plastic, colour=(200, 200, 200))
    tris = heightfield_triangles(n_cells, libm_free=libm_free)
    P.call("add_triangles", tris, mat=plastic, colour=(90, 160, 60))
    P.call("construct_bsp", max_obj_num, 0)
    if light:
        P.call("add_area_light", (-10, 20, 20), 1.0, 0.0, 1)
    if leaf_refine is not None:
        P.each(lambda s: s.set_leaf_refine(leaf_refine) if hasattr(s, "set_leaf_refine") else None)
    return P


def fnv_hash_ids(ids):
    """the hash of SURVEY 8c over (id+1) in raster order"""
    h = 1469598103934665603
    M = (1 << 64) - 1
    for i in np.asarray(ids).tolist():
        h ^= (i + 1) & 0xFFFFFFFF
        h = (h * 1099511628211) & M
    return "%016x" % h


# ---- parametric objects (SURVEY 8d C4 / C5 geometry, untextured) ---------------------------------------------
PAPER_CTRL = [[(-2, 6, 9), (-1, 2, 9), (-1, -2, 9), (-2, -6, 9)],
              [(-1, 6, 3), (0, 2, 3), (0, -2, 3), (-1, -6, 3)],
              [(-1, 6, -3), (0, 2, -2), (0, -2, -3), (-1, -6, -3)],
              [(-2, 6, -9), (-1, 2, -9), (-1, -2, -9), (-2, -6, -9)]]      # Layout.h:113-117
PAPER_CENTER = (20, 25, 17)                                               # Layout.h:118

C4_CAMERA = dict(view=(-10, 20, 20), eye=(1, 0, 0), left=(-0.8, -1, 0), dist=5, hor=6, ver=3.375, res=320, depth=2)


def build_paper(make_stage, bsp=True, mesh="walllight", material="rubber"):
    """C4: Layout2_WallAndPaper (Layout.h:106-132) untextured: wall plane, the bicubic "paper" patch and the
    WallLight.obj mesh (11,576 triangles, ReadObjFile_Triangle(..., (20,13,20), 0.7, _METAL_), Layout.h:125-126), parallel
    light (1,-1,-1) L=2: 11,578 objects.  mesh="icosahedron": the small stand-in of round 1 (1,369 triangles beside the
    patch) for quick tests; mesh=None: plane + patch only."""
    P = make_stage(0.05)
    P.call("add_plane", (20, 0, 0), (-1, 0, 0), mat=P.mat("rubber"), colour=(120, 120, 120))
    P.call("add_bezier", PAPER_CTRL, PAPER_CENTER, mat=P.mat(material), colour=(255, 255, 255))
    if mesh == "walllight":
        P.call("read_obj_triangle", wall_light_obj(), (20, 13, 20), 0.7, mat=P.mat("metal"), colour=(50, 50, 60))
    elif mesh == "icosahedron" or mesh is True:
        P.call("read_obj_triangle", icosa_obj(), (18.5, 13, 20), 0.004, mat=P.mat("metal"), colour=(50, 50, 60))
    P.call("add_parallel_light", (1, -1, -1), 2.0)
    if bsp:
        P.call("construct_bsp")
    return P


CUP2_PROFILES = [([(2, 0.2), (0.2, 0.2), (0.2, 2.2), (0.2, 4.2)], 50),          # Object_Cup2, Layout.h:190-235
                 ([(0.2, 4.2), (0.2, 6.2), (2, 6.2), (1.7, 10.2)], 50),
                 ([(1.6, 10.2), (1.65, 8.2), (1.3, 6.2), (0, 6.2)], 50),
                 ([(2, 0), (2.1, 0.1), (2, 0.2)], 10),
                 ([(1.7, 10.2), (1.65, 10.3), (1.6, 10.2)], 10)]

C5_CAMERA = dict(view=(-23.5, 27.5, 24), eye=(2.9, -3, -2), left=(-1, -1, 0), dist=15, hor=8, ver=4.5, res=240, depth=8)


WINE_CUPS = (((-2, -2, 0), 0.7, (151, 57, 29)), ((2.2, 2.2, 0), 0.5, (100, 25, 30)), ((5, 5, 0), 0.8, (100, 25, 30)))   # Layout.h:255-286


def _bezier_point(ctrl, t):
    """CBezierCurve::Point2D by de Casteljau (Bezier.h:27-43)"""
    pts = np.array(ctrl, dtype=np.float64)
    while len(pts) > 1:
        pts = pts[:-1] * (1 - t) + pts[1:] * t
    return float(pts[0][0]), float(pts[0][1])


def build_wine(make_stage, cups=((-2, -2, 0),), light="point", cup_material="glass", u_samples=None):
    """C5 geometry: Layout5_Wine (Layout.h:237-300) untextured and WITHOUT a tree: three planes, per cup the
    five surfaces of revolution of Object_Cup2, the wine surface and its circle lid.  cups: centres (every wine surface
    with uRange 0.7, as the first cup), or "layout" = the three cups of Layout.h:255-286 with uRange 0.7 / 0.5 / 0.8 and
    their colours (all three wine surfaces rotate bezierCurve1: the layout passes it to rObj2 and rObj3 as well)."""
    e = 1e-8
    P = make_stage(0.5)
    P.call("add_plane", (0, 0, 0), (0, 0, 1), mat=P.mat("rubber"), colour=(140, 100, 60))
    P.call("add_plane", (10, 0, 0), (-1, 0, 0), mat=P.mat("plastic"), colour=(90, 90, 110))
    P.call("add_plane", (0, -10, 0), (0, 1, 0), mat=P.mat("plastic"), colour=(90, 110, 90))
    wine = [(0, 6.2 - 10 * e), (1.3 + 10 * e, 6.2 - 10 * e), (1.65 + 10 * e, 8.2 - 10 * e), (1.6 + 10 * e, 10.2 - 10 * e)]
    spec = WINE_CUPS if cups == "layout" else tuple((c, 0.7, (151, 57, 29)) for c in cups)
    for c, u_range, wine_colour in spec:
        for prof, ns in CUP2_PROFILES:
            P.call("add_rotational", prof, c, (u_samples or ns), 1.0, mat=P.mat(cup_material), colour=(155, 155, 155))
        P.call("add_rotational", wine, c, (u_samples or 50), u_range, mat=P.mat("ice"), colour=wine_colour)
        pu, pv = _bezier_point(wine, u_range)                        # CCircle at the curve point of u = uRange (Layout.h:262-264)
        P.call("add_circle", (c[0], c[1], pv), (0, 0, 1), pu + 10 * e, mat=P.mat("ice"), colour=wine_colour)
    if light == "point":
        P.call("add_point_light", (-10, 25, 25), 2.0)
    elif light == "area":
        P.call("add_area_light", (-10, 25, 25), 2.0, 2.0, 2)
    elif light == "area5":                     # AddAreaLight((-10,25,25), 2, 2, 5)
        P.call("add_area_light", (-10, 25, 25), 2.0, 2.0, 5)
    return P


ROOM_CAMERA = dict(view=(-10, 0, 2.5), eye=(1, 0, 0), left=(0, -1, 0), dist=10, hor=8, ver=6, res=100, depth=1)   # Layout.h:320


def build_room(make_stage, mirror=True):
    """Layout6_Room (Layout.h:302-323) with a synthetic corridor: Corridor_Triangle.obj is one of the blobs missing from
    the reference checkout (.MISSING_LARGE_BLOBS), so a floor, a side wall with a window gap and the icosahedron mesh
    stand in for it.  Light shafts come from an XZ rectangle at y = 6 towards (1,-2,-1) as in Layout.h:317-318."""
    P = make_stage(0.1)
    plastic = P.mat("plastic")
    P.call("add_plane", (50, 0, 0), (-1, 0, 0), mat=plastic, colour=(90, 90, 120))
    def quad(a, b, c, d, mat, colour):
        P.call("add_triangle", [a, b, c], None, mat=mat, colour=colour)
        P.call("add_triangle", [a, c, d], None, mat=mat, colour=colour)
    quad((-12, -8, 0), (30, -8, 0), (30, 8, 0), (-12, 8, 0), plastic, (241, 46, 24))             # floor
    quad((-12, 5, 0), (2, 5, 0), (2, 5, 8), (-12, 5, 8), plastic, (200, 200, 200))               # wall, left of the window
    quad((6, 5, 0), (30, 5, 0), (30, 5, 8), (6, 5, 8), plastic, (200, 200, 200))                 # wall, right of the window
    quad((2, 5, 0), (6, 5, 0), (6, 5, 1), (2, 5, 1), plastic, (200, 200, 200))                   # sill
    P.call("read_obj_triangle", icosa_obj(), (9, -2, 0), 0.003, mat=P.mat("metal" if mirror else "plastic"), colour=(241, 46, 24))
    P.call("construct_bsp")
    P.call("add_area_light", (-10, 20, 20), 2.0, 10.0, 2)
    return P


# ---- textures and RotationalSurfaceToMesh (N3) ------------------------------------------------------
def synth_texture(rows, cols, seed, kind="colour"):
    """deterministic (rows, cols, 3) uint8 B,G,R images standing in for the reference's JPEG assets: 'colour' = blocks
    of random colours over a gradient, 'bump' = a normal map whose blue (N) channel stays dominant"""
    rng = np.random.default_rng(seed)
    r = np.arange(rows)[:, None]; c = np.arange(cols)[None, :]
    if kind == "colour":
        blocks = rng.integers(30, 255, size=((rows + 7) // 8, (cols + 7) // 8, 3))
        img = np.kron(blocks, np.ones((8, 8, 1), dtype=np.int64))[:rows, :cols]
        img = (img * 0.7 + 0.3 * ((r * 255 // max(1, rows - 1) + c * 255 // max(1, cols - 1)) // 2)[..., None]).astype(np.uint8)
        return np.ascontiguousarray(img)
    img = np.zeros((rows, cols, 3), dtype=np.uint8)
    img[..., 0] = 200 + rng.integers(0, 55, size=(rows, cols))                       # B -> N
    img[..., 1] = (127.5 + 90 * np.sin(r * 0.7) * np.cos(c * 0.45)).astype(np.uint8)  # G -> tangent
    img[..., 2] = (127.5 + 90 * np.cos(r * 0.31 + c * 0.53)).astype(np.uint8)         # R -> bitangent
    return img


APPLE_PROFILE = [(0, 0.3), (2.2, -0.9), (2.6, 2.4), (0.05, 2.0)]       # a closed-ish blob profile (radius, height)
TEX_CAMERA = dict(view=(-14, 3, 7), eye=(1, -0.1, -0.3), left=(0, -1, 0), dist=10, hor=9.6, ver=7.2, res=40, depth=3)


def build_textured(make_stage, bsp=True, light="point_area"):
    """every texture path of Object.h on one stage: textured + bump-mapped planes (Layout.h:76-87), textured and
    bump-mapped spheres (Layout.h:171-175), a textured bump-mapped Bezier patch (Layout.h:121-122), a textured
    bump-mapped surface of revolution, a textured mesh from RotationalSurfaceToMesh_Texture (Layout.h:60), a plain
    RotationalSurfaceToMesh, a textured triangle with explicit textureCoord, a bump-mapped triangle (no tangent frame,
    Object.h:63-64) and a textured circle (default UV, Object.h:61)"""
    P = make_stage(0.3)
    t_floor = P.call("add_texture_image", synth_texture(96, 80, 1))
    t_wall = P.call("add_texture_image", synth_texture(64, 64, 2))
    t_ball = P.call("add_texture_image", synth_texture(50, 100, 3))
    t_skin = P.call("add_texture_image", synth_texture(40, 60, 4))
    b_floor = P.call("add_texture_image", synth_texture(48, 48, 5, "bump"))
    b_ball = P.call("add_texture_image", synth_texture(32, 64, 6, "bump"))
    plastic, metal, porcelain = P.mat("plastic"), P.mat("metal"), P.mat("porcelain")
    floor = P.call("add_plane", (0, 0, 0), (0, 0, 1), mat=porcelain, colour=(255, 255, 255))
    P.call("add_texture", floor, t_floor, 5.0); P.call("add_bump_texture", floor, b_floor, 3.0)
    wall = P.call("add_plane", (12, 0, 0), (-1, 0, 0), mat=plastic, colour=(255, 255, 255))
    P.call("add_texture", wall, t_wall, 2.0); P.call("add_bump_texture", wall, b_floor, 5.0)
    side = P.call("add_plane", (0, -9, 0), (0.3, 1, 0.1), mat=plastic, colour=(255, 255, 255))      # a tilted normal: the other u0 branches
    P.call("add_texture", side, t_wall, 7.0)
    s1 = P.call("add_sphere", (2, 5, 1.5), 1.5, mat=plastic, colour=(255, 255, 255))
    P.call("add_texture", s1, t_ball)
    s2 = P.call("add_sphere", (0, 1.5, 1.2), 1.2, mat=metal, colour=(230, 200, 120))
    P.call("add_bump_texture", s2, b_ball)
    ctrl = (np.array(PAPER_CTRL, dtype=np.float64) * 0.3).tolist()
    paper = P.call("add_bezier", ctrl, (9, -3, 4), mat=P.mat("rubber"), colour=(255, 255, 255))
    P.call("add_texture", paper, t_skin); P.call("add_bump_texture", paper, b_ball)
    cup = P.call("add_rotational", [(1.2, 0), (0.4, 1.0), (1.6, 2.0), (1.1, 3.2)], (4, -2, 0), 20, 1.0, mat=porcelain, colour=(255, 255, 255))
    P.call("add_texture", cup, t_skin); P.call("add_bump_texture", cup, b_ball)
    P.call("rotational_surface_to_mesh", APPLE_PROFILE, 5, 15, (1, -5, 0.9), mat=plastic, colour=(255, 255, 255), tex=t_ball)
    P.call("rotational_surface_to_mesh", APPLE_PROFILE, 4, 9, (6, 4, 0.9), mat=metal, colour=(80, 160, 90))
    tri = P.call("add_triangle_tex", [(3, 0, 0.02), (7, 0.5, 0.02), (5, 3, 2.5)], None, [(0, 0), (39, 0), (20, 59)], mat=plastic, colour=(255, 255, 255))
    P.call("add_texture", tri, t_skin); P.call("add_bump_texture", tri, b_ball)
    circ = P.call("add_circle", (-2, -2.5, 0.6), (0.1, 0.2, 1), 1.3, mat=plastic, colour=(10, 10, 10))
    P.call("add_texture", circ, t_wall)
    if bsp:
        P.call("construct_bsp")
    if light == "point_area":
        P.call("add_area_light", (-10, 12, 20), 1.2, 0.0, 1)
    elif light == "area":
        P.call("add_area_light", (-10, 12, 20), 1.2, 4.0, 2)
    return P
