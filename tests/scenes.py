"""Scene builders shared by tests, smoke() and bench.py.

Every builder adds the SAME objects, in the same order, to a product scene (ray_tracing_amd.Scene,
i.e. the C++ host layer) and to an oracle scene (tests/oracle_bind.OracleScene), so that primitive
ids agree.  Either side may be skipped (gpu=False / oracle=False).
"""
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ICOSA_OBJ = os.path.join(HERE, "golden", "assets", "Icosahedron_Triangle.obj")


class _Pair:
    """forwards scene-building calls to both sides"""

    def __init__(self, ambient, gpu=True, oracle=True):
        self.g = self.o = None
        if gpu:
            import ray_tracing_amd as rta
            self.rta = rta
            self.g = rta.Scene(ambient)
        if oracle:
            import oracle_bind as ob
            self.ob = ob
            self.o = ob.OracleScene(ambient)

    def mat(self, name):
        return (self.rta.material(name) if self.g is not None else None, self.ob.material(name) if self.o is not None else None)

    def call(self, fn, *args, mat=None, colour=None):
        out = None
        for side, m in ((self.g, mat[0] if mat else None), (self.o, mat[1] if mat else None)):
            if side is None:
                continue
            a = list(args)
            if mat is not None:
                a += [m, colour]
            out = getattr(side, fn)(*a)
        return out


def build_icosahedron(material="plastic", light="point_area", gpu=True, oracle=True, bsp=True, leaf_refine=None,
                      extra_sphere=False):
    """C1: Layout3_Icosahedron (Layout.h:136-160) without textures.
    light: 'point_area' = AddAreaLight((-10,20,20),1,0,1) (deterministic), 'area10' = Layout.h:155,
           'point' / 'parallel' use the brute-force shadow rules of Light.h."""
    P = _Pair(0.5, gpu, oracle)
    plastic = P.mat("plastic")
    P.call("add_plane", (0, 0, 0), (0, 0, 1), mat=plastic, colour=(255, 255, 255))
    P.call("add_plane", (10, 0, 0), (-1, 0, 0), mat=plastic, colour=(0, 0, 0))
    P.call("read_obj_triangle", ICOSA_OBJ, (0, -1, 0), 0.0015, mat=P.mat(material), colour=(241, 46, 24))
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
    if P.g is not None and leaf_refine is not None:
        P.g.set_leaf_refine(leaf_refine)
    return P.g, P.o


C1_CAMERA = dict(view=(-10, 0, 1.5), eye=(1, 0, 0), left=(0, -1, 0), dist=10, hor=6.4, ver=4.8, res=100, depth=1)


def heightfield_triangles(n_cells=708, half=5.0):
    """C2 mesh: heightfield on [-half,half]^2, n_cells per side, 2 triangles per cell, order i-major,
    j-minor, (P(i,j),P(i+1,j),P(i+1,j+1)) then (P(i,j),P(i+1,j+1),P(i,j+1)).  Height = the surveyor's probe
    function z = 0.6 sin1.7x cos2.3y + 0.25 sin(5.1x+1.3) sin(4.3y+0.7) + 0.05 sin23x cos19y (host libm)."""
    idx = np.arange(n_cells + 1, dtype=np.float64)
    xs = -half + (2 * half) * idx / n_cells
    X, Y = np.meshgrid(xs, xs, indexing="ij")
    Z = 0.6 * np.sin(1.7 * X) * np.cos(2.3 * Y) + 0.25 * np.sin(5.1 * X + 1.3) * np.sin(4.3 * Y + 0.7) + 0.05 * np.sin(23 * X) * np.cos(19 * Y)
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


def build_heightfield(n_cells=708, gpu=True, oracle=True, leaf_refine=None, max_obj_num=0, light=True):
    """C2: ground plane z=-1 (id 0) + heightfield, plastic, point-like area light."""
    P = _Pair(0.5, gpu, oracle)
    plastic = P.mat("plastic")
    P.call("add_plane", (0, 0, -1), (0, 0, 1), mat=plastic, colour=(200, 200, 200))
    tris = heightfield_triangles(n_cells)
    P.call("add_triangles", tris, mat=plastic, colour=(90, 160, 60))
    P.call("construct_bsp", max_obj_num, 0)
    if light:
        P.call("add_area_light", (-10, 20, 20), 1.0, 0.0, 1)
    if P.g is not None and leaf_refine is not None:
        P.g.set_leaf_refine(leaf_refine)
    return P.g, P.o


def oracle_camera(cam_args):
    import oracle_bind as ob
    return ob.make_camera(cam_args["view"], cam_args["eye"], cam_args["left"], cam_args["dist"], cam_args["hor"],
                          cam_args["ver"], cam_args["res"], cam_args["depth"])


def oracle_render_whitted(orc_scene, cam_args, background=(0, 0, 0), seed=0, **kw):
    return orc_scene.render_whitted(oracle_camera(cam_args), background=background, seed=seed, **kw)


def fnv_hash_ids(ids):
    """the hash of SURVEY 8c over (id+1) in raster order"""
    h = 1469598103934665603
    M = (1 << 64) - 1
    for i in np.asarray(ids).tolist():
        h ^= (i + 1) & 0xFFFFFFFF
        h = (h * 1099511628211) & M
    return "%016x" % h
