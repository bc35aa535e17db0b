"""Scene builders of the tests: the workloads of ray_tracing_amd/workloads.py built on BOTH sides -- the product scene
(ray_tracing_amd.Scene, the C++ host layer) and the checker's (tests/oracle_bind.OracleScene) -- with the same calls in
the same order, so that primitive ids agree.  Either side may be skipped (gpu=False / oracle=False)."""
import os

import numpy as np

from ray_tracing_amd import workloads as W
from ray_tracing_amd.workloads import (C1_CAMERA, C2_CAMERA, C3_CAMERA, C4_CAMERA, C5_CAMERA, ROOM_CAMERA, TEX_CAMERA, PAPER_CTRL, PAPER_CENTER,  # noqa: F401
                                       CUP2_PROFILES, APPLE_PROFILE, fnv_hash_ids, synth_texture, heightfield_triangles)

# the meshes of the reference's layouts are test data: kept under tests/golden/assets, handed to the workload builders from here
ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "assets")
W.set_asset_dir(ASSETS)
ICOSA_OBJ = os.path.join(ASSETS, "Icosahedron_Triangle.obj")     # the reference's ObjFiles/Icosahedron_Triangle.obj (a mesh: data)
WALL_LIGHT_OBJ = os.path.join(ASSETS, "WallLight.obj")           # the reference's ObjFiles/WallLight.obj, 11,576 faces (Layout.h:125)


def _stage(gpu, oracle):
    """stage factory for the workload builders; remembers which side is which"""
    def make(ambient):
        sides, tags = [], []
        if gpu:
            import ray_tracing_amd as rta
            sides.append((rta.Scene(ambient), rta.material)); tags.append("g")
        if oracle:
            import oracle_bind as ob
            sides.append((ob.OracleScene(ambient), ob.material)); tags.append("o")
        st = W.Stage(sides)
        st.tags = tags
        return st
    return make


class _Pair:
    """a two-sided stage for tests that add their own objects: .g / .o = product / oracle scene, .call / .mat as Stage"""

    def __init__(self, ambient, gpu=True, oracle=True):
        self._st = _stage(gpu, oracle)(ambient)
        self.g, self.o = _go(self._st)
        self.call, self.mat = self._st.call, self._st.mat


def _go(stage):
    d = {t: s for t, (s, _) in zip(stage.tags, stage.sides)}
    return d.get("g"), d.get("o")


def build_icosahedron(material="plastic", light="point_area", gpu=True, oracle=True, **kw):
    return _go(W.build_icosahedron(_stage(gpu, oracle), material=material, light=light, **kw))


def build_heightfield(n_cells=708, gpu=True, oracle=True, **kw):
    return _go(W.build_heightfield(_stage(gpu, oracle), n_cells, **kw))


def build_paper(gpu=True, oracle=True, **kw):
    return _go(W.build_paper(_stage(gpu, oracle), **kw))


def build_wine(gpu=True, oracle=True, **kw):
    return _go(W.build_wine(_stage(gpu, oracle), **kw))


def build_room(gpu=True, oracle=True, **kw):
    return _go(W.build_room(_stage(gpu, oracle), **kw))


def build_textured(gpu=True, oracle=True, **kw):
    return _go(W.build_textured(_stage(gpu, oracle), **kw))


def oracle_camera(cam_args):
    import oracle_bind as ob
    return ob.make_camera(cam_args["view"], cam_args["eye"], cam_args["left"], cam_args["dist"], cam_args["hor"],
                          cam_args["ver"], cam_args["res"], cam_args["depth"])


def oracle_render_whitted(orc_scene, cam_args, background=(0, 0, 0), seed=0, **kw):
    return orc_scene.render_whitted(oracle_camera(cam_args), background=background, seed=seed, **kw)
