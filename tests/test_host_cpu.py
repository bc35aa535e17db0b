"""Host layer (C++ Scene mirror) on CPU: OBJ parsing + BSP build/flatten against the oracle's
pointer tree, and that both product libraries load and export every symbol include/rt_hip.h declares."""
import ctypes as C
import os
import re

import numpy as np

import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol():
    import ray_tracing_amd as rta
    hdr = open(os.path.join(ROOT, "include", "rt_hip.h")).read()
    names = sorted(set(re.findall(r"\b(rt_hip_[a-z_0-9]+)\s*\(", hdr)))
    assert len(names) >= 14
    for n in names:
        assert hasattr(rta.hip, n), n


def test_struct_sizes_match_header():
    import ray_tracing_amd as rta
    assert C.sizeof(rta.CameraStruct) == 104
    assert C.sizeof(rta.Params) == 104
    assert C.sizeof(rta.Stats) == 64


def test_no_gpu_means_loud_failure():
    import ray_tracing_amd as rta
    if rta.device_count() > 0:
        return
    try:
        rta.Context(0)
    except rta.RtError as e:
        assert e.code == -2
    else:
        raise AssertionError("Context() must fail without a GPU")


def test_obj_parse_and_flattened_tree_equal_oracle():
    g, o = scenes.build_icosahedron()
    assert g.num_objects == o.num_objects == 1371
    gs, gbox = g.tree_stats()
    os_, obox = o.tree_stats()
    assert gs == os_
    assert gbox == obox                      # bit-identical root box
    tris = g.triangles()
    v = np.zeros(9)
    vn = np.zeros(9)
    import oracle_bind as ob
    for k in (0, 1, 700, 1368):
        has = ob.L.orc_get_triangle(o.h, k + 2, v.ctypes.data, vn.ctypes.data)
        assert has == 1 and tris[k]["has_vn"] == 1
        assert (tris[k]["v"].reshape(9) == v).all() and (tris[k]["vn"].reshape(9) == vn).all()


def test_flattened_tree_structure():
    g, _ = scenes.build_icosahedron(oracle=False)
    nodes, refs = g.tree_arrays()
    assert len(nodes) == 99 and len(refs) == 3565
    inner = nodes[nodes["split_axis"] >= 0]
    assert (inner["a"] > 0).all() and (inner["b"] > inner["a"]).all()
    # axis cycles X->Y->Z with depth (BSPTree.h:69-92) and children halve the parent box
    for i, n in enumerate(nodes):
        if n["split_axis"] < 0:
            continue
        l, r = nodes[n["a"]], nodes[n["b"]]
        ax = n["split_axis"]
        mid = (n["box_min"][ax] + n["box_max"][ax]) / 2
        assert l["box_max"][ax] == mid and r["box_min"][ax] == mid
        for c in (l, r):
            if c["split_axis"] >= 0:
                assert c["split_axis"] == (ax + 1) % 3


def test_small_scene_root_is_leaf():
    import ray_tracing_amd as rta
    s = rta.Scene(0.1)
    m = rta.material("plastic")
    s.add_plane((0, 0, 0), (0, 0, 1), m, (1, 2, 3))
    s.add_sphere((0, 0, 1), 1.0, m, (1, 2, 3))
    s.construct_bsp()
    st, _ = s.tree_stats()
    assert st[0] == 1 and st[1] == 1 and st[3] == 2 and st[7] == 1
    s2 = rta.Scene(0.1)
    s2.add_plane((0, 0, 0), (0, 0, 1), m, (1, 2, 3))
    s2.construct_bsp()                       # only infinite objects: no tree (Scene.h:289-290)
    assert s2.tree_stats()[0][7] == 0
