"""Per-kernel traversal statistics on C2: primary rays alone (scene without light) vs primary+shadow."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ray_tracing_amd as rta
import scenes

refine = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ctx = rta.Context(0)
def run(with_light):
    P = scenes._Pair(0.5, True, False)
    pl = P.mat("plastic")
    P.call("add_plane", (0, 0, -1), (0, 0, 1), mat=pl, colour=(200, 200, 200))
    P.call("add_triangles", scenes.heightfield_triangles(708), mat=pl, colour=(90, 160, 60))
    P.call("construct_bsp", 0, 0)
    if with_light:
        P.call("add_area_light", (-10, 20, 20), 1.0, 0.0, 1)
    P.g.set_leaf_refine(refine)
    cam = rta.Camera(**scenes.C2_CAMERA)
    n = cam.width * cam.height
    rgb = np.zeros((n, 3)); ids = np.zeros(n, np.int32)
    p = rta.Params(background=(200, 200, 200), collect_stats=1)
    ctx.stats(True)
    rta.check(rta.hip.rt_hip_render_whitted(P.g.device(ctx), cam.raw, p, rgb.ctypes.data, ids.ctypes.data))
    return ctx.stats(True), ids
a, ids = run(False)
b, _ = run(True)
print("hit pixels: terrain %d plane %d miss %d" % ((ids > 0).sum(), (ids == 0).sum(), (ids < 0).sum()))
def show(name, s, rays):
    print(name, {k: round(v / rays, 2) for k, v in s.items() if k not in ("rays", "shadow_rays")}, "rays", rays)
show("primary", a, a["rays"])
d = {k: b[k] - a[k] for k in a}
show("shadow ", d, d["rays"])
