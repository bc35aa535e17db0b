"""Diagnostic (library built with -DRT_TIMERS, passed in RT_HIP_LIB): shader cycles per section of the walk on C2."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ray_tracing_amd as rta
import scenes
ctx = rta.Context(0)
g, _ = scenes.build_heightfield(708, gpu=True, oracle=False)
cam = rta.Camera(**scenes.C2_CAMERA)
n = cam.width * cam.height
rgb = np.zeros((n, 3)); ids = np.zeros(n, np.int32)
p = rta.Params(background=(200, 200, 200), collect_stats=0)
rta.hip.rt_hip_debug_read_timers.argtypes = [C.c_void_p, C.c_int]
tm = np.zeros(16, np.uint64)
rta.check(rta.hip.rt_hip_render_whitted(g.device(ctx), cam.raw, p, rgb.ctypes.data, ids.ctypes.data))
rta.check(rta.hip.rt_hip_debug_read_timers(tm.ctypes.data, 1))
rta.check(rta.hip.rt_hip_render_whitted(g.device(ctx), cam.raw, p, rgb.ctypes.data, ids.ctypes.data))
rta.check(rta.hip.rt_hip_debug_read_timers(tm.ctypes.data, 1))
names = ["pop+ballot", "inner(pair fetch, 2 slabs, push)", "stage wait", "leaf header slab", "record loop", "#inner steps", "#chunks staged", "#records"]
for which, off in (("closest (k_primary)", 0), ("occluded (k_shadow)", 8)):
    t = tm[off:off + 8].astype(np.float64)
    tot = t[:5].sum()
    print(which, "total Mcycles %.1f" % (tot / 1e6))
    for k in range(5):
        print("  %-34s %5.1f %%" % (names[k], 100 * t[k] / tot))
    print("  inner steps %d (%.0f cyc each incl. pop), chunks %d (stage %.0f cyc each, header %.0f), records %d (%.0f cyc each)" % (
        t[5], (t[1] + t[0]) / max(1, t[5] + t[6]), t[6], t[2] / max(1, t[6]), t[3] / max(1, t[6]), t[7], t[4] / max(1, t[7])))
