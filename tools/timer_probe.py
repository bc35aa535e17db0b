"""Diagnostic (library built with -DRT_TIMERS, passed in RT_HIP_LIB): shader cycles per section of the walk on C2, and the trip counts of
its loops (node-loop iterations, inner steps, pushes, chunks staged, chunks culled, cull steps, events, hand-back rounds) per wave-walk --
the dynamic side of the instruction itemisation in profiles/README.md (tools/isa_items.py has the static side)."""
import ctypes as C, json, os, sys
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
tm = np.zeros(32, np.uint64)
rta.check(rta.hip.rt_hip_render_whitted(g.device(ctx), cam.raw, p, rgb.ctypes.data, ids.ctypes.data))
rta.check(rta.hip.rt_hip_debug_read_timers(tm.ctypes.data, 1))
rta.check(rta.hip.rt_hip_render_whitted(g.device(ctx), cam.raw, p, rgb.ctypes.data, ids.ctypes.data))
rta.check(rta.hip.rt_hip_debug_read_timers(tm.ctypes.data, 1))
names = ["pop+ballot", "inner(pair fetch, 2 slabs, push)", "stage wait", "leaf header slab", "record loop"]
cnames = {5: "inner_steps", 6: "chunks_staged", 7: "records_culled", 8: "walks", 9: "node_loop_iterations", 10: "chunks_culled", 11: "cull_steps",
          12: "handback_rounds", 13: "interested_rays_per_culled_chunk_sum", 14: "pushes", 15: "events"}
out = {}
for which, off in (("closest", 0), ("occluded", 16)):
    t = tm[off:off + 16].astype(np.float64)
    tot = t[:5].sum()
    print(which, "total Mcycles %.1f" % (tot / 1e6))
    for k in range(5):
        print("  %-34s %5.1f %%" % (names[k], 100 * t[k] / tot))
    d = {cnames[k]: int(t[k]) for k in cnames}
    d["cycles_pct"] = {names[k]: round(100 * t[k] / tot, 1) for k in range(5)}
    out[which] = d
    w = max(1.0, t[8])
    print("  per wave-walk (%d walks):" % t[8], {cnames[k]: round(t[k] / w, 2) for k in cnames if k != 8})
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
