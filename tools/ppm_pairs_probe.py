"""Diagnostic (library built with -DRT_SOLVE_STATS, passed in RT_HIP_LIB): (ray, object) pairs, roots and Newton steps of ONE C5 photon pass (10 M photons)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ray_tracing_amd as rta
import scenes
ctx = rta.Context(0)
f = rta.hip.rt_hip_debug_read_solve_steps
f.argtypes = [C.c_void_p, C.c_int]
z = np.zeros(16, dtype=np.uint64)
g, _ = scenes.build_wine(cups="layout", light="area5", oracle=False)
cam = rta.Camera(**scenes.C5_CAMERA)
pm = rta.PhotonMap(ctx, g, cam, photon_num=400000, times=64, background=(10, 10, 10), seed=1)
rta.check(f(z.ctypes.data, 1))
pm.run_pass(0)
rta.check(f(z.ctypes.data, 1))
info = pm.info()
print({"photon_segments": info["photon_segments"], "closest_hit_pairs": int(z[8]), "with_a_root": int(z[9]), "lane_steps": int(z[3]), "steps_per_pair": float(z[3]) / max(1.0, float(z[8])),
       "pairs_per_segment": float(z[8]) / max(1, info["photon_segments"])})
