"""Diagnostic: the C5 eye pass (rt_hip_ppm_begin at 1920x1080, depth 8) twice in one process: the first call pays the workspace allocations."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ray_tracing_amd as rta
import scenes
ctx = rta.Context(0)
g, _ = scenes.build_wine(cups="layout", light="area5", oracle=False)
cam = rta.Camera(**scenes.C5_CAMERA)
for k in range(3):
    t = time.perf_counter()
    pm = rta.PhotonMap(ctx, g, cam, photon_num=400000, times=64, background=(10, 10, 10), seed=1)
    dt = time.perf_counter() - t
    t = time.perf_counter(); pm.run_pass(0); dp = time.perf_counter() - t
    t = time.perf_counter(); pm.run_pass(1); dp2 = time.perf_counter() - t
    print("eye pass %d: %.3f s; photon pass 0: %.3f s, pass 1: %.3f s" % (k, dt, dp, dp2), pm.info())
    del pm
