"""Diagnostic: the photon map of the wine scene with the in-place and the deferred Newton solves, what differs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ray_tracing_amd as rta
import scenes
ctx = rta.Context(0)
g, _ = scenes.build_wine(cups="layout", light="area5", oracle=False)
c = dict(scenes.C5_CAMERA); c["res"] = 24; c["depth"] = 8
cam = rta.Camera(**c)
out = []
for defer in ("0", "1"):
    os.environ["RT_HIP_PARAM_DEFER"] = defer
    pm = rta.PhotonMap(ctx, g, cam, photon_num=1500, times=2, background=(10, 10, 10), seed=5)
    pm.run_pass(0)
    out.append((pm.hitpoints(), pm.picture(), pm.info()))
    del pm
a, b = out[0][0], out[1][0]
print("info", out[0][2], out[1][2])
print("shape", a.shape, b.shape)
for col, name in enumerate(["pixel", "path", "px", "py", "pz", "nx", "ny", "nz", "wgt", "R", "N"]):
    d = a[:, col] != b[:, col]
    print(name, int(d.sum()), float(np.abs(a[:, col] - b[:, col]).max()))
print("picture diff", int((out[0][1] != out[1][1]).sum()))
