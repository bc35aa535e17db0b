import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import ray_tracing_amd as rta, scenes
ctx = rta.Context(0)
g, _ = scenes.build_icosahedron(material="plastic", light="point_area", extra_sphere=True, oracle=False)
cam = rta.Camera(**dict(scenes.C1_CAMERA, hor=3.2, ver=2.4, res=20, depth=2))
hipm = C.cdll.LoadLibrary("libamdhip64.so")
def acc_of(m):
    p, n = m.accumulators(); h = np.zeros(n)
    torch.cuda.synchronize()
    rc = hipm.hipMemcpy(C.c_void_p(h.ctypes.data), C.c_void_p(p), C.c_size_t(n * 8), 2)
    assert rc == 0, rc
    return h.reshape(-1, 4)
def raw_hp(m):
    n = m.info()["hitpoints"]; out = np.zeros((n, 11))
    rta.check(rta.hip.rt_hip_ppm_hitpoints(m.raw, out.ctypes.data)); return out
def run(shards):
    m = rta.PhotonMap(ctx, g, cam, photon_num=10000, times=1, seed=3)
    hp = raw_hp(m)
    for f, c in shards: m.trace(0, f, c)
    a = acc_of(m)
    return a, hp, m
a1, h1, m1 = run([(0, 0)])
a2, h2, m2 = run([(0, 0)])
print("hp equal", (h1 == h2).all(), "sorted", (np.diff(h1[:, 0]) >= 0).all())
print("acc diff", int((a1[:, 0] != a2[:, 0]).sum()), a1[:5, 0], a2[:5, 0])
a1b = acc_of(m1)
print("re-read same", (a1b == a1).all())
m1.trace(0); a1c = acc_of(m1)
print("second trace doubles:", np.abs(a1c[:, 0] - 2 * a1[:, 0]).max())
