"""Lock-step efficiency of the Newton solve loops (a -DRT_SOLVE_STATS build: tools/build_variant.sh solvestats -DRT_SOLVE_STATS, then
RT_HIP_LIB=../../variants/librt_hip_solvestats.so python tools/solve_probe.py): steps the waves ran x 64 against the steps their lanes needed,
on C4's camera + shadow rays (Bezier patch) and C5's eye pass (surfaces of revolution)."""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ray_tracing_amd as rta
from ray_tracing_amd import workloads as W
W.set_asset_dir(os.path.join(ROOT, "tests", "golden", "assets"))
ctx = rta.Context(0)
f = rta.hip.rt_hip_debug_read_solve_steps
f.argtypes = [C.c_void_p, C.c_int]
z = np.zeros(16, dtype=np.uint64)
out = {}


def read(tag):
    rta.check(f(z.ctypes.data, 1))
    out[tag] = {"patch_wave_steps_x64": int(z[0]), "patch_lane_steps": int(z[1]), "patch_efficiency": float(z[1]) / max(1.0, float(z[0])),
                "revolution_wave_steps_x64": int(z[2]), "revolution_lane_steps": int(z[3]), "revolution_efficiency": float(z[3]) / max(1.0, float(z[2])),
                "patch_pairs": int(z[4]), "patch_pairs_with_a_root": int(z[5]), "revolution_pairs": int(z[6]), "revolution_pairs_with_a_root": int(z[7]),
                "closest_hit_pairs": int(z[8]), "closest_hit_pairs_with_a_root": int(z[9]), "occlusion_pairs": int(z[10]), "occlusion_pairs_with_a_root": int(z[11]), "occlusion_pairs_settled_by_an_earlier_batch": int(z[12]),
                "patch_steps_per_pair": float(z[1]) / max(1.0, float(z[4])), "revolution_steps_per_pair": float(z[3]) / max(1.0, float(z[6]))}
    print(tag, json.dumps(out[tag]))


g = W.build_paper(W.product_stage).sides[0][0]
cam = rta.Camera(**W.C4_CAMERA)
rta.check(f(z.ctypes.data, 1))
cam.render_raytracer(ctx, g, background=(30, 30, 30))
read("C4 one sample per pixel")
g = W.build_wine(W.product_stage, cups="layout", light="area5").sides[0][0]
cam = rta.Camera(**W.C5_CAMERA)
cam.render_raytracer(ctx, g, background=(10, 10, 10))
read("C5 eye pass")
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
