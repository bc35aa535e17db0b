"""Diagnostic: distribution of the per-wave run times of a C2 frame (rt_hip_last_wave_costs), one frame in flight.
How much of the launch is a tail, and how unequal the tiles are."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import ray_tracing_amd as rta
from ray_tracing_amd import workloads as W
ctx = rta.Context(0)
g = W.build_heightfield(W.product_stage, 708).sides[0][0]
cam = rta.Camera(**W.C2_CAMERA)
raw = g.device(ctx)
n = cam.width * cam.height
out = torch.zeros(n * 3, dtype=torch.uint8, device="cuda")
p = rta.Params(background=(200, 200, 200), frame_format=1)
for _ in range(3):
    rta.check(rta.hip.rt_hip_render_whitted_dev(raw, cam.raw, p, C.c_void_p(out.data_ptr()), None, None))
torch.cuda.synchronize()
nw = C.c_uint64()
rta.check(rta.hip.rt_hip_last_wave_costs(raw, None, 0, C.byref(nw)))
cost = np.zeros(nw.value, dtype=np.uint32)
rta.check(rta.hip.rt_hip_last_wave_costs(raw, C.c_void_p(cost.ctypes.data), nw.value, C.byref(nw)))
c = np.sort(cost.astype(np.float64))[::-1]
tot = c.sum()
slots = 256 * 16
print("waves %d, total %.3g cycles, mean %.0f, max %.0f (%.0f us at 2.4 GHz), median %.0f" % (len(c), tot, c.mean(), c[0], c[0] / 2400, np.median(c)))
print("ideal time with %d slots: total/slots = %.0f cycles = %.0f us; the longest wave alone: %.0f us" % (slots, tot / slots, tot / slots / 2400, c[0] / 2400))
for q in (0.001, 0.01, 0.05, 0.1, 0.25, 0.5):
    k = max(1, int(len(c) * q))
    print("  the longest %5.1f %% of the waves hold %5.1f %% of the time (shortest of them %.0f us)" % (100 * q, 100 * c[:k].sum() / tot, c[k - 1] / 2400))
