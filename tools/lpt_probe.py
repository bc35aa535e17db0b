"""Diagnostic (library built with -DRT_STAMPS): per-wave durations of k_primary and k_shadow on C2, and what a
longest-first order would buy (list scheduling on 4096 wave slots)."""
import ctypes as C, os, sys, heapq
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
for _ in range(2):
    rta.check(rta.hip.rt_hip_render_whitted(g.device(ctx), cam.raw, p, rgb.ctypes.data, ids.ctypes.data))
nw = 32640
rta.hip.rt_hip_debug_read_stamps2.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
dur = {}
for name, k in (("shadow", 0), ("primary", 1)):
    st = np.zeros((nw, 2), np.uint64)
    rta.check(rta.hip.rt_hip_debug_read_stamps2(st.ctypes.data, nw, k))
    ok = st[:, 0] > 0
    t0 = st[ok, 0].min()
    s = (st[:, 0] - t0).astype(np.float64) / 100.0; e = (st[:, 1] - t0).astype(np.float64) / 100.0
    d = np.where(ok, e - s, 0.0)
    dur[name] = d
    print("%s: span %.0f us, mean dur %.1f, max %.0f, sum/4096 = %.0f us" % (name, e[ok].max(), d[ok].mean(), d.max(), d.sum() / 4096))

def simulate(cost, order, slots=4096):
    h = [0.0] * slots
    heapq.heapify(h)
    end = 0.0
    for w in order:
        t = heapq.heappop(h) + cost[w]
        end = max(end, t)
        heapq.heappush(h, t)
    return end

print("corr(primary, shadow) = %.3f" % np.corrcoef(dur["primary"], dur["shadow"])[0, 1])
for name in ("primary", "shadow"):
    c = dur[name]
    heavy = c > 30
    cur = np.concatenate([np.nonzero(heavy)[0], np.nonzero(~heavy)[0]])
    print(name, "simulated span: heavy-first (today) %.0f us, exact LPT %.0f us" % (simulate(c, cur), simulate(c, np.argsort(-c))))
c = dur["shadow"]
print("shadow ordered by primary duration: %.0f us" % simulate(c, np.argsort(-dur["primary"])))
# the contended durations overstate what a wave needs alone; scale-free check with noise
for noise in (0.25, 0.5):
    rng = np.random.default_rng(0)
    pred = c * np.exp(rng.normal(0, noise, c.shape))
    print("shadow ordered by a predictor with %.0f%% log-noise: %.0f us" % (100 * noise, simulate(c, np.argsort(-pred))))
