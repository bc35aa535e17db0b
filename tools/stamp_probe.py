"""Diagnostic: per-wave start/end wall-clock stamps of the stats build of k_shadow on C2."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ray_tracing_amd as rta
import scenes
ctx = rta.Context(0)
g, _ = scenes.build_heightfield(708, gpu=True, oracle=False, leaf_refine=32)
cam = rta.Camera(**scenes.C2_CAMERA)
n = cam.width * cam.height
rgb = np.zeros((n, 3)); ids = np.zeros(n, np.int32)
p = rta.Params(background=(200, 200, 200), collect_stats=0)
for _ in range(2):
    rta.check(rta.hip.rt_hip_render_whitted(g.device(ctx), cam.raw, p, rgb.ctypes.data, ids.ctypes.data))
nw = 32640
st = np.zeros((nw, 2), np.uint64)
rta.hip.rt_hip_debug_read_stamps.argtypes = [C.c_void_p, C.c_size_t]
rta.check(rta.hip.rt_hip_debug_read_stamps(st.ctypes.data, nw))
t0 = st[:, 0].min(); s = (st[:, 0] - t0).astype(np.float64) / 100.0; e = (st[:, 1] - t0).astype(np.float64) / 100.0   # us (100 MHz)
dur = e - s
print("kernel span %.0f us; wave dur us: mean %.1f median %.1f p90 %.1f p99 %.1f max %.1f" % (e.max(), dur.mean(), np.median(dur), np.percentile(dur, 90), np.percentile(dur, 99), dur.max()))
print("sum dur / span = avg resident waves %.0f (of %d SIMDs)" % (dur.sum() / e.max(), 1024))
for q in (50, 75, 90, 95, 99, 100):
    print("  %3d%% of waves ended by %.0f us" % (q, np.percentile(e, q)))
# resident waves over time
ts = np.linspace(0, e.max(), 21)
print("resident:", [int(((s <= t) & (e > t)).sum()) for t in ts])
print("last start %.0f us" % s.max())
# which waves are the slowest?  relate to the spread of primary hit depths inside the tile
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_bind as ob
ocam = scenes.oracle_camera(scenes.C2_CAMERA)
rays, W, H = ob.OracleScene(0.0).camera_rays(ocam)
hits = g.first_inter_pt(ctx, rays)
tmap = hits["t"].reshape(H, W); idm = hits["prim_id"].reshape(H, W)
tiles_x, tiles_y = (W + 7) // 8, (H + 7) // 8
sx, sy = (tiles_x + 15) // 16, (tiles_y + 7) // 8
w = np.arange(nw); st_i, in_st = w // 128, w % 128
tx = (st_i % sx) * 16 + in_st % 16; ty = (st_i // sx) * 8 + in_st // 16
ok = (tx < tiles_x) & (ty < tiles_y)
spread = np.zeros(nw); terr = np.zeros(nw)
for k in np.nonzero(ok)[0]:
    blk_t = tmap[ty[k] * 8:ty[k] * 8 + 8, tx[k] * 8:tx[k] * 8 + 8]; blk_i = idm[ty[k] * 8:ty[k] * 8 + 8, tx[k] * 8:tx[k] * 8 + 8]
    spread[k] = blk_t.max() - blk_t.min(); terr[k] = (blk_i > 0).mean()
order = np.argsort(-dur)
print("slowest waves: dur us, tile, depth spread, terrain frac")
for k in order[:12]:
    print("  %.0f us tile (%d,%d) spread %.3f terr %.2f" % (dur[k], tx[k], ty[k], spread[k], terr[k]))
hv = terr > 0
print("corr(dur, spread) over terrain tiles: %.2f" % np.corrcoef(dur[hv], spread[hv])[0, 1])
for lo, hi in ((0, 0.05), (0.05, 0.2), (0.2, 0.5), (0.5, 1), (1, 2), (2, 100)):
    m = hv & (spread >= lo) & (spread < hi)
    if m.any(): print("  spread [%.2f,%.2f): %d waves, mean dur %.0f us, total %.0f ms" % (lo, hi, m.sum(), dur[m].mean(), dur[m].sum() / 1e3))
