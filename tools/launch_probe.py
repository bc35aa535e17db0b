"""Diagnostic: host time per depth-1 frame call (enqueue only) for a small tile share of C2, and GPU time of the same sequence."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ray_tracing_amd as rta
from ray_tracing_amd import workloads as W
ctx = rta.Context(0)
g = W.build_heightfield(W.product_stage, 708).sides[0][0]
cam = rta.Camera(**W.C2_CAMERA)
raw = g.device(ctx)
stride = int(sys.argv[1]) if len(sys.argv) > 1 else 32
tot, per = C.c_uint32(), C.c_uint32()
rta.check(rta.hip.rt_hip_tile_shard_info(cam.raw, stride, C.byref(tot), C.byref(per)))
out = torch.zeros(per.value * 8192 * 3, dtype=torch.uint8, device="cuda")
p = rta.Params(background=(200, 200, 200), frame_format=1, tile_offset=0, tile_stride=stride)
st = torch.cuda.current_stream()
for _ in range(20):
    rta.check(rta.hip.rt_hip_render_whitted_dev(raw, cam.raw, p, C.c_void_p(out.data_ptr()), None, C.c_void_p(st.cuda_stream)))
torch.cuda.synchronize()
N = 400
t = time.perf_counter()
for _ in range(N):
    rta.check(rta.hip.rt_hip_render_whitted_dev(raw, cam.raw, p, C.c_void_p(out.data_ptr()), None, C.c_void_p(st.cuda_stream)))
t_host = time.perf_counter() - t
torch.cuda.synchronize()
t_all = time.perf_counter() - t
print("share 1/%d: host enqueue %.1f us per frame, until done %.1f us per frame" % (stride, t_host / N * 1e6, t_all / N * 1e6))
