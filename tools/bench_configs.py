"""Timings of the other BASELINE configs on one GPU; C2 is bench.py.  Prints one JSON line per config.
Sizes that would run for minutes are cut (fewer samples / photons) and the cut is stated in the line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import numpy as np
import torch
import ray_tracing_amd as rta
import scenes

ctx = rta.Context(0)
dev = torch.device("cuda", 0)
FULL = "FULL" in (a.upper() for a in sys.argv[1:])   # the configs at BASELINE's full sample / photon counts (minutes)
ONLY = set(a.upper() for a in sys.argv[1:]) - {"FULL"}          # e.g. "C5" "C4": run only those


def want(name):
    return not ONLY or name in ONLY


def ray_count(raw):
    a, b = C.c_uint64(), C.c_uint64()
    rta.check(rta.hip.rt_hip_last_ray_count(raw, C.byref(a), C.byref(b)))
    return a.value, b.value


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


# ---- C1: icosahedron 640x480, metal, depth 3 ----
if want("C1"):
    g, _ = scenes.build_icosahedron(material="metal", light="point_area", oracle=False)
    cam = rta.Camera(**dict(scenes.C1_CAMERA, depth=3))
    n = cam.width * cam.height
    rgb = torch.zeros(n * 3, dtype=torch.float64, device=dev)
    p = rta.Params(background=(200, 200, 200))
    raw = g.device(ctx)
    dt = timed(lambda: rta.check(rta.hip.rt_hip_render_whitted_dev(raw, cam.raw, p, rgb.data_ptr(), None, None)), 20)
    rays, sh = ray_count(raw)
    print(json.dumps({"config": "C1 icosahedron 640x480 metal depth 3", "s_per_frame": round(dt, 6), "rays": rays, "Mrays_per_s": round(rays / dt / 1e6, 1)}), flush=True)

# ---- C3: depth of field, 1080p, area light 10x10 (100 shadow rays per camera sample) ----
if want("C3"):
    g, _ = scenes.build_icosahedron(material="plastic", light="area10", oracle=False)
    cam = rta.Camera(view=(-10, 0, 1.5), eye=(1, 0, 0), left=(0, -1, 0), dist=10, hor=6, ver=3.375, res=320, depth=1)
    n = cam.width * cam.height
    rgb = torch.zeros(n * 3, dtype=torch.float64, device=dev)
    spp_run = 4
    p = rta.Params(background=(200, 200, 200), spp=1024, lens_radius=0.1, sample_first=0, sample_count=spp_run, seed=1)
    raw = g.device(ctx)
    dt = timed(lambda: rta.check(rta.hip.rt_hip_render_focusing_dev(raw, cam.raw, p, rgb.data_ptr(), None)), 2)
    rays, sh = ray_count(raw)
    print(json.dumps({"config": "C3 depth of field %dx%d, area light 10x10" % (cam.width, cam.height), "samples_timed": spp_run, "s_per_sample_pass": round(dt / spp_run, 5),
                      "s_per_frame_at_1024spp_extrapolated": round(dt / spp_run * 1024, 1), "rays": rays, "Mrays_per_s": round(rays / dt / 1e6, 1)}), flush=True)
    if FULL:
        p = rta.Params(background=(200, 200, 200), spp=1024, lens_radius=0.1, seed=1)
        t = time.perf_counter()
        rta.check(rta.hip.rt_hip_render_focusing_dev(raw, cam.raw, p, rgb.data_ptr(), None)); torch.cuda.synchronize()
        dt = time.perf_counter() - t
        rays, sh = ray_count(raw)
        print(json.dumps({"config": "C3 FULL depth of field %dx%d @1024 spp, area light 10x10" % (cam.width, cam.height), "s_per_frame": round(dt, 2), "rays": rays,
                          "Mrays_per_s": round(rays / dt / 1e6, 1)}), flush=True)

# ---- C4: wall + Bezier patch + mesh, 1080p, depth 2: one sample per pixel, then the jittered fixed-spp render (every pixel re-sampled) ----
if want("C4"):
    g, _ = scenes.build_paper(oracle=False)
    cam = rta.Camera(**scenes.C4_CAMERA)
    n = cam.width * cam.height
    rgb = torch.zeros(n * 3, dtype=torch.float64, device=dev)
    p = rta.Params(background=(30, 30, 30))
    raw = g.device(ctx)
    dt = timed(lambda: rta.check(rta.hip.rt_hip_render_whitted_dev(raw, cam.raw, p, rgb.data_ptr(), None, None)), 2)
    rays, sh = ray_count(raw)
    print(json.dumps({"config": "C4 Bezier patch scene %dx%d depth 2, 1 spp" % (cam.width, cam.height), "s_per_frame": round(dt, 4), "rays": rays,
                      "Mrays_per_s": round(rays / dt / 1e6, 1)}), flush=True)
    spp_run = 16
    aa = rta.AaParams(oversample_times=spp_run, depth_grad_threshold=-1.0)
    dt = timed(lambda: rta.check(rta.hip.rt_hip_render_antialiasing_dev(raw, cam.raw, p, aa, rgb.data_ptr(), None, None)), 1)
    rays, sh = ray_count(raw)
    print(json.dumps({"config": "C4 Bezier patch scene %dx%d depth 2, pixel-jittered, every pixel re-sampled" % (cam.width, cam.height), "spp_timed": spp_run,
                      "s_per_frame": round(dt, 4), "rays": rays, "Mrays_per_s": round(rays / dt / 1e6, 1),
                      "s_per_frame_at_256spp_extrapolated": round(dt * 256 / spp_run, 2)}), flush=True)
    if FULL:
        aa = rta.AaParams(oversample_times=256, depth_grad_threshold=-1.0)
        t = time.perf_counter()
        rta.check(rta.hip.rt_hip_render_antialiasing_dev(raw, cam.raw, p, aa, rgb.data_ptr(), None, None)); torch.cuda.synchronize()
        dt = time.perf_counter() - t
        rays, sh = ray_count(raw)
        print(json.dumps({"config": "C4 FULL Bezier patch scene %dx%d depth 2 @256 spp pixel-jittered" % (cam.width, cam.height), "s_per_frame": round(dt, 2), "rays": rays,
                          "Mrays_per_s": round(rays / dt / 1e6, 1)}), flush=True)

# ---- C5: wine scene (three cups, no tree), 1080p depth 8, photon mapping ----
if want("C5"):
    g, _ = scenes.build_wine(oracle=False, cups=((-2, -2, 0), (2.2, 2.2, 0), (5, 5, 0)), light="area")
    cam = rta.Camera(**scenes.C5_CAMERA)
    t = time.perf_counter()
    photon_num = 40000          # x 4 light samples = 160,000 photons per pass here (BASELINE: 10 M per pass)
    pm = rta.PhotonMap(ctx, g, cam, photon_num=photon_num, times=64, background=(10, 10, 10), seed=1)
    torch.cuda.synchronize()
    t_eye = time.perf_counter() - t
    info = pm.info()
    t = time.perf_counter()
    pm.run_pass(0)
    t_pass = time.perf_counter() - t
    i2 = pm.info()
    print(json.dumps({"config": "C5 wine scene %dx%d depth 8, photon mapping" % (cam.width, cam.height), "eye_pass_s": round(t_eye, 3), "hit_points": info["hitpoints"],
                      "photons_in_pass": info["photons_per_pass"], "s_per_pass": round(t_pass, 3), "photon_segments": i2["photon_segments"], "photon_hits": i2["photon_hits"],
                      "Mphoton_segments_per_s": round(i2["photon_segments"] / t_pass / 1e6, 2),
                      "s_per_pass_at_10M_photons_extrapolated": round(t_pass * 1e7 / info["photons_per_pass"], 1)}), flush=True)
    if FULL:
        del pm
        g, _ = scenes.build_wine(oracle=False, cups=((-2, -2, 0), (2.2, 2.2, 0), (5, 5, 0)), light="area5")
        t = time.perf_counter()
        pm = rta.PhotonMap(ctx, g, cam, photon_num=400000, times=64, background=(10, 10, 10), seed=1)       # x 25 light samples = 10 M photons per pass
        torch.cuda.synchronize()
        t_eye = time.perf_counter() - t
        info = pm.info()
        t0 = time.perf_counter()
        for k in range(64):
            t = time.perf_counter()
            pm.run_pass(k)
            print("  pass %d: %.2f s" % (k, time.perf_counter() - t), flush=True)
        t_all = time.perf_counter() - t0
        i2 = pm.info()
        pic = pm.picture()
        print(json.dumps({"config": "C5 FULL wine scene %dx%d depth 8, 64 passes x 10 M photons" % (cam.width, cam.height), "eye_pass_s": round(t_eye, 2),
                          "hit_points": info["hitpoints"], "photons_per_pass": info["photons_per_pass"], "s_all_passes": round(t_all, 1), "s_per_pass": round(t_all / 64, 2),
                          "photon_segments": i2["photon_segments"], "photon_hits": i2["photon_hits"], "Mphoton_segments_per_s": round(i2["photon_segments"] / t_all / 1e6, 1),
                          "picture_mean": round(float(pic.mean()), 2)}), flush=True)
