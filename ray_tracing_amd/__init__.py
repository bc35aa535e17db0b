"""ray_tracing_amd -- Python harness over the MI355X ray-scene hot path.

The product is native: ``lib/librt_hip.so`` (hand-written gfx950 kernels behind the ``rt_hip_*``
C ABI of ``include/rt_hip.h``) and ``lib/librt_host.so`` (the C++ host layer that mirrors the
reference's CScene / CCamera interface).  This module only binds them with ctypes so that tests
and ``bench.py`` can drive them; it computes nothing and has NO fallback: importing it without
the built libraries raises, and every call without a GPU fails with RT_HIP_EHIP.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBDIR = os.path.join(_HERE, "lib")


def _load(name):
    path = os.path.join(_LIBDIR, name)
    if not os.path.exists(path):
        raise ImportError(
            "%s is missing: run `python __graft_entry__.py` (hipcc) first; there is no CPU fallback" % path)
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


class _Variant:
    """RT_HIP_LIB=<experiment build>: entry points the experiment build lacks (tools/build_variant.sh -DRT_MINIMAL) come from the full library"""

    def __init__(self, variant, full):
        self._v, self._f = variant, full

    def __getattr__(self, name):
        try:
            fn = getattr(self._v, name)
        except AttributeError:
            fn = getattr(self._f, name)
        setattr(self, name, fn)
        return fn


if os.environ.get("RT_HIP_LIB"):
    hip = _Variant(_load(os.environ["RT_HIP_LIB"]), _load("librt_hip.so"))
else:
    hip = _load("librt_hip.so")     # the C ABI (include/rt_hip.h)
host = _load("librt_host.so")   # C++ host layer, flat entry points (host/rt_host_capi.cpp)

d3 = C.c_double * 3
d8 = C.c_double * 8

# ---- ABI structs (include/rt_hip.h) -------------------------------------------------------------
RAY_DTYPE = np.dtype([("o", "f8", 3), ("d", "f8", 3)])
HIT_DTYPE = np.dtype([("t", "f8"), ("u", "f8"), ("v", "f8"), ("prim_id", "i4"), ("reserved", "i4")])
NODE_DTYPE = np.dtype([("box_min", "f8", 3), ("box_max", "f8", 3), ("split_axis", "i4"), ("a", "i4"), ("b", "i4"), ("reserved", "i4")])
TRI_DTYPE = np.dtype([("v", "f8", (3, 3)), ("vn", "f8", (3, 3)), ("has_vn", "i4"), ("reserved", "i4")])
assert RAY_DTYPE.itemsize == 48 and HIT_DTYPE.itemsize == 32 and NODE_DTYPE.itemsize == 64 and TRI_DTYPE.itemsize == 152


class CameraStruct(C.Structure):
    _fields_ = [("view_pt", d3), ("eye_dir", d3), ("left", d3), ("dist_of_plane", C.c_double), ("hor_range", C.c_double),
                ("ver_range", C.c_double), ("resolution", C.c_int32), ("depth", C.c_int32)]


def camera_array(cams):
    """the rt_hip_camera array that rt_hip_render_whitted_batch_dev takes, from Camera objects"""
    arr = (CameraStruct * len(cams))()
    for k, c in enumerate(cams):
        C.memmove(C.byref(arr[k]), c.raw, C.sizeof(CameraStruct))
    return arr


class Params(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("background", d3), ("weight_threshold", C.c_double), ("seed", C.c_uint64),
                ("spp", C.c_uint32), ("lens_radius", C.c_double), ("pixel_first", C.c_uint64), ("pixel_count", C.c_uint64),
                ("sample_first", C.c_uint32), ("sample_count", C.c_uint32), ("collect_stats", C.c_uint32), ("tile_offset", C.c_uint32),
                ("tile_stride", C.c_uint32), ("frame_format", C.c_uint32)]

    def __init__(self, background=(0, 0, 0), seed=0, **kw):
        super().__init__()
        self.struct_size = C.sizeof(Params)
        self.background = d3(*background)
        self.seed = seed
        for k, v in kw.items():
            setattr(self, k, v)


class AaParams(C.Structure):
    """rt_hip_aa_params: zero fields take the reference's constants (Camera.h:77-82)"""
    _fields_ = [("struct_size", C.c_uint32), ("oversample_times", C.c_uint32), ("r0", C.c_double),
                ("depth_grad_threshold", C.c_double), ("norm_grad_threshold", C.c_double), ("type_grad_threshold", C.c_double),
                ("colour_grad_threshold", C.c_double), ("focusing", C.c_uint32), ("reserved", C.c_uint32)]

    def __init__(self, **kw):
        super().__init__()
        self.struct_size = C.sizeof(AaParams)
        for k, v in kw.items():
            setattr(self, k, v)


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "shadow_rays", "node_visits", "prim_tests", "exact_tests",
                                          "wave_node_visits", "wave_prim_tests", "reserved")]


class Profile(C.Structure):
    _fields_ = [("ms", C.c_double * 12), ("launches", C.c_uint64 * 12)]


LEAF_REFINE_DEFAULT, LEAF_REFINE_NONE, LEAF_REFINE_OWN = 0, 0xFFFFFFFF, 0x40000000      # include/rt_hip.h
KERNEL_NAMES = ["k_primary", "k_trace_nodes", "k_shade", "k_combine", "k_resolve", "k_trace_closest", "k_trace_occluded", "k_shadow", "k_order", "k_aa", "k_frame_direct", "-"]

hip.rt_hip_last_error.restype = C.c_char_p
host.rth_last_error.restype = C.c_char_p
host.rth_context_raw.restype = C.c_void_p
host.rth_num_objects.restype = C.c_size_t
host.rth_tree_arrays.restype = C.c_size_t
host.rth_triangles.restype = C.c_size_t
host.rth_camera_rgb.restype = C.POINTER(C.c_double)
host.rth_camera_prim_id.restype = C.POINTER(C.c_int32)
host.rth_camera_image.restype = C.POINTER(C.c_uint8)
host.rth_camera_raw.restype = C.POINTER(CameraStruct)

# explicit prototypes where pointers / 64-bit values are passed
_vp = C.c_void_p
host.rth_context_new.argtypes = [C.c_int, C.POINTER(_vp)]
host.rth_context_free.argtypes = [_vp]
host.rth_context_raw.argtypes = [_vp]
host.rth_scene_new.argtypes = [C.c_double, C.POINTER(_vp)]
host.rth_scene_free.argtypes = [_vp]
host.rth_material_preset.argtypes = [C.c_int, d8]
host.rth_add_sphere.argtypes = [_vp, d3, C.c_double, d8, d3]
host.rth_add_plane.argtypes = [_vp, d3, d3, d8, d3]
host.rth_add_circle.argtypes = [_vp, d3, d3, C.c_double, d8, d3]
host.rth_add_bezier.argtypes = [_vp, _vp, d3, d8, d3]
host.rth_add_rotational.argtypes = [_vp, _vp, C.c_int, d3, C.c_int, C.c_double, d8, d3]
host.rth_add_triangle.argtypes = [_vp, _vp, _vp, d8, d3]
host.rth_add_triangles.argtypes = [_vp, _vp, C.c_size_t, d8, d3]
host.rth_add_triangle_tex.argtypes = [_vp, _vp, _vp, _vp, d8, d3, C.POINTER(C.c_int)]
host.rth_add_texture_image.argtypes = [_vp, C.c_uint, C.c_uint, _vp, C.POINTER(C.c_int)]
host.rth_read_ppm_texture.argtypes = [_vp, C.c_char_p, C.POINTER(C.c_int)]
host.rth_texture_info.argtypes = [_vp, C.c_int, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.POINTER(C.c_uint8))]
host.rth_obj_add_texture.argtypes = [_vp, C.c_int, C.c_int, C.c_double]
host.rth_obj_add_bump_texture.argtypes = [_vp, C.c_int, C.c_int, C.c_double]
host.rth_rotational_surface_to_mesh.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, d3, d8, d3, C.POINTER(C.c_int)]
host.rth_read_obj_triangle.argtypes = [_vp, C.c_char_p, d3, C.c_double, d8, d3, C.POINTER(C.c_int)]
host.rth_add_point_light.argtypes = [_vp, d3, C.c_double]
host.rth_add_parallel_light.argtypes = [_vp, d3, C.c_double]
host.rth_add_area_light.argtypes = [_vp, d3, C.c_double, C.c_double, C.c_int]
host.rth_construct_bsp.argtypes = [_vp, C.c_int, C.c_int]
host.rth_set_leaf_refine.argtypes = [_vp, C.c_uint]
host.rth_num_objects.argtypes = [_vp]
host.rth_tree_stats.argtypes = [_vp, C.POINTER(C.c_longlong), C.POINTER(C.c_double)]
host.rth_tree_arrays.argtypes = [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(C.c_size_t)]
host.rth_triangles.argtypes = [_vp, C.POINTER(_vp)]
host.rth_scene_device.argtypes = [_vp, _vp, C.POINTER(_vp)]
host.rth_first_inter_pt.argtypes = [_vp, _vp, _vp, C.c_size_t, C.c_int, C.c_int, _vp]
host.rth_camera_new.argtypes = [d3, d3, d3, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.POINTER(_vp)]
host.rth_camera_free.argtypes = [_vp]
host.rth_camera_size.argtypes = [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
host.rth_camera_seed.argtypes = [_vp, C.c_uint64]
host.rth_camera_raw.argtypes = [_vp]
host.rth_render_raytracer.argtypes = [_vp, _vp, _vp, d3]
host.rth_render_focusing.argtypes = [_vp, _vp, _vp, C.c_double, d3, C.c_uint]
host.rth_render_raytracer_antialiasing.argtypes = [_vp, _vp, _vp, d3, C.c_uint]
host.rth_render_focusing_antialiasing.argtypes = [_vp, _vp, _vp, C.c_double, d3, C.c_uint, C.c_uint]
host.rth_camera_edge_mask.argtypes = [_vp]
host.rth_render_photon_mapper.argtypes = [_vp, _vp, _vp, C.c_int, d3, C.c_uint, C.c_char_p, C.c_char_p]
host.rth_camera_edge_mask.restype = C.POINTER(C.c_uint8)
host.rth_camera_rgb.argtypes = [_vp]
host.rth_camera_prim_id.argtypes = [_vp]
host.rth_camera_image.argtypes = [_vp]
host.rth_camera_write_ppm.argtypes = [_vp, C.c_char_p]

hip.rt_hip_device_count.argtypes = [C.POINTER(C.c_int)]
hip.rt_hip_get_stats.argtypes = [_vp, C.POINTER(Stats), C.c_int]
hip.rt_hip_trace_closest.argtypes = [_vp, _vp, C.c_size_t, C.c_int, C.c_int, _vp]
hip.rt_hip_trace_closest_dev.argtypes = [_vp, _vp, C.c_size_t, C.c_int, C.c_int, _vp, C.c_uint32, _vp]
hip.rt_hip_trace_occluded.argtypes = [_vp, _vp, _vp, C.c_size_t, C.c_int, _vp]
hip.rt_hip_trace_occluded_dev.argtypes = [_vp, _vp, _vp, C.c_size_t, C.c_int, _vp, C.c_uint32, _vp]
hip.rt_hip_camera_size.argtypes = [C.POINTER(CameraStruct), C.POINTER(C.c_int), C.POINTER(C.c_int)]
hip.rt_hip_render_whitted.argtypes = [_vp, C.POINTER(CameraStruct), C.POINTER(Params), _vp, _vp]
hip.rt_hip_render_whitted_dev.argtypes = [_vp, C.POINTER(CameraStruct), C.POINTER(Params), _vp, _vp, _vp]
hip.rt_hip_render_whitted_noalloc.argtypes = [_vp, C.POINTER(CameraStruct), C.POINTER(Params), _vp, _vp, _vp]
hip.rt_hip_render_whitted_batch_dev.argtypes = [_vp, C.POINTER(CameraStruct), C.c_uint32, C.POINTER(Params), _vp, C.c_uint64, _vp, C.c_uint64, _vp]
hip.rt_hip_tile_shard_info.argtypes = [C.POINTER(CameraStruct), C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
hip.rt_hip_tiles_to_frame_dev.argtypes = [C.POINTER(CameraStruct), C.c_uint32, _vp, C.c_uint64, C.c_uint32, _vp, _vp]
hip.rt_hip_tiles_to_frames_dev.argtypes = [C.POINTER(CameraStruct), C.c_uint32, _vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, _vp, C.c_uint64, _vp]
hip.rt_hip_comm_unique_id.argtypes = [_vp]
hip.rt_hip_comm_create.argtypes = [_vp, _vp, C.c_int, C.c_int, C.POINTER(_vp)]
hip.rt_hip_comm_info.argtypes = [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
hip.rt_hip_comm_all_gather.argtypes = [_vp, _vp, _vp, C.c_size_t, _vp]
hip.rt_hip_comm_all_reduce_sum_f64.argtypes = [_vp, _vp, C.c_size_t, _vp]
hip.rt_hip_comm_destroy.argtypes = [_vp]
hip.rt_hip_render_focusing.argtypes = [_vp, C.POINTER(CameraStruct), C.POINTER(Params), _vp]
hip.rt_hip_render_focusing_dev.argtypes = [_vp, C.POINTER(CameraStruct), C.POINTER(Params), _vp, _vp]
hip.rt_hip_render_antialiasing.argtypes = [_vp, C.POINTER(CameraStruct), C.POINTER(Params), C.POINTER(AaParams), _vp, _vp]
hip.rt_hip_render_antialiasing_dev.argtypes = [_vp, C.POINTER(CameraStruct), C.POINTER(Params), C.POINTER(AaParams), _vp, _vp, _vp]
hip.rt_hip_last_aa_edges.argtypes = [_vp, C.POINTER(C.c_uint64)]
hip.rt_hip_ppm_state_size.argtypes = [_vp, C.POINTER(C.c_uint64)]
hip.rt_hip_ppm_save_state.argtypes = [_vp, _vp, C.c_uint64]
hip.rt_hip_ppm_load_state.argtypes = [_vp, _vp, C.c_uint64, C.POINTER(C.c_uint32)]
hip.rt_hip_godray_create.argtypes = [_vp, d3, C.c_uint64, C.POINTER(_vp)]
hip.rt_hip_godray_add_area_xz.argtypes = [_vp, C.c_double, C.c_double * 2, C.c_double * 2, C.c_double, d3]
hip.rt_hip_godray_build.argtypes = [_vp]
hip.rt_hip_godray_num_photons.argtypes = [_vp, C.POINTER(C.c_uint64)]
hip.rt_hip_godray_photons.argtypes = [_vp, _vp]
hip.rt_hip_render_godray.argtypes = [_vp, _vp, C.POINTER(CameraStruct), C.POINTER(Params), _vp, _vp]
hip.rt_hip_render_godray_dev.argtypes = [_vp, _vp, C.POINTER(CameraStruct), C.POINTER(Params), _vp, _vp, _vp]
hip.rt_hip_godray_destroy.argtypes = [_vp]
hip.rt_hip_ppm_begin_aa.argtypes = [_vp, C.POINTER(CameraStruct), C.POINTER(Params), C.POINTER(AaParams), C.c_uint32, C.c_uint32, C.POINTER(_vp)]
hip.rt_hip_ppm_edge_mask.argtypes = [_vp, _vp]
hip.rt_hip_render_godray_antialiasing.argtypes = [_vp, _vp, C.POINTER(CameraStruct), C.POINTER(Params), C.POINTER(AaParams), _vp, _vp]
hip.rt_hip_render_godray_antialiasing_dev.argtypes = [_vp, _vp, C.POINTER(CameraStruct), C.POINTER(Params), C.POINTER(AaParams), _vp, _vp, _vp]
hip.rt_hip_ppm_begin.argtypes = [_vp, C.POINTER(CameraStruct), C.POINTER(Params), C.c_uint32, C.c_uint32, C.POINTER(_vp)]
hip.rt_hip_ppm_info.argtypes = [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
hip.rt_hip_ppm_trace.argtypes = [_vp, C.c_uint32, C.c_uint64, C.c_uint64, _vp]
hip.rt_hip_ppm_accumulators.argtypes = [_vp, C.POINTER(_vp), C.POINTER(C.c_uint64)]
hip.rt_hip_ppm_update.argtypes = [_vp, _vp]
hip.rt_hip_ppm_pass.argtypes = [_vp, C.c_uint32]
hip.rt_hip_ppm_resolve.argtypes = [_vp, _vp]
hip.rt_hip_ppm_resolve_dev.argtypes = [_vp, _vp, _vp]
hip.rt_hip_ppm_hitpoints.argtypes = [_vp, _vp]
hip.rt_hip_ppm_end.argtypes = [_vp]
hip.rt_hip_scene_share.argtypes = [_vp, C.POINTER(_vp)]
hip.rt_hip_scene_destroy.argtypes = [_vp]
hip.rt_hip_set_profiling.argtypes = [_vp, C.c_int]
hip.rt_hip_get_profile.argtypes = [_vp, C.POINTER(Profile), C.c_int]
hip.rt_hip_last_ray_count.argtypes = [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
hip.rt_hip_last_proved_shadow_rays.argtypes = [_vp, C.POINTER(C.c_uint64)]
hip.rt_hip_last_wave_costs.argtypes = [_vp, _vp, C.c_uint64, C.POINTER(C.c_uint64)]


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("rt_hip error %d: %s" % (code, msg))
        self.code = code


def _chk_host(rc):
    if rc != 0:
        raise RtError(rc, host.rth_last_error().decode(errors="replace"))


def check(rc):
    """raise for a failed rt_hip_* call"""
    if rc != 0:
        raise RtError(rc, hip.rt_hip_last_error().decode(errors="replace"))


MATERIALS = {"mirror": 200, "glass": 201, "ice": 202, "metal": 203, "porcelain": 204, "rubber": 205, "plastic": 206}


def material(name):
    """CMaterial preset (Material.h:45-124) as 8 doubles"""
    m = d8()
    host.rth_material_preset(MATERIALS[name] if isinstance(name, str) else int(name), m)
    return m


def device_count():
    n = C.c_int(0)
    rc = hip.rt_hip_device_count(C.byref(n))
    return n.value if rc == 0 else 0


class Context:
    def __init__(self, device=0):
        self._h = _vp()
        _chk_host(host.rth_context_new(device, C.byref(self._h)))

    @property
    def raw(self):
        return host.rth_context_raw(self._h)

    def stats(self, reset=True):
        s = Stats()
        check(hip.rt_hip_get_stats(self.raw, C.byref(s), 1 if reset else 0))
        d = {n: getattr(s, n) for n, _ in Stats._fields_ if n != "reserved"}
        d["redo_rays"] = s.reserved & 0xFFFFFFFF     # rays re-walked with inline confirmation
        d["redo_waves"] = s.reserved >> 32
        return d

    def __del__(self):
        if getattr(self, "_h", None) and host is not None:
            host.rth_context_free(self._h)
            self._h = None


class GodRay:
    """CGodRay (GodRay.h:109-245) through rt_hip_godray_*"""

    def __init__(self, ctx, scene, light_colour, seed=0):
        self._scene, self._ctx, self.seed = scene, ctx, seed
        self.raw = _vp()
        check(hip.rt_hip_godray_create(scene.device(ctx), d3(*light_colour), seed, C.byref(self.raw)))

    def add_area_light_source_xz(self, y, mn, mx, interval, direction):
        check(hip.rt_hip_godray_add_area_xz(self.raw, y, (C.c_double * 2)(*mn), (C.c_double * 2)(*mx), interval, d3(*direction)))

    def photon_mapping_pass(self):
        check(hip.rt_hip_godray_build(self.raw))

    def photons(self):
        n = C.c_uint64()
        check(hip.rt_hip_godray_num_photons(self.raw, C.byref(n)))
        out = np.zeros((n.value, 4))
        if n.value:
            check(hip.rt_hip_godray_photons(self.raw, out.ctypes.data))
        return out

    def render(self, camera, background=(0, 0, 0)):
        """CCamera::Render_RayTracer_GodRay: (rgb (n,3) doubles, prim ids)"""
        n = camera.width * camera.height
        rgb = np.zeros((n, 3)); ids = np.zeros(n, np.int32)
        p = Params(background=background, seed=self.seed)        # area-light shadow samples
        check(hip.rt_hip_render_godray(self._scene.device(self._ctx), self.raw, camera.raw, p, rgb.ctypes.data, ids.ctypes.data))
        return rgb, ids

    def render_antialiasing(self, camera, aa=None, background=(0, 0, 0)):
        """CCamera::Render_RayTracer_GodRay_Antialiasing: (rgb (n,3) doubles, edge mask (n,) uint8)"""
        n = camera.width * camera.height
        rgb = np.zeros((n, 3)); mask = np.zeros(n, np.uint8)
        p = Params(background=background, seed=self.seed)
        check(hip.rt_hip_render_godray_antialiasing(self._scene.device(self._ctx), self.raw, camera.raw, p, aa if aa is not None else AaParams(),
                                                    rgb.ctypes.data, mask.ctypes.data))
        return rgb, mask

    def __del__(self):
        if getattr(self, "raw", None) and hip is not None:
            hip.rt_hip_godray_destroy(self.raw)
            self.raw = None


class PhotonMap:
    """CPhotonMapper driven as in CCamera::Render_PhotonMapper (Camera.h:291-313) through rt_hip_ppm_*"""

    def __init__(self, ctx, scene, camera, photon_num=50000, times=1, background=(0, 0, 0), seed=0, aa=None):
        """aa = AaParams: CCamera::Render_PhotonMapper_Antialiasing (Camera.h:315-423) through rt_hip_ppm_begin_aa"""
        self._scene, self._ctx, self._cam = scene, ctx, camera
        self.raw = _vp()
        p = Params(background=background, seed=seed)
        if aa is None:
            check(hip.rt_hip_ppm_begin(scene.device(ctx), camera.raw, p, photon_num, times, C.byref(self.raw)))
        else:
            check(hip.rt_hip_ppm_begin_aa(scene.device(ctx), camera.raw, p, aa, photon_num, times, C.byref(self.raw)))
        self.width, self.height = camera.width, camera.height

    def edge_mask(self):
        out = np.zeros(self.width * self.height, dtype=np.uint8)
        check(hip.rt_hip_ppm_edge_mask(self.raw, out.ctypes.data))
        return out

    def info(self):
        a, b, c, d = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(hip.rt_hip_ppm_info(self.raw, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return dict(hitpoints=a.value, photons_per_pass=b.value, photon_segments=c.value, photon_hits=d.value)

    def run_pass(self, index):
        check(hip.rt_hip_ppm_pass(self.raw, index))

    def trace(self, index, first=0, count=0, stream=None):
        check(hip.rt_hip_ppm_trace(self.raw, index, first, count, stream))

    def update(self, stream=None):
        check(hip.rt_hip_ppm_update(self.raw, stream))

    def accumulators(self):
        p, n = _vp(), C.c_uint64()
        check(hip.rt_hip_ppm_accumulators(self.raw, C.byref(p), C.byref(n)))
        return p.value, n.value

    def picture(self):
        out = np.zeros((self.width * self.height, 3))
        check(hip.rt_hip_ppm_resolve(self.raw, out.ctypes.data))
        return out

    def save_state(self):
        """checkpoint: bytes holding what the passes change (R, N per hit point, picture sums, census, next pass)"""
        n = C.c_uint64()
        check(hip.rt_hip_ppm_state_size(self.raw, C.byref(n)))
        buf = np.zeros(n.value, dtype=np.uint8)
        check(hip.rt_hip_ppm_save_state(self.raw, buf.ctypes.data, n.value))
        return buf.tobytes()

    def load_state(self, state):
        """resume: returns the index of the next pass; refuses a state of a different run"""
        buf = np.frombuffer(state, dtype=np.uint8)
        nxt = C.c_uint32()
        check(hip.rt_hip_ppm_load_state(self.raw, buf.ctypes.data, buf.size, C.byref(nxt)))
        return nxt.value

    def hitpoints(self):
        """(n, 11): pixel, path, pt[3], norm[3], wgt, R, N sorted by (pixel, path)"""
        n = self.info()["hitpoints"]
        out = np.zeros((n, 11))
        if n:
            check(hip.rt_hip_ppm_hitpoints(self.raw, out.ctypes.data))
            out = out[np.lexsort((out[:, 1], out[:, 0]))]
        return out

    def __del__(self):
        if getattr(self, "raw", None) and hip is not None:
            hip.rt_hip_ppm_end(self.raw)
            self.raw = None


class SharedScene:
    """rt_hip_scene_share handle; keeps its parent alive"""

    def __init__(self, parent, ctx):
        self._parent, self._ctx = parent, ctx
        self.raw = _vp()
        check(hip.rt_hip_scene_share(parent.device(ctx), C.byref(self.raw)))

    def __del__(self):
        if getattr(self, "raw", None) and hip is not None:
            hip.rt_hip_scene_destroy(self.raw)
            self.raw = None


def load_image(path):
    """cv::imread stand-in for the Python host: any file PIL decodes -> (rows, cols, 3) uint8 in B,G,R order.
    (JPEG decoders differ in the last bit; results are pinned on images given as arrays.)"""
    from PIL import Image
    rgb = np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8)
    return np.ascontiguousarray(rgb[:, :, ::-1])


class Scene:
    """rt::host::Scene (the CScene mirror)"""

    def __init__(self, ambient=0.0):
        self._h = _vp()
        _chk_host(host.rth_scene_new(float(ambient), C.byref(self._h)))

    def add_sphere(self, c, r, mat, colour):
        return host.rth_add_sphere(self._h, d3(*c), r, mat, d3(*colour))

    def add_plane(self, p0, n, mat, colour):
        return host.rth_add_plane(self._h, d3(*p0), d3(*n), mat, d3(*colour))

    def add_circle(self, p0, n, r, mat, colour):
        return host.rth_add_circle(self._h, d3(*p0), d3(*n), r, mat, d3(*colour))

    def add_bezier(self, ctrl, center, mat, colour):
        """CBezierObject: ctrl[4][4][3] control points + center"""
        c = np.ascontiguousarray(ctrl, dtype=np.float64).reshape(48)
        _chk_host(host.rth_add_bezier(self._h, c.ctypes.data, d3(*center), mat, d3(*colour)))
        return self.num_objects - 1

    def add_rotational(self, profile, center, u_sample_num, u_range, mat, colour):
        """CRotationalObject: profile = Bezier control points (radius, height)"""
        pr = np.ascontiguousarray(profile, dtype=np.float64).reshape(-1, 2)
        _chk_host(host.rth_add_rotational(self._h, pr.ctypes.data, pr.shape[0], d3(*center), int(u_sample_num), float(u_range), mat, d3(*colour)))
        return self.num_objects - 1

    def add_triangle(self, v, vn, mat, colour):
        v = np.ascontiguousarray(v, dtype=np.float64).reshape(9)
        vnp = None
        if vn is not None:
            vn = np.ascontiguousarray(vn, dtype=np.float64).reshape(9)
            vnp = vn.ctypes.data
        return host.rth_add_triangle(self._h, v.ctypes.data, vnp, mat, d3(*colour))

    def add_triangles(self, v, mat, colour):
        v = np.ascontiguousarray(v, dtype=np.float64).reshape(-1, 9)
        _chk_host(host.rth_add_triangles(self._h, v.ctypes.data, v.shape[0], mat, d3(*colour)))

    def add_triangle_tex(self, v, vn, tex_coord, mat, colour):
        """CTriangle with textureCoord (Object.h:386-400): three (row, col) pairs in texel units"""
        v = np.ascontiguousarray(v, dtype=np.float64).reshape(9)
        tc = np.ascontiguousarray(tex_coord, dtype=np.float64).reshape(6)
        vnp = None
        if vn is not None:
            vn = np.ascontiguousarray(vn, dtype=np.float64).reshape(9)
            vnp = vn.ctypes.data
        out = C.c_int(-1)
        _chk_host(host.rth_add_triangle_tex(self._h, v.ctypes.data, vnp, tc.ctypes.data, mat, d3(*colour), C.byref(out)))
        return out.value

    def add_texture_image(self, bgr):
        """registers a (rows, cols, 3) uint8 B,G,R image (cv::Mat CV_8UC3 of cv::imread); returns its index"""
        a = np.ascontiguousarray(bgr, dtype=np.uint8)
        assert a.ndim == 3 and a.shape[2] == 3
        out = C.c_int(-1)
        _chk_host(host.rth_add_texture_image(self._h, a.shape[0], a.shape[1], a.ctypes.data, C.byref(out)))
        return out.value

    def read_ppm_texture(self, path):
        out = C.c_int(-1)
        _chk_host(host.rth_read_ppm_texture(self._h, os.fsencode(path), C.byref(out)))
        return out.value

    def texture_image(self, tex):
        rows, cols, p = C.c_uint(), C.c_uint(), C.POINTER(C.c_uint8)()
        _chk_host(host.rth_texture_info(self._h, tex, C.byref(rows), C.byref(cols), C.byref(p)))
        return np.ctypeslib.as_array(p, shape=(rows.value, cols.value, 3)).copy()

    def add_texture(self, obj, tex, scale=1.0):
        """CBaseObject::AddTexture / CInftyPlane::AddTexture(file, textureScale)"""
        _chk_host(host.rth_obj_add_texture(self._h, obj, tex, float(scale)))

    def add_bump_texture(self, obj, tex, scale=1.0):
        """CBaseObject::AddBumpTexture / CInftyPlane::AddBumpTexture(file, bumpTextureScale)"""
        _chk_host(host.rth_obj_add_bump_texture(self._h, obj, tex, float(scale)))

    def rotational_surface_to_mesh(self, profile, u_num, v_num, center, mat, colour, tex=-1):
        """CScene::RotationalSurfaceToMesh / RotationalSurfaceToMesh_Texture (Scene.h:168-266); returns the triangles added"""
        pr = np.ascontiguousarray(profile, dtype=np.float64).reshape(-1, 2)
        n = C.c_int(0)
        _chk_host(host.rth_rotational_surface_to_mesh(self._h, pr.ctypes.data, pr.shape[0], int(u_num), int(v_num), int(tex), d3(*center), mat,
                                                     d3(*colour), C.byref(n)))
        return n.value

    def read_obj_triangle(self, path, center, scale, mat, colour):
        n = C.c_int(0)
        _chk_host(host.rth_read_obj_triangle(self._h, os.fsencode(path), d3(*center), scale, mat, d3(*colour), C.byref(n)))
        return n.value

    def add_point_light(self, p, lum):
        host.rth_add_point_light(self._h, d3(*p), lum)

    def add_parallel_light(self, d, lum):
        host.rth_add_parallel_light(self._h, d3(*d), lum)

    def add_area_light(self, p, lum, length, sample_num):
        host.rth_add_area_light(self._h, d3(*p), lum, length, sample_num)

    def construct_bsp(self, max_obj_num=0, max_depth=0):
        _chk_host(host.rth_construct_bsp(self._h, max_obj_num, max_depth))

    def set_leaf_refine(self, n):
        """which hierarchy the device walks (include/rt_hip.h): LEAF_REFINE_DEFAULT = the library's own, LEAF_REFINE_NONE =
        the reference's tree as built, n = that tree with leaves split down to n objects, LEAF_REFINE_OWN | n = own, leaf size n"""
        host.rth_set_leaf_refine(self._h, n)

    @property
    def num_objects(self):
        return host.rth_num_objects(self._h)

    def tree_stats(self):
        st = (C.c_longlong * 8)()
        box = (C.c_double * 6)()
        host.rth_tree_stats(self._h, st, box)
        return list(st), list(box)

    def tree_arrays(self):
        nodes, refs, nr = _vp(), _vp(), C.c_size_t(0)
        nn = host.rth_tree_arrays(self._h, C.byref(nodes), C.byref(refs), C.byref(nr))
        if nn == 0:
            return np.zeros(0, NODE_DTYPE), np.zeros(0, np.uint32)
        n = np.ctypeslib.as_array(C.cast(nodes, C.POINTER(C.c_uint8)), shape=(nn * 64,)).view(NODE_DTYPE).copy()
        r = np.ctypeslib.as_array(C.cast(refs, C.POINTER(C.c_uint32)), shape=(nr.value,)).copy() if nr.value else np.zeros(0, np.uint32)
        return n, r

    def triangles(self):
        t = _vp()
        n = host.rth_triangles(self._h, C.byref(t))
        if n == 0:
            return np.zeros(0, TRI_DTYPE)
        return np.ctypeslib.as_array(C.cast(t, C.POINTER(C.c_uint8)), shape=(n * 152,)).view(TRI_DTYPE).copy()

    def device(self, ctx):
        """rt_hip_scene* for direct ABI calls (uploads on first use)"""
        raw = _vp()
        _chk_host(host.rth_scene_device(self._h, ctx._h, C.byref(raw)))
        return raw

    def share(self, ctx):
        """another rt_hip_scene* on the same geometry with its own render workspace (a frame in flight)"""
        return SharedScene(self, ctx)

    def first_inter_pt(self, ctx, rays, use_tree=True, record=True):
        """CScene::FirstInterPt_BSPTree / FirstInterPt for an (n,6) array of rays"""
        r = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        hits = np.zeros(r.shape[0], HIT_DTYPE)
        _chk_host(host.rth_first_inter_pt(self._h, ctx._h, r.ctypes.data, r.shape[0], int(use_tree), int(record), hits.ctypes.data))
        return hits

    def occluded(self, ctx, rays, ref_pts=None, use_tree=True):
        r = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        out = np.zeros(r.shape[0], np.uint8)
        rp = None
        if ref_pts is not None:
            ref_pts = np.ascontiguousarray(ref_pts, dtype=np.float64).reshape(-1, 3)
            assert ref_pts.shape[0] == r.shape[0]
            rp = ref_pts.ctypes.data
        check(hip.rt_hip_trace_occluded(self.device(ctx), r.ctypes.data, rp, r.shape[0], int(use_tree), out.ctypes.data))
        return out

    def __del__(self):
        if getattr(self, "_h", None) and host is not None:
            host.rth_scene_free(self._h)
            self._h = None


class Camera:
    """rt::host::Camera (the CCamera mirror)"""

    def __init__(self, view, eye, left, dist, hor, ver, res, depth):
        self._h = _vp()
        _chk_host(host.rth_camera_new(d3(*view), d3(*eye), d3(*left), dist, hor, ver, res, depth, C.byref(self._h)))
        w, h = C.c_int(), C.c_int()
        host.rth_camera_size(self._h, C.byref(w), C.byref(h))
        self.width, self.height = w.value, h.value

    @property
    def raw(self):
        return host.rth_camera_raw(self._h)

    def set_seed(self, seed):
        host.rth_camera_seed(self._h, seed)

    def render_raytracer(self, ctx, scene, background=(0, 0, 0)):
        _chk_host(host.rth_render_raytracer(self._h, ctx._h, scene._h, d3(*background)))

    def render_focusing(self, ctx, scene, radius, background=(0, 0, 0), spp=100):
        _chk_host(host.rth_render_focusing(self._h, ctx._h, scene._h, radius, d3(*background), spp))

    def render_raytracer_antialiasing(self, ctx, scene, background=(0, 0, 0), oversample_times=100):
        """CCamera::Render_RayTracer_Antialiasing, Camera.h:83-169"""
        _chk_host(host.rth_render_raytracer_antialiasing(self._h, ctx._h, scene._h, d3(*background), oversample_times))

    def render_focusing_antialiasing(self, ctx, scene, radius, background=(0, 0, 0), spp=100, oversample_times=100):
        """CCamera::Render_Focusing_Antialiasing, Camera.h:201-280"""
        _chk_host(host.rth_render_focusing_antialiasing(self._h, ctx._h, scene._h, radius, d3(*background), spp, oversample_times))

    def render_photon_mapper(self, ctx, scene, times, background=(0, 0, 0), photon_num=50000, checkpoint=None, progress=None):
        """CCamera::Render_PhotonMapper, Camera.h:291-313; checkpoint / progress: optional files (state after every pass /
        the picture after every pass as binary PPM)"""
        _chk_host(host.rth_render_photon_mapper(self._h, ctx._h, scene._h, times, d3(*background), photon_num,
                                                os.fsencode(checkpoint) if checkpoint else None, os.fsencode(progress) if progress else None))

    def edge_mask(self):
        p = host.rth_camera_edge_mask(self._h)
        return np.ctypeslib.as_array(p, shape=(self.width * self.height,)).copy()

    def rgb(self):
        n = self.width * self.height
        return np.ctypeslib.as_array(host.rth_camera_rgb(self._h), shape=(n, 3)).copy()

    def prim_id(self):
        p = host.rth_camera_prim_id(self._h)
        return np.ctypeslib.as_array(p, shape=(self.width * self.height,)).copy()

    def image(self):
        return np.ctypeslib.as_array(host.rth_camera_image(self._h), shape=(self.height, self.width, 3)).copy()

    def write_ppm(self, path):
        _chk_host(host.rth_camera_write_ppm(self._h, os.fsencode(path)))

    def __del__(self):
        if getattr(self, "_h", None) and host is not None:
            host.rth_camera_free(self._h)
            self._h = None
