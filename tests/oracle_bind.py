"""ctypes binding of oracle/librt_oracle.so -- the CPU checker (test infrastructure only)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_path = os.path.join(ROOT, "oracle", "librt_oracle.so")
if not os.path.exists(_path):
    raise ImportError("oracle/librt_oracle.so missing: run `make -C oracle` or `python __graft_entry__.py`")
L = C.CDLL(_path)

d3 = C.c_double * 3
d8 = C.c_double * 8
_vp = C.c_void_p


class OrcCamera(C.Structure):
    _fields_ = [("view_pt", d3), ("eye_dir", d3), ("left_arg", d3), ("dist", C.c_double), ("hor", C.c_double),
                ("ver", C.c_double), ("res", C.c_int32), ("depth", C.c_int32)]


L.orc_scene_new.restype = _vp
L.orc_scene_new.argtypes = [C.c_double]
L.orc_scene_free.argtypes = [_vp]
L.orc_material_preset.argtypes = [C.c_int, d8]
L.orc_add_sphere.argtypes = [_vp, d3, C.c_double, d8, d3]
L.orc_add_plane.argtypes = [_vp, d3, d3, d8, d3]
L.orc_add_circle.argtypes = [_vp, d3, d3, C.c_double, d8, d3]
L.orc_add_triangle.argtypes = [_vp, _vp, _vp, d8, d3]
L.orc_add_texture_image.argtypes = [_vp, C.c_int, C.c_int, _vp]
L.orc_obj_add_texture.argtypes = [_vp, C.c_int, C.c_int, C.c_double]
L.orc_obj_add_bump_texture.argtypes = [_vp, C.c_int, C.c_int, C.c_double]
L.orc_triangle_set_tex_coord.argtypes = [_vp, C.c_int, _vp]
L.orc_rotational_surface_to_mesh.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, d3, d8, d3]
L.orc_add_bezier.argtypes = [_vp, _vp, d3, d8, d3]
L.orc_add_rotational.argtypes = [_vp, _vp, C.c_int, d3, C.c_int, C.c_double, d8, d3]
L.orc_read_obj_triangle.argtypes = [_vp, C.c_char_p, d3, C.c_double, d8, d3]
L.orc_num_objects.argtypes = [_vp]
L.orc_get_triangle.argtypes = [_vp, C.c_int, _vp, _vp]
L.orc_add_point_light.argtypes = [_vp, d3, C.c_double]
L.orc_add_parallel_light.argtypes = [_vp, d3, C.c_double]
L.orc_add_area_light.argtypes = [_vp, d3, C.c_double, C.c_double, C.c_int]
L.orc_construct_bsp.argtypes = [_vp, C.c_int, C.c_int]
L.orc_tree_stats.argtypes = [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_double)]
L.orc_rng_u32.restype = C.c_uint32
L.orc_rng_u32.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
L.orc_trace_closest.argtypes = [_vp, _vp, C.c_size_t, C.c_int, C.c_int, _vp, _vp, _vp]
L.orc_camera_size.argtypes = [C.POINTER(OrcCamera), C.POINTER(C.c_int), C.POINTER(C.c_int)]
L.orc_camera_rays.argtypes = [C.POINTER(OrcCamera), _vp]
L.orc_render_whitted.argtypes = [_vp, C.POINTER(OrcCamera), d3, C.c_uint64, _vp, _vp, C.c_size_t, C.c_size_t, C.c_int, _vp]
L.orc_render_antialiasing.argtypes = [_vp, C.POINTER(OrcCamera), C.c_double * 6, C.c_int, C.c_double, C.c_int, d3, C.c_uint64, _vp, _vp, C.c_int]
L.orc_render_focusing.argtypes = [_vp, C.POINTER(OrcCamera), C.c_double, C.c_int, d3, C.c_uint64, _vp, C.c_size_t, C.c_size_t, C.c_int, _vp]
L.orc_godray_new.argtypes = [_vp, d3, C.c_uint64]
L.orc_godray_new.restype = _vp
L.orc_godray_add_area_xz.argtypes = [_vp, C.c_double, C.c_double * 2, C.c_double * 2, C.c_double, d3]
L.orc_godray_build.argtypes = [_vp]
L.orc_godray_num_photons.argtypes = [_vp]
L.orc_godray_num_photons.restype = C.c_size_t
L.orc_godray_photons.argtypes = [_vp, _vp]
L.orc_render_godray.argtypes = [_vp, C.POINTER(OrcCamera), d3, _vp, _vp]
L.orc_godray_free.argtypes = [_vp]
L.orc_ppm_begin.argtypes = [_vp, C.POINTER(OrcCamera), d3, C.c_uint64, C.c_int, C.c_int]
L.orc_ppm_begin.restype = _vp
L.orc_ppm_begin_aa.argtypes = [_vp, C.POINTER(OrcCamera), C.c_double * 6, d3, C.c_uint64, C.c_int, C.c_int]
L.orc_ppm_begin_aa.restype = _vp
L.orc_ppm_edge_mask.argtypes = [_vp, _vp]
L.orc_render_godray_aa.argtypes = [_vp, C.POINTER(OrcCamera), C.c_double * 6, d3, _vp, _vp]
L.orc_ppm_pass.argtypes = [_vp, C.c_int]
L.orc_ppm_num_hitpoints.argtypes = [_vp]
L.orc_ppm_num_hitpoints.restype = C.c_size_t
L.orc_ppm_hitpoints.argtypes = [_vp, _vp]
L.orc_ppm_picture.argtypes = [_vp, _vp]
L.orc_ppm_counters.argtypes = [_vp, _vp]
L.orc_ppm_end.argtypes = [_vp]
L.orc_object_first_inter_pt.argtypes = [_vp, C.c_int, _vp, _vp, _vp]

MATERIALS = {"mirror": 200, "glass": 201, "ice": 202, "metal": 203, "porcelain": 204, "rubber": 205, "plastic": 206}


def material(name):
    m = d8()
    L.orc_material_preset(MATERIALS[name], m)
    return m


def make_camera(view, eye, left, dist, hor, ver, res, depth):
    return OrcCamera(d3(*view), d3(*eye), d3(*left), dist, hor, ver, res, depth)


class OracleScene:
    def __init__(self, ambient=0.0):
        self.h = L.orc_scene_new(float(ambient))

    def add_sphere(self, c, r, mat, colour):
        return L.orc_add_sphere(self.h, d3(*c), r, mat, d3(*colour))

    def add_plane(self, p0, n, mat, colour):
        return L.orc_add_plane(self.h, d3(*p0), d3(*n), mat, d3(*colour))

    def add_circle(self, p0, n, r, mat, colour):
        return L.orc_add_circle(self.h, d3(*p0), d3(*n), r, mat, d3(*colour))

    def add_bezier(self, ctrl, center, mat, colour):
        c = np.ascontiguousarray(ctrl, dtype=np.float64).reshape(48)
        return L.orc_add_bezier(self.h, c.ctypes.data, d3(*center), mat, d3(*colour))

    def add_rotational(self, profile, center, u_sample_num, u_range, mat, colour):
        pr = np.ascontiguousarray(profile, dtype=np.float64).reshape(-1, 2)
        return L.orc_add_rotational(self.h, pr.ctypes.data, pr.shape[0], d3(*center), int(u_sample_num), float(u_range), mat, d3(*colour))

    def add_triangle(self, v, vn, mat, colour):
        v = np.ascontiguousarray(v, dtype=np.float64).reshape(9)
        vnp = None
        if vn is not None:
            vn = np.ascontiguousarray(vn, dtype=np.float64).reshape(9)
            vnp = vn.ctypes.data
        return L.orc_add_triangle(self.h, v.ctypes.data, vnp, mat, d3(*colour))

    def add_triangle_tex(self, v, vn, tex_coord, mat, colour):
        i = self.add_triangle(v, vn, mat, colour)
        tc = np.ascontiguousarray(tex_coord, dtype=np.float64).reshape(6)
        assert L.orc_triangle_set_tex_coord(self.h, i, tc.ctypes.data) == 0
        return i

    def add_texture_image(self, bgr):
        a = np.ascontiguousarray(bgr, dtype=np.uint8)
        t = L.orc_add_texture_image(self.h, a.shape[0], a.shape[1], a.ctypes.data)
        assert t >= 0
        return t

    def add_texture(self, obj, tex, scale=1.0):
        assert L.orc_obj_add_texture(self.h, obj, tex, float(scale)) == 0

    def add_bump_texture(self, obj, tex, scale=1.0):
        assert L.orc_obj_add_bump_texture(self.h, obj, tex, float(scale)) == 0

    def rotational_surface_to_mesh(self, profile, u_num, v_num, center, mat, colour, tex=-1):
        pr = np.ascontiguousarray(profile, dtype=np.float64).reshape(-1, 2)
        n = L.orc_rotational_surface_to_mesh(self.h, pr.ctypes.data, pr.shape[0], int(u_num), int(v_num), int(tex), d3(*center), mat, d3(*colour))
        assert n >= 0
        return n

    def add_triangles(self, v, mat, colour):
        v = np.ascontiguousarray(v, dtype=np.float64).reshape(-1, 9)
        col = d3(*colour)
        for k in range(v.shape[0]):
            L.orc_add_triangle(self.h, v[k].ctypes.data, None, mat, col)

    def read_obj_triangle(self, path, center, scale, mat, colour):
        n = L.orc_read_obj_triangle(self.h, os.fsencode(path), d3(*center), scale, mat, d3(*colour))
        if n < 0:
            raise IOError(path)
        return n

    def add_point_light(self, p, lum):
        L.orc_add_point_light(self.h, d3(*p), lum)

    def add_parallel_light(self, d, lum):
        L.orc_add_parallel_light(self.h, d3(*d), lum)

    def add_area_light(self, p, lum, length, sample_num):
        L.orc_add_area_light(self.h, d3(*p), lum, length, sample_num)

    def construct_bsp(self, max_obj_num=0, max_depth=0):
        L.orc_construct_bsp(self.h, max_obj_num, max_depth)

    @property
    def num_objects(self):
        return L.orc_num_objects(self.h)

    def get_triangle(self, i):
        """(v[9], vn[9]) of object i, or None when it is not a triangle"""
        v, vn = np.zeros(9), np.zeros(9)
        if L.orc_get_triangle(self.h, i, v.ctypes.data, vn.ctypes.data) < 0:
            return None
        return v, vn

    def tree_stats(self):
        st = (C.c_int64 * 8)()
        box = (C.c_double * 6)()
        L.orc_tree_stats(self.h, st, box)
        return list(st), list(box)

    def trace_closest(self, rays, use_bsp=True, record=True, want_uv=False):
        r = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        n = r.shape[0]
        ids = np.zeros(n, np.int32)
        pts = np.zeros((n, 3))
        uv = np.zeros((n, 2))
        L.orc_trace_closest(self.h, r.ctypes.data, n, int(use_bsp), int(record), ids.ctypes.data, pts.ctypes.data,
                            uv.ctypes.data if want_uv else None)
        return (ids, pts, uv) if want_uv else (ids, pts)

    def object_first_inter_pt(self, obj_id, ray):
        """CBaseObject::FirstInterPt of one object: (point, (u, v)); point = (1e8,)*3 for a miss"""
        r = np.ascontiguousarray(ray, dtype=np.float64).reshape(6)
        pt = np.zeros(3); uv = np.zeros(2)
        L.orc_object_first_inter_pt(self.h, int(obj_id), r.ctypes.data, pt.ctypes.data, uv.ctypes.data)
        return pt, uv

    def render_whitted(self, cam, background=(0, 0, 0), seed=0, first=0, count=0, threads=0, counters=False):
        w, h = C.c_int(), C.c_int()
        L.orc_camera_size(C.byref(cam), C.byref(w), C.byref(h))
        n = count if count else w.value * h.value
        rgb = np.zeros((n, 3))
        ids = np.zeros(n, np.int32)
        cnt = np.zeros(4, np.int64)
        L.orc_render_whitted(self.h, C.byref(cam), d3(*background), seed, rgb.ctypes.data, ids.ctypes.data, first, count,
                             threads, cnt.ctypes.data if counters else None)
        return (rgb, ids, cnt) if counters else (rgb, ids)

    def render_focusing(self, cam, radius, spp, background=(0, 0, 0), seed=0, first=0, count=0, threads=0):
        w, h = C.c_int(), C.c_int()
        L.orc_camera_size(C.byref(cam), C.byref(w), C.byref(h))
        n = count if count else w.value * h.value
        rgb = np.zeros((n, 3))
        L.orc_render_focusing(self.h, C.byref(cam), radius, spp, d3(*background), seed, rgb.ctypes.data, first, count, threads, None)
        return rgb

    def render_antialiasing(self, cam, focusing=False, radius=0.0, spp=100, oversample_times=100, r0=0.007,
                            thresholds=(0.8, 0.8, 0.5, 80.0), background=(0, 0, 0), seed=0, threads=0):
        """Camera.h:83-169 / 201-280; returns (rgb (n,3), edge mask (n,) uint8)"""
        w, h = C.c_int(), C.c_int()
        L.orc_camera_size(C.byref(cam), C.byref(w), C.byref(h))
        n = w.value * h.value
        rgb = np.zeros((n, 3)); mask = np.zeros(n, dtype=np.uint8)
        aa = (C.c_double * 6)(float(oversample_times), r0, *thresholds)
        L.orc_render_antialiasing(self.h, C.byref(cam), aa, int(bool(focusing)), radius, spp, d3(*background), seed,
                                  rgb.ctypes.data, mask.ctypes.data, threads)
        return rgb, mask

    def camera_rays(self, cam):
        w, h = C.c_int(), C.c_int()
        L.orc_camera_size(C.byref(cam), C.byref(w), C.byref(h))
        rays = np.zeros((w.value * h.value, 6))
        L.orc_camera_rays(C.byref(cam), rays.ctypes.data)
        return rays, w.value, h.value

    def __del__(self):
        if getattr(self, "h", None):
            L.orc_scene_free(self.h)
            self.h = None


class OraclePhotonMap:
    """orc_ppm_*: Camera.h:291-313 + PhotonMapper.h on the CPU"""

    def __init__(self, scene, cam, photon_num=50000, times=1, background=(0, 0, 0), seed=0, aa=None):
        """aa = dict(oversample_times, r0, thresholds): Render_PhotonMapper_Antialiasing (Camera.h:315-423)"""
        self._scene = scene
        w, h = C.c_int(), C.c_int()
        L.orc_camera_size(C.byref(cam), C.byref(w), C.byref(h))
        self.width, self.height = w.value, h.value
        if aa is None:
            self.h = L.orc_ppm_begin(scene.h, C.byref(cam), d3(*background), seed, photon_num, times)
        else:
            a6 = (C.c_double * 6)(float(aa.get("oversample_times", 100)), aa.get("r0", 0.007), *aa.get("thresholds", (0.8, 0.8, 0.5, 80.0)))
            self.h = L.orc_ppm_begin_aa(scene.h, C.byref(cam), a6, d3(*background), seed, photon_num, times)

    def edge_mask(self):
        out = np.zeros(self.width * self.height, dtype=np.uint8)
        L.orc_ppm_edge_mask(self.h, out.ctypes.data)
        return out

    def run_pass(self, index):
        L.orc_ppm_pass(self.h, index)

    def hitpoints(self):
        n = L.orc_ppm_num_hitpoints(self.h)
        out = np.zeros((n, 14))
        if n:
            L.orc_ppm_hitpoints(self.h, out.ctypes.data)
        return out

    def picture(self):
        out = np.zeros((self.width * self.height, 3))
        L.orc_ppm_picture(self.h, out.ctypes.data)
        return out

    def counters(self):
        c = np.zeros(2, np.int64)
        L.orc_ppm_counters(self.h, c.ctypes.data)
        return dict(photon_segments=int(c[0]), photon_hits=int(c[1]))

    def __del__(self):
        if getattr(self, "h", None):
            L.orc_ppm_end(self.h)
            self.h = None


class OracleGodRay:
    """orc_godray_*: GodRay.h + Camera.h:425-448 on the CPU"""

    def __init__(self, scene, light_colour, seed=0):
        self._scene = scene
        self.h = L.orc_godray_new(scene.h, d3(*light_colour), seed)

    def add_area_light_source_xz(self, y, mn, mx, interval, direction):
        L.orc_godray_add_area_xz(self.h, y, (C.c_double * 2)(*mn), (C.c_double * 2)(*mx), interval, d3(*direction))

    def photon_mapping_pass(self):
        L.orc_godray_build(self.h)

    def photons(self):
        n = L.orc_godray_num_photons(self.h)
        out = np.zeros((n, 4))
        if n:
            L.orc_godray_photons(self.h, out.ctypes.data)
        return out

    def render(self, cam, background=(0, 0, 0)):
        w, h = C.c_int(), C.c_int()
        L.orc_camera_size(C.byref(cam), C.byref(w), C.byref(h))
        n = w.value * h.value
        rgb = np.zeros((n, 3)); ids = np.zeros(n, np.int32)
        L.orc_render_godray(self.h, C.byref(cam), d3(*background), rgb.ctypes.data, ids.ctypes.data)
        return rgb, ids

    def render_antialiasing(self, cam, oversample_times=100, r0=0.007, thresholds=(0.8, 0.8, 0.5, 80.0), background=(0, 0, 0)):
        """Camera.h:450-537; returns (rgb (n,3), edge mask (n,) uint8)"""
        w, h = C.c_int(), C.c_int()
        L.orc_camera_size(C.byref(cam), C.byref(w), C.byref(h))
        n = w.value * h.value
        rgb = np.zeros((n, 3)); mask = np.zeros(n, dtype=np.uint8)
        aa = (C.c_double * 6)(float(oversample_times), r0, *thresholds)
        L.orc_render_godray_aa(self.h, C.byref(cam), aa, d3(*background), rgb.ctypes.data, mask.ctypes.data)
        return rgb, mask

    def __del__(self):
        if getattr(self, "h", None):
            L.orc_godray_free(self.h)
            self.h = None
