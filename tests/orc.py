"""ctypes binding of oracle/liboracle.so (the CPU restatement; test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

import sceneio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ASSETS = os.path.join(ROOT, "tests", "golden", "assets")

_lib = None

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def build():
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    src = os.path.join(ORACLE_DIR, "oracle.cpp")
    if not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return so


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    vp = C.c_void_p
    L.orc_scene_new.restype = vp
    L.orc_scene_free.argtypes = [vp]
    L.orc_scene_error.restype = C.c_char_p
    L.orc_scene_error.argtypes = [vp]
    L.orc_add_texture_constant.argtypes = [vp, C.c_float, C.c_float, C.c_float]
    L.orc_add_texture_test.argtypes = [vp]
    L.orc_add_texture_checker.argtypes = [vp, C.c_float, C.c_int, C.c_int]
    L.orc_add_texture_image.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, u8p]
    L.orc_add_texture_noise.argtypes = [vp, C.c_float]
    L.orc_add_texture_turbulence.argtypes = [vp, C.c_float, C.c_uint32, C.c_float]
    L.orc_add_material.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_float]
    L.orc_add_sphere.argtypes = [vp, f32p, f32p, f32p, C.c_float, C.c_int]
    L.orc_add_mesh.argtypes = [vp, f32p, f32p, f32p, C.c_uint32, f64p, C.c_uint32, f64p, C.c_uint32, i32p, i32p, C.c_int, C.c_int, C.c_int]
    L.orc_add_medium_sphere.argtypes = [vp, f32p, f32p, f32p, C.c_float, C.c_int, C.c_float, C.c_int]
    L.orc_add_medium_last_boundary.argtypes = [vp, C.c_float, C.c_int]
    L.orc_wrap_last_medium.argtypes = [vp, C.c_float, C.c_int]
    L.orc_set_camera.argtypes = [vp, f32p, f32p, f32p, C.c_float, C.c_float, C.c_float, C.c_float]
    L.orc_set_logistics.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
    L.orc_finalize.argtypes = [vp, C.c_int]
    L.orc_num_shapes.argtypes = [vp]
    L.orc_num_shapes.restype = C.c_uint32
    L.orc_num_important.argtypes = [vp]
    L.orc_num_important.restype = C.c_uint32
    L.orc_get_important.argtypes = [vp, i32p]
    L.orc_bvh_num_nodes.argtypes = [vp]
    L.orc_bvh_num_nodes.restype = C.c_uint32
    L.orc_bvh_export.argtypes = [vp, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_bvh_export.restype = C.c_uint32
    L.orc_get_camera.argtypes = [vp, f32p]
    L.orc_get_shape_bboxes.argtypes = [vp, f32p]
    L.orc_intersect_batch.argtypes = [vp, C.c_uint64, f32p, f32p, C.c_float, C.c_float, C.c_int, C.c_uint32, i32p, f32p, C.c_void_p]
    L.orc_hit_properties.argtypes = [vp, C.c_uint64, f32p, f32p, i32p, f32p, f32p]
    L.orc_camera_rays.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, u32p, u32p, u32p, f32p]
    L.orc_render.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32,
                             C.c_uint32, f32p, C.c_void_p, C.c_void_p]
    L.orc_resolve_rgb8.argtypes = [f32p, C.c_uint32, C.c_uint32, C.c_uint32, u8p]
    L.orc_unit_triangle_hit.argtypes = [f32p, f32p, f32p, C.c_int, f32p, f32p, C.c_float, C.c_float, C.POINTER(C.c_float)]
    L.orc_unit_sphere_hit.argtypes = [f32p, f32p, f32p, C.c_float, f32p, f32p, C.c_float, C.c_float, C.POINTER(C.c_float)]
    L.orc_unit_aabb_intersect.argtypes = [f32p, f32p, f32p, f32p, C.c_float, C.c_float, C.c_int]
    L.orc_unit_noise.argtypes = [C.c_float, C.c_float, C.c_float]
    L.orc_unit_noise.restype = C.c_float
    L.orc_unit_turbulence.argtypes = [C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_float]
    L.orc_unit_turbulence.restype = C.c_float
    L.orc_unit_matrix_inverse.argtypes = [f32p, f32p]
    L.orc_unit_transform_matrix.argtypes = [f32p, f32p, f32p, f32p]
    L.orc_unit_detmath.argtypes = [C.c_int, C.c_uint64, f32p, f32p, f32p]
    L.orc_unit_rng.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, f32p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_float)]
    L.orc_unit_samplers.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_float, f32p]
    L.orc_texture_values.argtypes = [vp, C.c_int, C.c_uint64, f32p, f32p, f32p]
    L.orc_unit_reflect_refract.argtypes = [f32p, f32p, C.c_float, f32p, f32p, C.POINTER(C.c_float), C.c_float]
    _lib = L
    return L


def a32(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float32))


FLAG_RECURSIVE = 1
FLAG_TIGHT_AABB = 2
BG_BLACK, BG_SKY = 0, 1


class OracleScene:
    """Builds an oracle scene from sceneio.read_scene()'s neutral description."""

    def __init__(self, desc):
        L = lib()
        self.L = L
        self.h = L.orc_scene_new()
        self.desc = desc
        self._keep = []
        for t in desc["textures"]:
            ty = t["type"]
            if ty == sceneio.TEX_CONSTANT:
                r = L.orc_add_texture_constant(self.h, *t["color"])
            elif ty == sceneio.TEX_TEST:
                r = L.orc_add_texture_test(self.h)
            elif ty == sceneio.TEX_CHECKER:
                r = L.orc_add_texture_checker(self.h, t["repeat"], t["odd"], t["even"])
            elif ty == sceneio.TEX_IMAGE:
                px = t["texels"]
                r = L.orc_add_texture_image(self.h, px.shape[1], px.shape[0], px.shape[2], px.reshape(-1))
            elif ty == sceneio.TEX_NOISE:
                r = L.orc_add_texture_noise(self.h, t["scale"])
            else:
                r = L.orc_add_texture_turbulence(self.h, t["scale"], t["depth"], t["omega"])
            self._check(r)
        for m in desc["materials"]:
            self._check(L.orc_add_material(self.h, m["type"], m["albedo"], m["bump"], m["param"]))
        for s in desc["shapes"]:
            self._add_shape(s)
        c = desc["camera"]
        L.orc_set_camera(self.h, a32(c["position"]), a32(c["look_at"]), a32(c["up"]), c["fov"], c["aspect_ratio"], c["aperture"],
                         c["focus_distance"])
        lg = desc["logistics"]
        L.orc_set_logistics(self.h, lg["resolution_x"], lg["resolution_y"], lg["samples"], 1 if lg["use_importance_sampling"] else 0)
        self._check(L.orc_finalize(self.h, desc["aggregate"]))
        self.width, self.height, self.samples = lg["resolution_x"], lg["resolution_y"], lg["samples"]

    @classmethod
    def from_file(cls, path, overrides=None):
        return cls(sceneio.read_scene(path, overrides))

    def _check(self, r):
        if r < 0:
            raise sceneio.SceneError(self.L.orc_scene_error(self.h).decode())

    def _add_mesh(self, s, as_boundary):
        o = s["object"]
        tr, ro, sc = s["transform"]
        self._check(self.L.orc_add_mesh(self.h, a32(tr), a32(ro), a32(sc), len(o["vertices"]), np.ascontiguousarray(o["vertices"]).reshape(-1),
                                        len(o["tex"]), np.ascontiguousarray(o["tex"]).reshape(-1) if len(o["tex"]) else np.zeros(2),
                                        len(o["vidx"]), np.ascontiguousarray(o["vidx"]).reshape(-1),
                                        np.ascontiguousarray(o["tidx"]).reshape(-1), 1 if s["cull"] else 0, s["material"],
                                        1 if as_boundary else 0))

    def _add_shape(self, s):
        L = self.L
        if s["type"] == "sphere":
            tr, ro, sc = s["transform"]
            self._check(L.orc_add_sphere(self.h, a32(tr), a32(ro), a32(sc), s["radius"], s["material"]))
        elif s["type"] == "mesh":
            self._add_mesh(s, False)
        else:
            b = s["boundary"]
            if b["type"] == "sphere":
                tr, ro, sc = b["transform"]
                self._check(L.orc_add_medium_sphere(self.h, a32(tr), a32(ro), a32(sc), b["radius"], b["material"], s["density"], s["phase"]))
            elif b["type"] == "mesh":
                self._add_mesh(b, True)
                self._check(L.orc_add_medium_last_boundary(self.h, s["density"], s["phase"]))
            else:  # a medium inside a medium: the inner one first, then it becomes the boundary of this one
                self._add_shape(b)
                self._check(L.orc_wrap_last_medium(self.h, s["density"], s["phase"]))

    def close(self):
        if self.h:
            self.L.orc_scene_free(self.h)
            self.h = None

    # ---- queries ----
    def num_shapes(self):
        return self.L.orc_num_shapes(self.h)

    def important(self):
        n = self.L.orc_num_important(self.h)
        out = np.zeros(max(n, 1), dtype=np.int32)
        self.L.orc_get_important(self.h, out)
        return out[:n]

    def bvh(self):
        n = self.L.orc_bvh_num_nodes(self.h)
        boxes = np.zeros((n, 6), dtype=np.float32)
        meta = np.zeros((n, 4), dtype=np.int32)
        total = self.L.orc_bvh_export(self.h, None, None, None)
        leaf = np.zeros(max(total, 1), dtype=np.int32)
        self.L.orc_bvh_export(self.h, boxes.ctypes.data, meta.ctypes.data, leaf.ctypes.data)
        return boxes, meta, leaf[:total]

    def camera(self):
        out = np.zeros(19, dtype=np.float32)
        self.L.orc_get_camera(self.h, out)
        return out

    def shape_bboxes(self):
        out = np.zeros((self.num_shapes(), 6), dtype=np.float32)
        self.L.orc_get_shape_bboxes(self.h, out)
        return out

    def intersect(self, o, d, t_min=0.001, t_max=float(np.finfo(np.float32).max), tight=False, seed=0, want_stats=False):
        o, d = a32(o).reshape(-1, 3), a32(d).reshape(-1, 3)
        n = len(o)
        prim = np.zeros(n, dtype=np.int32)
        t = np.zeros(n, dtype=np.float32)
        st = np.zeros(4, dtype=np.uint64)
        self.L.orc_intersect_batch(self.h, n, o, d, t_min, t_max, 1 if tight else 0, seed, prim, t, st.ctypes.data)
        return (prim, t, st) if want_stats else (prim, t)

    def hit_properties(self, o, d, prim, t):
        o, d = a32(o).reshape(-1, 3), a32(d).reshape(-1, 3)
        out = np.zeros((len(o), 14), dtype=np.float32)
        self.L.orc_hit_properties(self.h, len(o), o, d, np.ascontiguousarray(prim, dtype=np.int32), a32(t), out)
        return out

    def camera_rays(self, W, H, seed, px, py, sample):
        px = np.ascontiguousarray(px, dtype=np.uint32)
        py = np.ascontiguousarray(py, dtype=np.uint32)
        sample = np.ascontiguousarray(sample, dtype=np.uint32)
        out = np.zeros((len(px), 6), dtype=np.float32)
        self.L.orc_camera_rays(self.h, W, H, seed, len(px), px, py, sample, out)
        return out[:, :3].copy(), out[:, 3:].copy()

    def render(self, W=None, H=None, spp=None, first_sample=0, seed=0, bg=BG_BLACK, flags=0, max_depth=-1, threads=None, rows=None,
               rays_per_pixel=None):
        W = W or self.width
        H = H or self.height
        spp = spp or self.samples
        threads = threads or min(8, os.cpu_count() or 1)
        y0, y1 = rows if rows else (0, H)
        img = np.zeros((H, W, 3), dtype=np.float32)
        st = np.zeros(4, dtype=np.uint64)
        self.L.orc_render(self.h, W, H, spp, first_sample, seed, bg, flags, max_depth, threads, y0, y1, img.reshape(-1), st.ctypes.data,
                          rays_per_pixel.ctypes.data if rays_per_pixel is not None else None)
        return img, st

    def texture_values(self, tex, uv, p):
        uv, p = a32(uv).reshape(-1, 2), a32(p).reshape(-1, 3)
        out = np.zeros((len(uv), 3), dtype=np.float32)
        self.L.orc_texture_values(self.h, tex, len(uv), uv, p, out)
        return out


def resolve_rgb8(rgb_sum, spp):
    H, W, _ = rgb_sum.shape
    out = np.zeros((H, W, 3), dtype=np.uint8)
    lib().orc_resolve_rgb8(np.ascontiguousarray(rgb_sum, dtype=np.float32).reshape(-1), W, H, spp, out.reshape(-1))
    return out


def asset(name):
    return os.path.join(ASSETS, name)
