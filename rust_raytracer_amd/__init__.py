"""rust_raytracer_amd — MI355X path-tracing core behind the trevday/rust-raytracer scene format.

Thin ctypes binding of librrt_hip.so (C ABI in include/rrt.h).  There is no CPU fallback: if the
library is missing this module raises at first use.  Names follow the reference: a `Scene` is what
scene::deserialize returns (scene.rs:64-68), `Scene.hit` is Aggregate::hit (aggregate.rs:118-124),
`Scene.render` is the queue-fill + thread_work loop (main.rs:150-165, 200-257) and `resolve_rgb8`
is the final conversion (main.rs:173-186).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RRT_LIB") or os.path.join(_HERE, "librrt_hip.so")

BG_BLACK, BG_SKY = 0, 1
# Traversal modes (rrt.h, RRT_AABB_*).  Shallow trees (at most 16 nodes, Lists): all three return the same hits, bit for bit.  Deep
# trees: REFERENCE = EXHAUSTIVE = the reference; TIGHT may drop a rounding artefact of the triangle test for a ray within about
# 0.01 rad of a triangle's plane.  Clean trees ignore the mode.
AABB_TIGHT, AABB_REFERENCE, AABB_EXHAUSTIVE = 0, 1, 2
RENDER_ACCUMULATE = 1
SHARD_SAMPLES, SHARD_TILES = 0, 1
KIND_TRIANGLE, KIND_SPHERE, KIND_MEDIUM = 0, 1, 2
T_MIN = 0.001
T_MAX = float(np.finfo(np.float32).max)


class RrtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("rrt error %d: %s" % (code, message))
        self.code = code
        self.message = message


class _Node(C.Structure):
    _fields_ = [("bmin", C.c_float * 3), ("a", C.c_uint32), ("bmax", C.c_float * 3), ("b", C.c_uint32)]


class _Slot(C.Structure):
    _fields_ = [("v0", C.c_float * 3), ("prim", C.c_uint32), ("e1", C.c_float * 3), ("info", C.c_uint32), ("e2", C.c_float * 3), ("aux", C.c_uint32)]


class _TriShade(C.Structure):
    _fields_ = [("nrm", C.c_float * 3), ("pad", C.c_uint32), ("uv0", C.c_float * 2), ("uv1", C.c_float * 2), ("uv2", C.c_float * 2),
                ("pu", C.c_float * 3), ("pv", C.c_float * 3)]


class _Sphere(C.Structure):
    _fields_ = [("w2l", C.c_float * 12), ("l2w", C.c_float * 12), ("radius", C.c_float), ("material", C.c_uint32), ("pad", C.c_uint32 * 6)]


class _Medium(C.Structure):
    _fields_ = [("boundary_kind", C.c_uint32), ("boundary_index", C.c_uint32), ("density", C.c_float), ("material", C.c_uint32)]


class _Material(C.Structure):
    _fields_ = [("type", C.c_uint32), ("albedo", C.c_int32), ("bump", C.c_int32), ("param", C.c_float)]


class _Texture(C.Structure):
    _fields_ = [("type", C.c_uint32), ("a", C.c_float), ("b", C.c_float), ("c", C.c_float), ("u0", C.c_uint32), ("u1", C.c_uint32),
                ("u2", C.c_uint32), ("u3", C.c_uint32)]


class _Light(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("index", C.c_uint32), ("cull", C.c_uint32), ("area", C.c_float), ("v0", C.c_float * 3),
                ("v1", C.c_float * 3), ("v2", C.c_float * 3), ("e1", C.c_float * 3), ("e2", C.c_float * 3), ("nrm", C.c_float * 3),
                ("pad", C.c_uint32 * 2)]


class _Camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("lower_left_corner", C.c_float * 3), ("horizontal", C.c_float * 3), ("vertical", C.c_float * 3),
                ("u", C.c_float * 3), ("v", C.c_float * 3), ("lens_radius", C.c_float)]


class _SceneDesc(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("nodes", C.POINTER(_Node)), ("n_slots", C.c_uint32), ("slots", C.POINTER(_Slot)),
                ("tri_shade", C.POINTER(_TriShade)), ("n_spheres", C.c_uint32), ("spheres", C.POINTER(_Sphere)), ("n_media", C.c_uint32),
                ("media", C.POINTER(_Medium)), ("n_boundary_tris", C.c_uint32), ("boundary_tris", C.POINTER(_Slot)),
                ("boundary_tri_shade", C.POINTER(_TriShade)), ("n_materials", C.c_uint32), ("materials", C.POINTER(_Material)),
                ("n_textures", C.c_uint32), ("textures", C.POINTER(_Texture)), ("texel_bytes", C.c_uint64), ("texels", C.POINTER(C.c_uint8)),
                ("n_lights", C.c_uint32), ("lights", C.POINTER(_Light)), ("camera", _Camera), ("resolution_x", C.c_uint32),
                ("resolution_y", C.c_uint32), ("samples", C.c_uint32), ("bvh_depth", C.c_uint32), ("aggregate_kind", C.c_uint32)]


class _RenderParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("samples", C.c_uint32), ("first_sample", C.c_uint32), ("seed", C.c_uint32),
                ("background", C.c_uint32), ("max_depth", C.c_int32), ("aabb_mode", C.c_uint32), ("tile_count", C.c_uint32),
                ("tile_index", C.c_uint32), ("flags", C.c_uint32)]


class _Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("paths", C.c_uint64), ("render_seconds", C.c_double), ("trace_seconds", C.c_double),
                ("accumulate_seconds", C.c_double), ("reduce_seconds", C.c_double), ("copy_seconds", C.c_double)]


def _stats_dict(st):
    return {"rays": st.rays, "paths": st.paths, "seconds": st.render_seconds, "trace_seconds": st.trace_seconds,
            "accumulate_seconds": st.accumulate_seconds, "reduce_seconds": st.reduce_seconds, "copy_seconds": st.copy_seconds}


_lib = None
_f3 = C.c_float * 3


def lib():
    """Loads librrt_hip.so; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("librrt_hip.so has not been built (run `python rust_raytracer_amd/build.py`); there is no CPU fallback")
    # PyTorch ships its own HIP runtime (libamdhip64.so.7).  Import it first so that this process holds exactly one
    # runtime and torch tensors, streams and RCCL buffers are valid arguments to the *_device entry points.
    import torch  # noqa: F401

    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.rrt_last_error.restype = C.c_char_p
    L.rrt_version.restype = C.c_char_p
    L.rrt_host_scene_load.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.POINTER(vp)]
    L.rrt_host_scene_load_as.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_int, C.POINTER(vp)]
    L.rrt_host_scene_new.argtypes = [C.POINTER(vp)]
    L.rrt_host_add_texture.argtypes = [vp, C.POINTER(_Texture), vp, C.POINTER(C.c_int32)]
    L.rrt_host_add_material.argtypes = [vp, C.c_uint32, C.c_int32, C.c_int32, C.c_float, C.POINTER(C.c_int32)]
    L.rrt_host_add_sphere.argtypes = [vp, _f3, _f3, _f3, C.c_float, C.c_int32]
    L.rrt_host_add_mesh.argtypes = [vp, _f3, _f3, _f3, C.c_uint32, vp, C.c_uint32, vp, C.c_uint32, vp, vp, C.c_int, C.c_int32]
    L.rrt_host_add_medium_sphere.argtypes = [vp, _f3, _f3, _f3, C.c_float, C.c_int32, C.c_float, C.c_int32]
    L.rrt_host_wrap_last_medium.argtypes = [vp, C.c_float, C.c_int32]
    L.rrt_host_set_camera.argtypes = [vp, _f3, _f3, _f3, C.c_float, C.c_float, C.c_float, C.c_float]
    L.rrt_host_set_logistics.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
    L.rrt_host_finalize.argtypes = [vp, C.c_int]
    L.rrt_host_desc.argtypes = [vp]
    L.rrt_host_desc.restype = C.POINTER(_SceneDesc)
    L.rrt_host_scene_times.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.rrt_host_scene_destroy.argtypes = [vp]
    L.rrt_scene_create.argtypes = [C.POINTER(_SceneDesc), C.POINTER(vp)]
    L.rrt_scene_destroy.argtypes = [vp]
    L.rrt_scene_create_on.argtypes = [C.POINTER(_SceneDesc), C.c_int, C.POINTER(vp)]
    L.rrt_scene_check.argtypes = [C.POINTER(_SceneDesc), C.POINTER(C.c_uint64)]
    L.rrt_scene_upload_seconds.argtypes = [vp, C.POINTER(C.c_double)]
    L.rrt_scene_upload_breakdown.argtypes = [vp] + [C.POINTER(C.c_double)] * 4
    L.rrt_device_count.argtypes = [C.POINTER(C.c_int)]
    L.rrt_render_multi.argtypes = [C.POINTER(vp), C.c_uint32, C.c_uint32, C.POINTER(_RenderParams), vp, C.POINTER(_Stats)]
    L.rrt_intersect_batch.argtypes = [vp, vp, vp, C.c_size_t, C.c_float, C.c_float, C.c_uint32, C.c_uint32, vp, vp]
    L.rrt_intersect_batch_device.argtypes = [vp, vp, vp, C.c_size_t, C.c_float, C.c_float, C.c_uint32, C.c_uint32, vp, vp, vp]
    L.rrt_render.argtypes = [vp, C.POINTER(_RenderParams), vp, C.POINTER(_Stats)]
    L.rrt_render_device.argtypes = [vp, C.POINTER(_RenderParams), vp, vp, vp]
    L.rrt_render_device_noalloc.argtypes = [vp, C.POINTER(_RenderParams), vp, vp, vp]
    L.rrt_scene_set_staging_limit.argtypes = [vp, C.c_uint64]
    L.rrt_scene_set_lds_stack.argtypes = [vp, C.c_uint32]
    L.rrt_scene_set_pooling.argtypes = [vp, C.c_int]
    L.rrt_scene_set_timing.argtypes = [vp, C.c_int]
    L.rrt_scene_kernel_times.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
    L.rrt_resolve_rgb8.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp]
    L.rrt_resolve_rgb8_device.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp]
    L.rrt_probe.argtypes = [vp, C.c_uint32, C.c_uint32, vp, C.c_size_t, vp]
    L.rrt_selftest_reciprocal.argtypes = [C.POINTER(C.c_uint64)]
    L.rrt_write_png.argtypes = [C.c_char_p, vp, C.c_uint32, C.c_uint32, C.c_int]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise RrtError(rc, lib().rrt_last_error().decode(errors="replace"))


def version():
    return lib().rrt_version().decode()


def _vec(v):
    return _f3(float(v[0]), float(v[1]), float(v[2]))


class HostScene:
    """The scene as scene::deserialize builds it (shapes, light list, SAH BVH), flattened for the device."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def load(cls, path, width=0, height=0, samples=0, aspect_ratio=float("nan"), tree="BVH"):
        """tree: what the file's "BVH" aggregate becomes - "BVH" (reference tree, host build), "BVH_GPU" (same tree, device build)
        or "BVH_CLEAN" (clean tree with the List aggregate's results)."""
        h = C.c_void_p()
        _check(lib().rrt_host_scene_load_as(os.fsencode(path), width, height, samples, aspect_ratio, {"BVH": 1, "BVH_GPU": 2, "BVH_CLEAN": 3}[tree],
                                            C.byref(h)))
        return cls(h)

    @classmethod
    def new(cls):
        h = C.c_void_p()
        _check(lib().rrt_host_scene_new(C.byref(h)))
        return cls(h)

    # builder calls, one per scene.rs deserialize_* function
    def add_texture(self, type, a=0.0, b=0.0, c=0.0, u0=0, u1=0, u2=0, texels=None):
        t = _Texture(type, a, b, c, u0, u1, u2, 0)
        tid = C.c_int32(-1)
        px = None
        if texels is not None:
            texels = np.ascontiguousarray(texels, dtype=np.uint8)
            px = texels.ctypes.data
        _check(lib().rrt_host_add_texture(self._h, C.byref(t), px, C.byref(tid)))
        return tid.value

    def add_material(self, type, albedo=-1, bump=-1, param=0.0):
        mid = C.c_int32(-1)
        _check(lib().rrt_host_add_material(self._h, type, albedo, bump, param, C.byref(mid)))
        return mid.value

    def add_sphere(self, radius, material, translate=(0, 0, 0), rotate=(0, 0, 0), scale=(1, 1, 1)):
        _check(lib().rrt_host_add_sphere(self._h, _vec(translate), _vec(rotate), _vec(scale), radius, material))

    def add_mesh(self, vertices, triangles, material, cull=False, tex=None, tex_triangles=None, translate=(0, 0, 0), rotate=(0, 0, 0),
                 scale=(1, 1, 1)):
        v = np.ascontiguousarray(vertices, dtype=np.float64).reshape(-1, 3)
        vi = np.ascontiguousarray(triangles, dtype=np.int32).reshape(-1, 3)
        t = np.ascontiguousarray(tex, dtype=np.float64).reshape(-1, 2) if tex is not None else np.zeros((0, 2))
        ti = np.ascontiguousarray(tex_triangles, dtype=np.int32).reshape(-1, 3) if tex_triangles is not None else None
        _check(lib().rrt_host_add_mesh(self._h, _vec(translate), _vec(rotate), _vec(scale), len(v), v.ctypes.data, len(t),
                                       t.ctypes.data if len(t) else None, len(vi), vi.ctypes.data, ti.ctypes.data if ti is not None else None,
                                       1 if cull else 0, material))

    def add_medium_sphere(self, radius, boundary_material, density, phase_material, translate=(0, 0, 0), rotate=(0, 0, 0), scale=(1, 1, 1)):
        _check(lib().rrt_host_add_medium_sphere(self._h, _vec(translate), _vec(rotate), _vec(scale), radius, boundary_material, density,
                                                phase_material))

    def wrap_last_medium(self, density, phase_material):
        """The medium added last becomes the boundary of a new one, which takes its place in the scene (scene.rs:573-602 with a
        ConstantMedium for a boundary)."""
        _check(lib().rrt_host_wrap_last_medium(self._h, density, phase_material))

    def set_camera(self, position, look_at, up, fov, aspect_ratio, aperture, focus_distance):
        _check(lib().rrt_host_set_camera(self._h, _vec(position), _vec(look_at), _vec(up), fov, aspect_ratio, aperture, focus_distance))

    def set_logistics(self, resolution_x, resolution_y, samples, use_importance_sampling=True):
        _check(lib().rrt_host_set_logistics(self._h, resolution_x, resolution_y, samples, 1 if use_importance_sampling else 0))

    def finalize(self, aggregate="BVH"):
        _check(lib().rrt_host_finalize(self._h, {"List": 0, "BVH": 1, "BVH_GPU": 2, "BVH_CLEAN": 3}[aggregate]))

    @property
    def desc(self):
        d = lib().rrt_host_desc(self._h)
        if not d:
            raise RrtError(1, lib().rrt_last_error().decode())
        return d.contents

    # flattened arrays as numpy (copies), for inspection and tests
    def nodes(self):
        d = self.desc
        out = np.zeros(d.n_nodes, dtype=[("bmin", "f4", 3), ("a", "u4"), ("bmax", "f4", 3), ("b", "u4")])
        if d.n_nodes:
            C.memmove(out.ctypes.data, d.nodes, out.nbytes)
        return out

    def slots(self):
        d = self.desc
        out = np.zeros(d.n_slots, dtype=[("v0", "f4", 3), ("prim", "u4"), ("e1", "f4", 3), ("info", "u4"), ("e2", "f4", 3), ("aux", "u4")])
        if d.n_slots:
            C.memmove(out.ctypes.data, d.slots, out.nbytes)
        return out

    def lights(self):
        d = self.desc
        out = np.zeros(d.n_lights, dtype=[("kind", "u4"), ("index", "u4"), ("cull", "u4"), ("area", "f4"), ("v0", "f4", 3), ("v1", "f4", 3),
                                          ("v2", "f4", 3), ("e1", "f4", 3), ("e2", "f4", 3), ("nrm", "f4", 3), ("pad", "u4", 2)])
        if d.n_lights:
            C.memmove(out.ctypes.data, d.lights, out.nbytes)
        return out

    def media(self):
        d = self.desc
        out = np.zeros(d.n_media, dtype=[("boundary_kind", "u4"), ("boundary_index", "u4"), ("density", "f4"), ("material", "u4")])
        if d.n_media:
            C.memmove(out.ctypes.data, d.media, out.nbytes)
        return out

    def textures(self):
        d = self.desc
        out = np.zeros(d.n_textures, dtype=[("type", "u4"), ("a", "f4"), ("b", "f4"), ("c", "f4"), ("u0", "u4"), ("u1", "u4"), ("u2", "u4"),
                                            ("u3", "u4")])
        if d.n_textures:
            C.memmove(out.ctypes.data, d.textures, out.nbytes)
        return out

    def texels(self):
        d = self.desc
        out = np.zeros(d.texel_bytes, dtype=np.uint8)
        if d.texel_bytes:
            C.memmove(out.ctypes.data, d.texels, out.nbytes)
        return out

    def camera(self):
        c = self.desc.camera
        return np.array(list(c.origin) + list(c.lower_left_corner) + list(c.horizontal) + list(c.vertical) + list(c.u) + list(c.v)
                        + [c.lens_radius], dtype=np.float32)

    def check(self):
        """What uploading this scene would put into HBM, worked out on the CPU alone (rrt_scene_check): a dict of record counts, the
        traversal stack depth, the kernel variant's feature bits and the harvest / deep / clean choices."""
        v = (C.c_uint64 * 8)()
        _check(lib().rrt_scene_check(C.byref(self.desc), v))
        return {"node_records": v[0], "pair_records": v[1], "plain_records": v[2], "guard_entries": v[3], "stack_depth": v[4], "features": v[5],
                "harvest": bool(v[6] & 1), "deep": bool(v[6] & 2), "clean": bool(v[6] & 4), "desc_hash": v[7]}

    def times(self):
        """(load_seconds, build_seconds): reading + decoding the scene file, and building lights / aggregate / flattened arrays."""
        a, b = C.c_double(0), C.c_double(0)
        _check(lib().rrt_host_scene_times(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def close(self):
        if self._h:
            lib().rrt_host_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Scene:
    """A scene resident in the current GPU's HBM."""

    def __init__(self, host, device=None):
        """Uploads to the current device, or to `device` (an index) when given."""
        self.host = host
        d = host.desc
        self.width, self.height, self.samples = d.resolution_x, d.resolution_y, d.samples
        h = C.c_void_p()
        if device is None:
            _check(lib().rrt_scene_create(C.byref(d), C.byref(h)))
        else:
            _check(lib().rrt_scene_create_on(C.byref(d), device, C.byref(h)))
        self._h = h

    def upload_seconds(self):
        """Host wall-clock seconds rrt_scene_create took."""
        t = C.c_double(0)
        _check(lib().rrt_scene_upload_seconds(self._h, C.byref(t)))
        return t.value

    def upload_parts(self):
        """What rrt_scene_create's time went to, in ms: the first device call of the process (context), the first access to the
        library's code object (its load onto the device), host-side preparation, device allocations and copies."""
        v = [C.c_double(0) for _ in range(4)]
        _check(lib().rrt_scene_upload_breakdown(self._h, *[C.byref(x) for x in v]))
        return {"first_device_call": v[0].value * 1e3, "code_object": v[1].value * 1e3, "host_preparation": v[2].value * 1e3, "allocations_and_copies": v[3].value * 1e3}

    @classmethod
    def load(cls, path, width=0, height=0, samples=0, aspect_ratio=float("nan"), tree="BVH"):
        return cls(HostScene.load(path, width, height, samples, aspect_ratio, tree))

    def hit(self, origins, directions, t_min=T_MIN, t_max=T_MAX, aabb_mode=AABB_TIGHT, seed=0):
        """Aggregate::hit for a batch of rays: (prim_id int32[n] with -1 = miss, t float32[n])."""
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        if len(o) != len(d):
            raise ValueError("origins and directions differ in length")
        prim = np.full(len(o), -1, dtype=np.int32)
        t = np.zeros(len(o), dtype=np.float32)
        _check(lib().rrt_intersect_batch(self._h, o.ctypes.data, d.ctypes.data, len(o), t_min, t_max, aabb_mode, seed, prim.ctypes.data,
                                         t.ctypes.data))
        return prim, t

    def _params(self, width, height, samples, first_sample, seed, background, max_depth, aabb_mode, tile_count, tile_index, accumulate=False):
        return _RenderParams(width or 0, height or 0, samples or 0, first_sample, seed, background, max_depth, aabb_mode, tile_count,
                             tile_index, RENDER_ACCUMULATE if accumulate else 0)

    def render(self, width=None, height=None, samples=None, first_sample=0, seed=0, background=BG_BLACK, max_depth=-1,
               aabb_mode=AABB_TIGHT, tile_count=1, tile_index=0, out=None, accumulate=False):
        """Per-pixel radiance sums, float32 [H, W, 3], and the stats dict (rays, paths, seconds).
        With accumulate=True the samples are added to the sums already in `out` (earlier sample ranges)."""
        W, H = width or self.width, height or self.height
        img = out if out is not None else np.zeros((H, W, 3), dtype=np.float32)
        if img.shape != (H, W, 3) or img.dtype != np.float32 or not img.flags["C_CONTIGUOUS"]:
            raise ValueError("out must be a C-contiguous float32 array of shape (H, W, 3)")
        p = self._params(width, height, samples, first_sample, seed, background, max_depth, aabb_mode, tile_count, tile_index, accumulate)
        st = _Stats()
        _check(lib().rrt_render(self._h, C.byref(p), img.ctypes.data, C.byref(st)))
        return img, _stats_dict(st)

    def render_device(self, d_rgb_sum, d_counters=0, stream=0, **kw):
        """Enqueues the render on `stream` into a device buffer (raw pointers as ints); does not synchronise.
        The buffer must hold height x width x 3 floats of the frame that is rendered (the scene's Logistics unless width / height
        are given); pass out_floats = its size to have that checked here - the library cannot see how large a device buffer is.
        noalloc=True takes rrt_render_device_noalloc: staging in the library's built-in area, no allocation during the call."""
        W, H = kw.get("width") or self.width, kw.get("height") or self.height
        if "out_floats" in kw and kw["out_floats"] < W * H * 3:
            raise ValueError("the output buffer holds %d floats, the %dx%d frame needs %d" % (kw["out_floats"], W, H, W * H * 3))
        p = self._params(kw.get("width"), kw.get("height"), kw.get("samples"), kw.get("first_sample", 0), kw.get("seed", 0),
                         kw.get("background", BG_BLACK), kw.get("max_depth", -1), kw.get("aabb_mode", AABB_TIGHT), kw.get("tile_count", 1),
                         kw.get("tile_index", 0), kw.get("accumulate", False))
        entry = lib().rrt_render_device_noalloc if kw.get("noalloc") else lib().rrt_render_device
        _check(entry(self._h, C.byref(p), d_rgb_sum, d_counters or None, stream or None))

    def set_staging_limit(self, nbytes):
        """Memory budget of the per-sample radiance staging buffer (0 = the 16 GiB default)."""
        _check(lib().rrt_scene_set_staging_limit(self._h, nbytes))

    def set_lds_stack(self, entries):
        """Deep trees: stack entries per ray kept in LDS by the render kernel (0 = the default); speed only, same results."""
        _check(lib().rrt_scene_set_lds_stack(self._h, entries))

    def set_pooling(self, enable=True):
        """Shallow plain scenes: the pooled schedule (default) or one path per lane; speed only, same results."""
        _check(lib().rrt_scene_set_pooling(self._h, 1 if enable else 0))

    def set_timing(self, enable=True):
        """Record device events around the trace and accumulate kernels of every following render pass."""
        _check(lib().rrt_scene_set_timing(self._h, 1 if enable else 0))

    def kernel_times(self):
        """(trace_ms, accumulate_ms, passes) summed over the passes recorded since the last call; waits for them."""
        t, a, n = C.c_double(0), C.c_double(0), C.c_uint32(0)
        _check(lib().rrt_scene_kernel_times(self._h, C.byref(t), C.byref(a), C.byref(n)))
        return t.value, a.value, n.value

    def probe(self, which, arg, inp, out_words):
        inp = np.ascontiguousarray(inp)
        n = inp.shape[0]
        out = np.zeros((n, out_words), dtype=np.float32)
        _check(lib().rrt_probe(self._h, which, arg, inp.ctypes.data, n, out.ctypes.data))
        return out

    def close(self):
        if self._h:
            lib().rrt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def device_count():
    n = C.c_int(0)
    _check(lib().rrt_device_count(C.byref(n)))
    return n.value


def render_multi(scenes, shard=SHARD_SAMPLES, width=None, height=None, samples=None, first_sample=0, seed=0, background=BG_BLACK, max_depth=-1,
                 aabb_mode=AABB_TIGHT, out=None, accumulate=False):
    """rrt_render_multi: one frame traced by several Scene handles (one per GPU, the same description in each), one host thread
    per handle, the shares gathered and added on the first handle's device.  Returns (float32 [H, W, 3] sums, stats dict)."""
    first = scenes[0]
    W, H = width or first.width, height or first.height
    img = out if out is not None else np.zeros((H, W, 3), dtype=np.float32)
    if img.shape != (H, W, 3) or img.dtype != np.float32 or not img.flags["C_CONTIGUOUS"]:
        raise ValueError("out must be a C-contiguous float32 array of shape (H, W, 3)")
    p = _RenderParams(width or 0, height or 0, samples or 0, first_sample, seed, background, max_depth, aabb_mode, 0, 0,
                      RENDER_ACCUMULATE if accumulate else 0)
    handles = (C.c_void_p * len(scenes))(*[s._h for s in scenes])
    st = _Stats()
    _check(lib().rrt_render_multi(handles, len(scenes), shard, C.byref(p), img.ctypes.data, C.byref(st)))
    return img, _stats_dict(st)


def write_png(path, rgb8, create_new=True):
    """Writes an (H, W, 3) uint8 image as PNG; refuses to overwrite an existing file unless create_new is False."""
    a = np.ascontiguousarray(rgb8, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("rgb8 must have shape (H, W, 3)")
    _check(lib().rrt_write_png(os.fsencode(path), a.ctypes.data, a.shape[1], a.shape[0], 1 if create_new else 0))


def selftest_reciprocal():
    """Number of float bit patterns (of all 2^32) on which the kernel's fast reciprocal differs from IEEE division."""
    n = C.c_uint64(0)
    _check(lib().rrt_selftest_reciprocal(C.byref(n)))
    return n.value


def resolve_rgb8(rgb_sum, spp):
    """main.rs:173-186 on the GPU: uint8 [H, W, 3]."""
    a = np.ascontiguousarray(rgb_sum, dtype=np.float32)
    H, W, _ = a.shape
    out = np.zeros((H, W, 3), dtype=np.uint8)
    _check(lib().rrt_resolve_rgb8(a.ctypes.data, W, H, spp, out.ctypes.data))
    return out


def resolve_rgb8_device(d_rgb_sum, width, height, spp, d_out, stream=0):
    _check(lib().rrt_resolve_rgb8_device(d_rgb_sum, width, height, spp, d_out, stream or None))
