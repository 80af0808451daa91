"""Test-side reader for the reference's scene format (JSON + .obj + image files).

This is test infrastructure: it feeds the CPU oracle and cross-checks the product's
own C++ loader.  It follows scene.rs:64-619 of the reference (what each key means,
which are optional) and the behaviour of the wavefront_obj 7.0.0 crate that the
reference uses for meshes (per-object 0-based indices; polygons fan-triangulated),
which is not part of the reference tree ("parity unpinned" for that crate).
"""
import json
import os

import numpy as np

TEX_CONSTANT, TEX_TEST, TEX_CHECKER, TEX_IMAGE, TEX_NOISE, TEX_TURBULENCE = range(6)
MAT_LAMBERT, MAT_METAL, MAT_DIELECTRIC, MAT_DIFFUSE_LIGHT, MAT_ISOTROPIC = range(5)


class SceneError(Exception):
    pass


def parse_obj(text):
    """Returns a list of objects: dict(name, vertices[n,3] f64, tex[m,2] f64, vidx[k,3], tidx[k,3] (-1 = none))."""
    objects = []
    cur = None
    v_off = t_off = 0

    def start(name):
        nonlocal cur, v_off, t_off
        if cur is not None:
            v_off += len(cur["vertices"])
            t_off += len(cur["tex"])
        cur = {"name": name, "vertices": [], "tex": [], "vidx": [], "tidx": []}
        objects.append(cur)

    for raw in text.splitlines():
        line = raw.split("#", 1)[0].strip()
        if not line:
            continue
        parts = line.split()
        tag, args = parts[0], parts[1:]
        if tag == "o":
            start(args[0] if args else "")
        elif tag == "v":
            if cur is None:
                start("")
            cur["vertices"].append([float(args[0]), float(args[1]), float(args[2])])
        elif tag == "vt":
            if cur is None:
                start("")
            cur["tex"].append([float(args[0]), float(args[1]) if len(args) > 1 else 0.0])
        elif tag == "f":
            if cur is None:
                start("")
            vs, ts = [], []
            for a in args:
                f = a.split("/")
                vi = int(f[0])
                vi = vi - 1 - v_off if vi > 0 else len(cur["vertices"]) + vi
                ti = -1
                if len(f) > 1 and f[1] != "":
                    ti = int(f[1])
                    ti = ti - 1 - t_off if ti > 0 else len(cur["tex"]) + ti
                vs.append(vi)
                ts.append(ti)
            if len(vs) < 3:
                raise SceneError("Only triangles are allowed in meshes")
            for k in range(1, len(vs) - 1):
                cur["vidx"].append([vs[0], vs[k], vs[k + 1]])
                cur["tidx"].append([ts[0], ts[k], ts[k + 1]])
        elif tag in ("l", "p"):
            raise SceneError("Only triangles are allowed in meshes")
        # vn, g, s, usemtl, mtllib: no effect on what the reference consumes
    for o in objects:
        o["vertices"] = np.asarray(o["vertices"], dtype=np.float64).reshape(-1, 3)
        o["tex"] = np.asarray(o["tex"], dtype=np.float64).reshape(-1, 2)
        o["vidx"] = np.asarray(o["vidx"], dtype=np.int32).reshape(-1, 3)
        o["tidx"] = np.asarray(o["tidx"], dtype=np.int32).reshape(-1, 3)
    return objects


def load_image(path):
    """Decoded texels as (h, w, channels) uint8 with channels 1 (grey) or 3 (RGB)."""
    from PIL import Image

    im = Image.open(path)
    if im.mode == "L":
        a = np.asarray(im, dtype=np.uint8)[:, :, None]
    else:
        a = np.asarray(im.convert("RGB"), dtype=np.uint8)
    return np.ascontiguousarray(a)


def _vec3(v, what):
    if isinstance(v, dict):
        v = [v["x"], v["y"], v["z"]]
    if len(v) != 3:
        raise SceneError("Deserializing in to %s requires a Vec of length 3!" % what)
    return [float(v[0]), float(v[1]), float(v[2])]


def _transform(d):
    t = d.get("transform", {})
    return (_vec3(t.get("translate", [0, 0, 0]), "Vector3"), _vec3(t.get("rotate", [0, 0, 0]), "Vector3"),
            _vec3(t.get("scale", [1, 1, 1]), "Vector3"))


def _req(d, key):
    if key not in d:
        raise SceneError("Required key %s is missing." % key)
    return d[key]


def read_scene(path, overrides=None):
    """Parse a scene file into a neutral description (plain dicts / numpy arrays).

    overrides: dict with any of resolution_x, resolution_y, samples, aspect_ratio,
    use_importance_sampling.
    """
    overrides = overrides or {}
    base = os.path.dirname(os.path.abspath(path))
    with open(path) as f:
        top = json.load(f)
    if not isinstance(top, dict):
        raise SceneError("Top level scene spec is not a JSON object.")
    lg = dict(_req(top, "Logistics"))
    cam = dict(_req(top, "Camera"))
    for k in ("resolution_x", "resolution_y", "samples", "use_importance_sampling"):
        if k in overrides:
            lg[k] = overrides[k]
    if "aspect_ratio" in overrides:
        cam["aspect_ratio"] = overrides["aspect_ratio"]

    textures = []  # list of dicts; index = id
    images = {}

    def add_texture(tj):
        if not isinstance(tj, dict):
            raise SceneError("Expected JSON object for value in Texture map")
        ty = _req(tj, "type")
        if ty == "Constant":
            c = _vec3(_req(tj, "color"), "RGB")
            textures.append({"type": TEX_CONSTANT, "color": c})
        elif ty == "Test":
            textures.append({"type": TEX_TEST})
        elif ty == "Checker":
            odd = add_texture(_req(tj, "odd"))
            even = add_texture(_req(tj, "even"))
            textures.append({"type": TEX_CHECKER, "repeat": float(_req(tj, "repeat")), "odd": odd, "even": even})
        elif ty == "Image":
            p = os.path.realpath(os.path.join(base, _req(tj, "image_path")))
            if p not in images:
                if not os.path.exists(p):
                    raise SceneError("There was a problem finding the given image path")
                images[p] = load_image(p)
            textures.append({"type": TEX_IMAGE, "texels": images[p]})
        elif ty == "Noise":
            textures.append({"type": TEX_NOISE, "scale": float(_req(tj, "scale"))})
        elif ty == "Turbulence":
            om = float(_req(tj, "omega"))
            if om > 1.0:
                raise SceneError("Turbulence omega is greater than 1.")
            if om < 0.0:
                raise SceneError("Turbulence omega is less than 0.")
            textures.append({"type": TEX_TURBULENCE, "scale": float(_req(tj, "scale")), "depth": int(_req(tj, "depth")), "omega": om})
        else:
            raise SceneError("Unsupported texture type: %s" % ty)
        return len(textures) - 1

    tex_by_name = {}
    tj = _req(top, "Textures")
    if not isinstance(tj, dict):
        raise SceneError("'Textures' is not a JSON object.")
    for name in sorted(tj):  # serde_json's map iterates in key order
        tex_by_name[name] = add_texture(tj[name])

    materials = []
    mat_by_name = {}
    mj = _req(top, "Materials")
    if not isinstance(mj, dict):
        raise SceneError("'Materials' is not a JSON object.")

    def tex_ref(name, what, owner):
        if name not in tex_by_name:
            raise SceneError("Missing %sTexture %s for %s." % (what, name, owner))
        return tex_by_name[name]

    for name in sorted(mj):
        m = mj[name]
        if not isinstance(m, dict):
            raise SceneError("Expected JSON object for value in Materials map")
        ty = _req(m, "type")
        if ty == "Lambert":
            alb = tex_ref(_req(m, "albedo"), "", "Lambert")
            bump = tex_ref(m["bump_map"], "bump map ", "Lambert") if m.get("bump_map") is not None else -1
            materials.append({"type": MAT_LAMBERT, "albedo": alb, "bump": bump, "param": 0.0})
        elif ty == "Metal":
            alb = tex_ref(_req(m, "albedo"), "", "Metal")
            bump = tex_ref(m["bump_map"], "bump map ", "Metal") if m.get("bump_map") is not None else -1
            materials.append({"type": MAT_METAL, "albedo": alb, "bump": bump, "param": float(_req(m, "roughness"))})
        elif ty == "Dielectric":
            materials.append({"type": MAT_DIELECTRIC, "albedo": -1, "bump": -1, "param": float(_req(m, "refractive_index"))})
        elif ty == "DiffuseLight":
            materials.append({"type": MAT_DIFFUSE_LIGHT, "albedo": tex_ref(_req(m, "emission"), "", "DiffuseLight"), "bump": -1, "param": 0.0})
        elif ty == "Isotropic":
            materials.append({"type": MAT_ISOTROPIC, "albedo": tex_ref(_req(m, "albedo"), "", "Isotropic"), "bump": -1, "param": 0.0})
        else:
            raise SceneError("Unsupported material type: %s" % ty)
        mat_by_name[name] = len(materials) - 1

    def shape_items(sj):
        if not isinstance(sj, dict):
            raise SceneError("Expected JSON object for value in Shapes array")
        ty = _req(sj, "type")
        if ty == "Sphere":
            mn = _req(sj, "material")
            if mn not in mat_by_name:
                raise SceneError("Missing Material %s for Sphere." % mn)
            return [{"type": "sphere", "radius": float(_req(sj, "radius")), "material": mat_by_name[mn], "transform": _transform(sj)}]
        if ty == "Mesh":
            mn = _req(sj, "material")
            if mn not in mat_by_name:
                raise SceneError("Missing Material %s for Mesh." % mn)
            with open(os.path.join(base, _req(sj, "file_path"))) as f:
                objs = parse_obj(f.read())
            return [{"type": "mesh", "object": o, "cull": bool(_req(sj, "enable_backface_culling")), "material": mat_by_name[mn],
                     "transform": _transform(sj)} for o in objs]
        if ty == "ConstantMedium":
            pn = _req(sj, "phase_func")
            if pn not in mat_by_name:
                raise SceneError("Missing Phase Function %s for ConstantMedium." % pn)
            inner = shape_items(_req(sj, "boundary"))
            count = sum(1 if i["type"] != "mesh" else len(i["object"]["vidx"]) for i in inner)
            if count != 1:
                raise SceneError("Only single shapes are allowed for constant mediums at the moment.")
            return [{"type": "medium", "boundary": inner[0] if inner[0]["type"] != "mesh" else [i for i in inner if len(i["object"]["vidx"])][0],
                     "density": float(_req(sj, "density")), "phase": mat_by_name[pn]}]
        raise SceneError("Unknown Shape 'type' %s given." % ty)

    shapes = []
    sj = _req(top, "Shapes")
    if not isinstance(sj, list):
        raise SceneError("'Shapes' is not a JSON array.")
    for s in sj:
        shapes.extend(shape_items(s))

    agg = _req(top, "Aggregate")
    if agg not in ("List", "BVH"):
        raise SceneError("Unknown Aggregate 'type' %s given." % agg)

    return {
        "logistics": {"resolution_x": int(_req(lg, "resolution_x")), "resolution_y": int(_req(lg, "resolution_y")),
                      "samples": int(_req(lg, "samples")),
                      "use_importance_sampling": True if lg.get("use_importance_sampling") is None else bool(lg["use_importance_sampling"])},
        "camera": {"position": _vec3(_req(cam, "position"), "Point3"), "look_at": _vec3(_req(cam, "look_at"), "Point3"),
                   "up": _vec3(_req(cam, "up"), "Vector3"), "fov": float(_req(cam, "fov")), "aspect_ratio": float(_req(cam, "aspect_ratio")),
                   "aperture": float(_req(cam, "aperture")), "focus_distance": float(_req(cam, "focus_distance"))},
        "textures": textures, "materials": materials, "shapes": shapes, "aggregate": 1 if agg == "BVH" else 0,
    }
