"""Seeded synthetic scenes, built identically for the CPU oracle (orc.OracleScene) and for the product (rrt.HostScene).

The scene description is the neutral dictionary sceneio.read_scene() produces, so both builders consume one object."""
import numpy as np

import sceneio


def terrain_desc(n, seed=1, width=64, height=64, samples=2, amplitude=0.15, cull=False):
    """A displaced (n x n)-quad grid over [-1,1]^2 (2*n*n triangles, one object, no texture coordinates), lit by an emissive
    quad above it: the shape of the reference's mesh workloads (one OBJ object + a Square light) at any size."""
    rng = np.random.default_rng(seed)
    g = np.linspace(-1.0, 1.0, n + 1)
    x, z = np.meshgrid(g, g, indexing="xy")
    y = amplitude * (np.sin(3.1 * x) * np.cos(2.3 * z) + 0.35 * rng.standard_normal(x.shape))
    verts = np.stack([x, y, z], axis=-1).reshape(-1, 3).astype(np.float64)
    i = np.arange(n)
    a = (i[:, None] * (n + 1) + i[None, :]).reshape(-1)  # [row z][col x]
    quads = np.stack([a, a + 1, a + n + 2, a + n + 1], axis=-1)
    tris = np.concatenate([quads[:, [0, 2, 1]], quads[:, [0, 3, 2]]], axis=0).astype(np.int32)  # facing +y
    rng.shuffle(tris, axis=0)  # file order is arbitrary in real meshes
    none = np.full_like(tris, -1)
    light_v = np.array([[-0.5, 1.5, -0.5], [0.5, 1.5, -0.5], [0.5, 1.5, 0.5], [-0.5, 1.5, 0.5]], dtype=np.float64)
    light_t = np.array([[0, 1, 2], [0, 2, 3]], dtype=np.int32)  # facing -y
    ident = ([0.0, 0.0, 0.0], [0.0, 0.0, 0.0], [1.0, 1.0, 1.0])
    empty_tex = np.zeros((0, 2))
    return {
        "logistics": {"resolution_x": width, "resolution_y": height, "samples": samples, "use_importance_sampling": True},
        "camera": {"position": [0.0, 1.2, 2.5], "look_at": [0.0, 0.0, 0.0], "up": [0.0, 1.0, 0.0], "fov": 45.0,
                   "aspect_ratio": width / height, "aperture": 0.0, "focus_distance": 1.0},
        "textures": [{"type": sceneio.TEX_CONSTANT, "color": [0.6, 0.55, 0.5]}, {"type": sceneio.TEX_CONSTANT, "color": [12.0, 12.0, 12.0]}],
        "materials": [{"type": sceneio.MAT_LAMBERT, "albedo": 0, "bump": -1, "param": 0.0},
                      {"type": sceneio.MAT_DIFFUSE_LIGHT, "albedo": 1, "bump": -1, "param": 0.0}],
        "shapes": [
            {"type": "mesh", "object": {"vertices": verts, "tex": empty_tex, "vidx": tris, "tidx": none}, "cull": cull, "material": 0,
             "transform": ident},
            {"type": "mesh", "object": {"vertices": light_v, "tex": empty_tex, "vidx": light_t, "tidx": np.full_like(light_t, -1)},
             "cull": False, "material": 1, "transform": ident},
        ],
        "aggregate": 1,
    }


def soup_desc(n=3000, seed=7, width=40, height=32, samples=2):
    """n random triangles in [-1,1]^3 with sizes over three orders of magnitude (a few span the whole scene), one object, lit
    by a quad above: boxes that overlap heavily, unlike the terrain's tidy grid."""
    rng = np.random.default_rng(seed)
    centre = rng.uniform(-1.0, 1.0, (n, 1, 3))
    size = 10.0 ** rng.uniform(-3.0, 0.0, (n, 1, 1))
    verts = (centre + size * rng.uniform(-1.0, 1.0, (n, 3, 3))).reshape(-1, 3).astype(np.float64)
    tris = np.arange(3 * n, dtype=np.int32).reshape(n, 3)
    d = terrain_desc(2, seed=seed, width=width, height=height, samples=samples)
    d["shapes"][0]["object"] = {"vertices": verts, "tex": np.zeros((0, 2)), "vidx": tris, "tidx": np.full_like(tris, -1)}
    return d


def spheres_desc(n=40, seed=3, width=48, height=40, samples=4, media=0, nested=False):
    """Rotated, anisotropically scaled spheres (whose reference bounding box, centre +- radius, ignores the scale and so does
    not contain them), a few with negative radius, over a two-triangle floor; Lambert, Metal, Dielectric and one light.
    media: that many constant-density media bounded by spheres; nested: every medium but each third one becomes the boundary
    of a second medium, and each third of those of a third."""
    rng = np.random.default_rng(seed)
    shapes = []
    for i in range(n):
        tr = [float(v) for v in rng.uniform(-4, 4, 3)]
        tr[1] = float(rng.uniform(0.2, 3.0))
        ro = [float(v) for v in rng.uniform(0, 360, 3)]
        sc = [float(v) for v in rng.uniform(0.5, 3.0, 3)] if i % 3 else [1.0, 1.0, 1.0]
        radius = float(rng.uniform(0.15, 0.5)) * (-1.0 if i % 11 == 5 else 1.0)
        shapes.append({"type": "sphere", "radius": radius, "material": int(i % 4), "transform": (tr, ro, sc)})
    for i in range(media):  # constant-density media bounded by (scaled) spheres, Isotropic phase function = material 4
        tr = [float(v) for v in rng.uniform(-3, 3, 3)]
        tr[1] = float(rng.uniform(0.5, 2.0))
        sc = [float(v) for v in rng.uniform(0.8, 2.0, 3)]
        boundary = {"type": "sphere", "radius": float(rng.uniform(0.4, 0.9)), "material": 0, "transform": (tr, [0.0, 0.0, 0.0], sc)}
        medium = {"type": "medium", "boundary": boundary, "density": float(rng.uniform(0.5, 3.0)), "phase": 4}
        where = int(rng.integers(0, len(shapes)))
        if nested and i % 3 != 2:
            # an outer medium is stopped inside the short stretch between its boundary's two answers: densities grow outwards
            medium = {"type": "medium", "boundary": medium, "density": 6.0 + 2.0 * (i % 4), "phase": 4}
            if i % 3 == 0:
                medium = {"type": "medium", "boundary": medium, "density": 30.0 + 10.0 * (i % 4), "phase": 4}
        shapes.insert(where, medium)
    floor_v = np.array([[-8, 0, -8], [8, 0, -8], [8, 0, 8], [-8, 0, 8]], dtype=np.float64)
    floor_t = np.array([[0, 2, 1], [0, 3, 2]], dtype=np.int32)
    shapes.append({"type": "mesh", "object": {"vertices": floor_v, "tex": np.zeros((0, 2)), "vidx": floor_t, "tidx": np.full_like(floor_t, -1)},
                   "cull": False, "material": 0, "transform": ([0.0, 0.0, 0.0], [0.0, 0.0, 0.0], [1.0, 1.0, 1.0])})
    return {
        "logistics": {"resolution_x": width, "resolution_y": height, "samples": samples, "use_importance_sampling": True},
        "camera": {"position": [0.0, 3.0, 9.0], "look_at": [0.0, 1.0, 0.0], "up": [0.0, 1.0, 0.0], "fov": 50.0,
                   "aspect_ratio": width / height, "aperture": 0.1, "focus_distance": 9.0},
        "textures": [{"type": sceneio.TEX_CONSTANT, "color": [0.7, 0.6, 0.5]}, {"type": sceneio.TEX_CONSTANT, "color": [0.9, 0.9, 0.8]},
                     {"type": sceneio.TEX_CONSTANT, "color": [6.0, 5.0, 4.0]}],
        "materials": [{"type": sceneio.MAT_LAMBERT, "albedo": 0, "bump": -1, "param": 0.0},
                      {"type": sceneio.MAT_METAL, "albedo": 1, "bump": -1, "param": 0.2},
                      {"type": sceneio.MAT_DIELECTRIC, "albedo": -1, "bump": -1, "param": 1.5},
                      {"type": sceneio.MAT_DIFFUSE_LIGHT, "albedo": 2, "bump": -1, "param": 0.0},
                      {"type": sceneio.MAT_ISOTROPIC, "albedo": 1, "bump": -1, "param": 0.0}],
        "shapes": shapes,
        "aggregate": 1,
    }


def slats_desc(n=40, seed=5, width=40, height=32, samples=3, aggregate=0, spheres=3):
    """n parallel quads stacked along z (every third one duplicated in place, so that exact distance ties occur and the
    earlier shape must win), a few spheres between them and a light: every ray along z crosses the boxes of most quads.
    With aggregate=0 ("List") the whole scene is one leaf."""
    rng = np.random.default_rng(seed)
    ident_r, ident_s = [0.0, 0.0, 0.0], [1.0, 1.0, 1.0]
    quad_t = np.array([[0, 1, 2], [0, 2, 3]], dtype=np.int32)
    shapes = []
    for i in range(n):
        z = -2.0 + 4.0 * i / max(n - 1, 1)
        w = float(rng.uniform(0.3, 1.2))
        ox, oy = (float(v) for v in rng.uniform(-0.6, 0.6, 2))
        v = np.array([[ox - w, oy - w, z], [ox + w, oy - w, z], [ox + w, oy + w, z], [ox - w, oy + w, z]], dtype=np.float64)
        tilt = [float(rng.uniform(-8, 8)), float(rng.uniform(-8, 8)), 0.0] if i % 4 == 1 else ident_r
        mesh = {"type": "mesh", "object": {"vertices": v, "tex": np.zeros((0, 2)), "vidx": quad_t, "tidx": np.full_like(quad_t, -1)},
                "cull": bool(i % 5 == 2), "material": int(i % 3), "transform": ([0.0, 0.0, 0.0], tilt, ident_s)}
        shapes.append(mesh)
        if i % 3 == 0:
            shapes.append(dict(mesh, material=int((i + 1) % 3)))
    for i in range(spheres):
        shapes.insert(int(rng.integers(0, len(shapes))),
                      {"type": "sphere", "radius": float(rng.uniform(0.15, 0.4)), "material": int(i % 3),
                       "transform": ([float(v) for v in rng.uniform(-1, 1, 3)], ident_r, ident_s)})
    light_v = np.array([[-0.5, 1.9, -0.5], [0.5, 1.9, -0.5], [0.5, 1.9, 0.5], [-0.5, 1.9, 0.5]], dtype=np.float64)
    shapes.append({"type": "mesh", "object": {"vertices": light_v, "tex": np.zeros((0, 2)), "vidx": quad_t, "tidx": np.full_like(quad_t, -1)},
                   "cull": False, "material": 3, "transform": ([0.0, 0.0, 0.0], ident_r, ident_s)})
    return {
        "logistics": {"resolution_x": width, "resolution_y": height, "samples": samples, "use_importance_sampling": True},
        "camera": {"position": [0.3, 0.2, 5.0], "look_at": [0.0, 0.0, 0.0], "up": [0.0, 1.0, 0.0], "fov": 35.0,
                   "aspect_ratio": width / height, "aperture": 0.0, "focus_distance": 1.0},
        "textures": [{"type": sceneio.TEX_CONSTANT, "color": [0.7, 0.6, 0.5]}, {"type": sceneio.TEX_CONSTANT, "color": [0.9, 0.9, 0.8]},
                     {"type": sceneio.TEX_CONSTANT, "color": [6.0, 5.0, 4.0]}],
        "materials": [{"type": sceneio.MAT_LAMBERT, "albedo": 0, "bump": -1, "param": 0.0},
                      {"type": sceneio.MAT_METAL, "albedo": 1, "bump": -1, "param": 0.1},
                      {"type": sceneio.MAT_DIELECTRIC, "albedo": -1, "bump": -1, "param": 1.5},
                      {"type": sceneio.MAT_DIFFUSE_LIGHT, "albedo": 2, "bump": -1, "param": 0.0}],
        "shapes": shapes,
        "aggregate": aggregate,
    }


def lone_light_desc(width=48, height=40, samples=16):
    """A Lambert sphere and an emissive sphere, both stretched upwards so that they stick out of the boxes the reference gives
    them (shape.rs:129-139), seen from below: two shapes, i.e. a BVH of a single leaf; paths bounce off the Lambert sphere's
    bulge above the leaf's box, and rays from there to the light's bulge never meet that box."""
    return {
        "logistics": {"resolution_x": width, "resolution_y": height, "samples": samples, "use_importance_sampling": True},
        "camera": {"position": [0.0, -3.0, 3.0], "look_at": [0.0, 1.3, 0.0], "up": [0.0, 1.0, 0.0], "fov": 40.0,
                   "aspect_ratio": width / height, "aperture": 0.0, "focus_distance": 1.0},
        "textures": [{"type": sceneio.TEX_CONSTANT, "color": [0.7, 0.6, 0.5]}, {"type": sceneio.TEX_CONSTANT, "color": [5.0, 4.0, 3.0]}],
        "materials": [{"type": sceneio.MAT_LAMBERT, "albedo": 0, "bump": -1, "param": 0.0},
                      {"type": sceneio.MAT_DIFFUSE_LIGHT, "albedo": 1, "bump": -1, "param": 0.0}],
        "shapes": [
            {"type": "sphere", "radius": 0.5, "material": 0, "transform": ([-0.8, 1.0, 0.0], [0.0, 15.0, 0.0], [1.0, 3.0, 1.0])},
            {"type": "sphere", "radius": 0.5, "material": 1, "transform": ([0.8, 1.0, 0.0], [0.0, -25.0, 0.0], [1.0, 3.0, 1.0])},
        ],
        "aggregate": 1,
    }


def floor_and_sphere_light_desc(width=48, height=40, samples=16):
    """A Lambert floor of two triangles and ONE emissive sphere above it, stretched upwards out of the box the reference gives it
    (shape.rs:129-139): three shapes in a single leaf whose only sphere is the light.  The sifting query tests such a sphere with
    the rays' own lanes, and the harvesting kernel takes the light's density from that very test."""
    floor_v = np.array([[-3.0, 0.0, -3.0], [3.0, 0.0, -3.0], [3.0, 0.0, 3.0], [-3.0, 0.0, 3.0]])
    floor_t = np.array([[0, 2, 1], [0, 3, 2]], dtype=np.int32)
    ident = ([0.0, 0.0, 0.0], [0.0, 0.0, 0.0], [1.0, 1.0, 1.0])
    return {
        "logistics": {"resolution_x": width, "resolution_y": height, "samples": samples, "use_importance_sampling": True},
        "camera": {"position": [0.0, 2.5, 6.0], "look_at": [0.0, 0.8, 0.0], "up": [0.0, 1.0, 0.0], "fov": 45.0,
                   "aspect_ratio": width / height, "aperture": 0.0, "focus_distance": 1.0},
        "textures": [{"type": sceneio.TEX_CONSTANT, "color": [0.7, 0.6, 0.5]}, {"type": sceneio.TEX_CONSTANT, "color": [5.0, 4.0, 3.0]}],
        "materials": [{"type": sceneio.MAT_LAMBERT, "albedo": 0, "bump": -1, "param": 0.0},
                      {"type": sceneio.MAT_DIFFUSE_LIGHT, "albedo": 1, "bump": -1, "param": 0.0}],
        "shapes": [
            {"type": "mesh", "object": {"vertices": floor_v, "tex": np.zeros((0, 2)), "vidx": floor_t, "tidx": np.full_like(floor_t, -1)},
             "cull": False, "material": 0, "transform": ident},
            {"type": "sphere", "radius": 0.5, "material": 1, "transform": ([0.4, 1.2, 0.0], [0.0, -25.0, 10.0], [1.0, 2.5, 1.0])},
        ],
        "aggregate": 1,
    }


def host_scene(desc, aggregate=None):
    """The product's host-side scene for a neutral description (constant textures, meshes and spheres).
    aggregate: "List", "BVH" (built on the host) or "BVH_GPU" (built on the device); default: what the description asks for."""
    import rust_raytracer_amd as rrt

    hs = rrt.HostScene.new()
    for t in desc["textures"]:
        assert t["type"] == sceneio.TEX_CONSTANT
        hs.add_texture(0, *t["color"])
    for m in desc["materials"]:
        hs.add_material(m["type"], m["albedo"], m["bump"], m["param"])
    for s in desc["shapes"]:
        if s["type"] == "medium":
            chain = [s]
            while chain[-1]["boundary"]["type"] == "medium":
                chain.append(chain[-1]["boundary"])
            b = chain[-1]["boundary"]
            assert b["type"] == "sphere"
            btr, bro, bsc = b["transform"]
            hs.add_medium_sphere(b["radius"], b["material"], chain[-1]["density"], chain[-1]["phase"], btr, bro, bsc)
            for outer in reversed(chain[:-1]):
                hs.wrap_last_medium(outer["density"], outer["phase"])
            continue
        tr, ro, sc = s["transform"]
        if s["type"] == "sphere":
            hs.add_sphere(s["radius"], s["material"], tr, ro, sc)
        else:
            o = s["object"]
            hs.add_mesh(o["vertices"], o["vidx"], s["material"], cull=s["cull"], translate=tr, rotate=ro, scale=sc)
    c = desc["camera"]
    hs.set_camera(c["position"], c["look_at"], c["up"], c["fov"], c["aspect_ratio"], c["aperture"], c["focus_distance"])
    lg = desc["logistics"]
    hs.set_logistics(lg["resolution_x"], lg["resolution_y"], lg["samples"], lg["use_importance_sampling"])
    hs.finalize(aggregate or ("BVH" if desc["aggregate"] else "List"))
    return hs
