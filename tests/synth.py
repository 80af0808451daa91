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


def host_scene(desc):
    """The product's host-side scene for a neutral description (constant textures, meshes and spheres)."""
    import rust_raytracer_amd as rrt

    hs = rrt.HostScene.new()
    for t in desc["textures"]:
        assert t["type"] == sceneio.TEX_CONSTANT
        hs.add_texture(0, *t["color"])
    for m in desc["materials"]:
        hs.add_material(m["type"], m["albedo"], m["bump"], m["param"])
    for s in desc["shapes"]:
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
    hs.finalize("BVH" if desc["aggregate"] else "List")
    return hs
