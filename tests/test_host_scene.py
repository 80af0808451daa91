"""The product's host side (C++ scene loader, SAH BVH, flattening; no GPU needed) against the oracle and the format rules."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

import orc
import sceneio
import rust_raytracer_amd as rrt


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = ["Cornell_Box.json", "Cornell_Box_With_Sphere.json", "Materials.json", "Example_Spec.json", "Globe.json"]


def test_abi_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "rrt.h")).read()
    names = set(re.findall(r"\b(rrt_[a-z0-9_]+)\s*\(", header))
    names -= {"rrt_status"}
    assert len(names) >= 20
    L = rrt.lib()
    for n in sorted(names):
        assert hasattr(L, n), "librrt_hip.so does not export %s" % n
    assert rrt.version().startswith("rrt-mi355x")


def test_struct_layouts_match_header_sizes():
    assert C.sizeof(rrt._Node) == 32 and C.sizeof(rrt._Slot) == 48 and C.sizeof(rrt._TriShade) == 64
    assert C.sizeof(rrt._Sphere) == 128 and C.sizeof(rrt._Medium) == 16 and C.sizeof(rrt._Material) == 16
    assert C.sizeof(rrt._Texture) == 32 and C.sizeof(rrt._Light) == 96 and C.sizeof(rrt._Camera) == 76


@pytest.mark.parametrize("name", SCENES)
def test_flattened_bvh_equals_oracle_bvh(name):
    hs = rrt.HostScene.load(orc.asset(name))
    o = orc.OracleScene.from_file(orc.asset(name))
    nodes, slots = hs.nodes(), hs.slots()
    boxes, meta, leaf = o.bvh()
    assert len(nodes) == len(meta)
    assert np.array_equal(nodes["bmin"], boxes[:, :3]) and np.array_equal(nodes["bmax"], boxes[:, 3:])
    is_leaf = (nodes["b"] & 0x80000000) != 0
    assert np.array_equal(is_leaf.astype(np.int32), meta[:, 0])
    inner = ~is_leaf
    assert np.array_equal(nodes["b"][inner], meta[inner, 1].astype(np.uint32))  # cut axis
    assert np.array_equal(nodes["a"][inner] - np.nonzero(inner)[0], meta[inner, 2].astype(np.uint32))  # right_offset
    assert np.array_equal((nodes["b"][is_leaf] & 0x7FFFFFFF), meta[is_leaf, 3].astype(np.uint32))
    assert np.array_equal(slots["prim"].astype(np.int32), leaf)  # leaf order of shapes
    assert np.array_equal(hs.camera().view(np.uint32), o.camera().view(np.uint32))
    # light list = important shapes in scene order
    imp = o.important()
    assert hs.desc.n_lights == len(imp)
    d = hs.desc
    assert d.bvh_depth >= 1 and d.n_slots == o.num_shapes()


def test_triangle_slots_hold_v0_e1_e2():
    hs = rrt.HostScene.load(orc.asset("Cornell_Box.json"))
    desc = sceneio.read_scene(orc.asset("Cornell_Box.json"))
    slots = hs.slots()
    # first shape of the file: Square.obj scaled to 555 x 555, rotated -90 deg about y, at x = 555 (the green wall)
    s = slots[np.nonzero(slots["prim"] == 0)[0][0]]
    v0, e1, e2 = s["v0"], s["e1"], s["e2"]
    assert abs(v0[0] - 555) < 1e-3 and abs(e1[0]) < 1e-3 and abs(e2[0]) < 1e-3
    assert (s["info"] >> 30) == rrt.KIND_TRIANGLE and (s["info"] & (1 << 28))  # culling on
    assert len(desc["shapes"]) == 8  # 6 squares + 2 cubes as meshes -> 36 triangles
    assert hs.desc.n_slots == 36
    lights = hs.lights()
    assert len(lights) == 2 and np.all(lights["kind"] == rrt.KIND_TRIANGLE)
    assert lights["area"][0] == pytest.approx(0.5 * 130 * 105, rel=1e-5)


def test_jpeg_decoder_matches_pil_for_the_shipped_textures():
    hs = rrt.HostScene.load(orc.asset("Globe.json"))
    tex, px = hs.textures(), hs.texels()
    for t, fn in zip(tex, ["earthmap1k.jpg", "earthbump1k.jpg"]):  # "World", "WorldBump" in key order
        ref = sceneio.load_image(orc.asset(fn))
        got = px[t["u3"]: t["u3"] + int(t["u0"]) * int(t["u1"]) * int(t["u2"])].reshape(t["u1"], t["u0"], t["u2"])
        assert got.shape == ref.shape and np.array_equal(got, ref)


def _texels_of(tmp_path, names):
    """Decodes the image files through the scene loader (one Image texture each) and returns their texel arrays."""
    scene = {"Logistics": {"resolution_x": 8, "resolution_y": 8, "samples": 1},
             "Camera": {"position": [0, 0, 1], "look_at": [0, 0, 0], "up": [0, 1, 0], "fov": 90, "aspect_ratio": 1, "aperture": 0, "focus_distance": 1},
             "Textures": {"T%04d" % i: {"type": "Image", "image_path": n} for i, n in enumerate(names)},
             "Materials": {"M": {"type": "Lambert", "albedo": "T0000"}}, "Aggregate": "BVH",
             "Shapes": [{"type": "Sphere", "radius": 0.5, "material": "M"}]}
    (tmp_path / "images.json").write_text(json.dumps(scene))
    hs = rrt.HostScene.load(str(tmp_path / "images.json"))
    tex, px = hs.textures(), hs.texels()
    return [px[t["u3"]: t["u3"] + int(t["u0"]) * int(t["u1"]) * int(t["u2"])].reshape(t["u1"], t["u0"], t["u2"]) for t in tex]


def test_progressive_and_multi_scan_jpeg_match_pil(tmp_path):
    """image::open (resources.rs:41) takes progressive JPEGs as well as baseline ones.  The shipped textures and random images, saved
    by PIL as progressive files (spectral selection and successive approximation in ten scans, with and without optimised tables,
    with restart intervals, greyscale and 4:4:4 colour, sizes that are no multiple of the block size), must decode to what PIL decodes."""
    from PIL import Image

    rng = np.random.default_rng(1)
    earth = Image.open(orc.asset("earthmap1k.jpg")).convert("RGB")
    bump = Image.open(orc.asset("earthbump1k.jpg")).convert("L")
    smooth = np.clip(np.add.outer(np.arange(61) * 3.0, np.arange(83) * 2.0)[..., None] + rng.normal(0, 20, (61, 83, 3)), 0, 255).astype(np.uint8)
    cases = [("earth_p.jpg", earth, dict(progressive=True, quality=90, subsampling=0)),
             ("earth_po.jpg", earth.resize((333, 167)), dict(progressive=True, optimize=True, quality=75, subsampling=0)),
             ("bump_p.jpg", bump, dict(progressive=True, quality=85)),
             ("bump_small_p.jpg", bump.resize((37, 19)), dict(progressive=True, quality=95)),
             ("smooth_p.jpg", Image.fromarray(smooth), dict(progressive=True, quality=60, subsampling=0)),
             ("noise_p.jpg", Image.fromarray(rng.integers(0, 256, (45, 70, 3), dtype=np.uint8)), dict(progressive=True, quality=98, subsampling=0)),
             ("noise_b.jpg", Image.fromarray(rng.integers(0, 256, (45, 70, 3), dtype=np.uint8)), dict(quality=98, subsampling=0, optimize=True)),
             ("one_p.jpg", Image.fromarray(rng.integers(0, 256, (1, 1, 3), dtype=np.uint8)), dict(progressive=True, subsampling=0))]
    for name, img, kw in cases:
        img.save(tmp_path / name, **kw)
    # restart intervals inside progressive scans (PIL writes them when asked for)
    earth.resize((200, 100)).save(tmp_path / "earth_pr.jpg", progressive=True, quality=80, subsampling=0, restart_marker_blocks=7)
    names = [c[0] for c in cases] + ["earth_pr.jpg"]
    raw = (tmp_path / "earth_p.jpg").read_bytes()
    assert b"\xff\xc2" in raw and raw.count(b"\xff\xda") >= 6  # a progressive frame and its many scans
    for name, got in zip(names, _texels_of(tmp_path, names)):
        ref = sceneio.load_image(str(tmp_path / name))
        assert got.shape == ref.shape, name
        assert np.array_equal(got, ref), "%s: %d texels differ" % (name, int((got != ref).sum()))


def _write_png(path, pixels, depth, ctype, interlace, palette=None):
    """A PNG file of `pixels` ([H, W] or [H, W, samples], integers of `depth` bits), all filter types in turn, Adam7 if asked for."""
    import struct
    import zlib

    a = np.asarray(pixels)
    if a.ndim == 2:
        a = a[:, :, None]
    H, W, S = a.shape

    def rows_of(sub):
        out = bytearray()
        h, w, _ = sub.shape
        bpp = max(1, S * depth // 8)
        prev = np.zeros(((w * S * depth + 7) // 8,), dtype=np.uint8)
        for y in range(h):
            if depth == 16:
                line = sub[y].astype(">u2").tobytes()
            elif depth == 8:
                line = sub[y].astype(np.uint8).tobytes()
            else:
                bits = np.unpackbits(sub[y].astype(np.uint8).reshape(-1, 1), axis=1)[:, 8 - depth:].reshape(-1)
                line = np.packbits(bits).tobytes()
            cur = np.frombuffer(line, dtype=np.uint8).astype(np.int32)
            f = (y + w) % 5
            left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]]) if len(cur) > bpp else np.zeros_like(cur)
            up = prev.astype(np.int32)
            upleft = np.concatenate([np.zeros(bpp, np.int32), up[:-bpp]]) if len(cur) > bpp else np.zeros_like(cur)
            if f == 0:
                enc = cur
            elif f == 1:
                enc = cur - left
            elif f == 2:
                enc = cur - up
            elif f == 3:
                enc = cur - ((left + up) >> 1)
            else:
                p = left + up - upleft
                pa, pb, pc = np.abs(p - left), np.abs(p - up), np.abs(p - upleft)
                enc = cur - np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, upleft))
            out += bytes([f]) + (enc & 255).astype(np.uint8).tobytes()
            prev = cur.astype(np.uint8)
        return bytes(out)

    if interlace:
        data = b""
        for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = a[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                data += rows_of(sub)
    else:
        data = rows_of(a)

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xffffffff)

    z = zlib.compress(data, 6)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, depth, ctype, 0, 0, 1 if interlace else 0))
    if palette is not None:
        out += chunk(b"PLTE", np.asarray(palette, dtype=np.uint8).tobytes())
    out += chunk(b"IDAT", z[: len(z) // 2]) + chunk(b"IDAT", z[len(z) // 2:]) + chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(out)


def test_interlaced_png_matches_pil(tmp_path):
    """image::open takes Adam7-interlaced PNGs.  Files written here pass by pass (every filter type, two IDAT chunks; widths and
    heights from 1 to 19 so that passes come out empty, one pixel wide or ragged; 1- to 16-bit samples, grey, RGB, RGBA, grey with
    alpha, palette) must decode to the pixels they were made from - as the same pixels stored without interlacing do - and to what PIL decodes."""
    rng = np.random.default_rng(2)
    names, expect = [], []
    for k, (W, H) in enumerate(((1, 1), (2, 3), (5, 4), (8, 8), (9, 17), (19, 7), (16, 16), (3, 1), (1, 9))):
        for depth, ctype, S in ((8, 2, 3), (8, 0, 1), (8, 6, 4), (16, 2, 3), (4, 0, 1), (1, 0, 1), (8, 4, 2), (2, 3, 1)):
            px = rng.integers(0, 1 << depth, (H, W, S))
            pal = rng.integers(0, 256, (4, 3)) if ctype == 3 else None
            # what the texels must be: 8 bits per sample (the high byte of 16, smaller depths scaled up), alpha dropped
            if ctype == 3:
                want = pal[px[:, :, 0]]
            else:
                want = px >> 8 if depth == 16 else (px if depth == 8 else px * 255 // ((1 << depth) - 1))
                want = want[:, :, :3] if ctype in (2, 6) else want[:, :, :1]
            for interlace in (1, 0):
                name = "p%d_%d_%d_%d.png" % (k, depth, ctype, interlace)
                _write_png(tmp_path / name, px, depth, ctype, interlace, pal)
                names.append(name)
                expect.append(want.astype(np.uint8))
    got = _texels_of(tmp_path, names)
    for name, g, want in zip(names, got, expect):
        assert g.shape == want.shape, name
        assert np.array_equal(g, want), "%s: %d texels differ" % (name, int((g != want).sum()))
        if "_8_2_" in name or "_8_0_" in name or "_8_6_" in name:  # and PIL reads the same from the file
            assert np.array_equal(g, sceneio.load_image(str(tmp_path / name))), name


def test_png_and_pnm_textures(tmp_path):
    from PIL import Image

    rng = np.random.default_rng(0)
    rgb = rng.integers(0, 256, (13, 17, 3), dtype=np.uint8)
    grey = rng.integers(0, 256, (9, 5), dtype=np.uint8)
    Image.fromarray(rgb).save(tmp_path / "a.png")
    Image.fromarray(grey).save(tmp_path / "g.png")
    Image.fromarray(rgb).save(tmp_path / "a.ppm")
    Image.fromarray(rgb).convert("RGBA").save(tmp_path / "rgba.png")
    scene = {"Logistics": {"resolution_x": 8, "resolution_y": 8, "samples": 1},
             "Camera": {"position": [0, 0, 1], "look_at": [0, 0, 0], "up": [0, 1, 0], "fov": 90, "aspect_ratio": 1, "aperture": 0, "focus_distance": 1},
             "Textures": {"A": {"type": "Image", "image_path": "a.png"}, "B": {"type": "Image", "image_path": "g.png"},
                          "C": {"type": "Image", "image_path": "a.ppm"}, "D": {"type": "Image", "image_path": "rgba.png"},
                          "E": {"type": "Image", "image_path": "a.png"}},
             "Materials": {"M": {"type": "Lambert", "albedo": "A"}}, "Aggregate": "BVH",
             "Shapes": [{"type": "Sphere", "radius": 0.5, "material": "M"}]}
    (tmp_path / "s.json").write_text(json.dumps(scene))
    hs = rrt.HostScene.load(str(tmp_path / "s.json"))
    tex, px = hs.textures(), hs.texels()

    def texels(i):
        t = tex[i]
        return px[t["u3"]: t["u3"] + int(t["u0"]) * int(t["u1"]) * int(t["u2"])].reshape(t["u1"], t["u0"], t["u2"])

    assert np.array_equal(texels(0), rgb) and np.array_equal(texels(1)[:, :, 0], grey)
    assert np.array_equal(texels(2), rgb) and np.array_equal(texels(3), rgb)
    assert tex[4]["u3"] == tex[0]["u3"]  # the same file is decoded once (resources.rs:34-39)


BASE = {"Logistics": {"resolution_x": 8, "resolution_y": 8, "samples": 1},
        "Camera": {"position": [0, 0, 1], "look_at": [0, 0, 0], "up": [0, 1, 0], "fov": 90, "aspect_ratio": 1, "aperture": 0, "focus_distance": 1},
        "Textures": {"T": {"type": "Constant", "color": [1, 1, 1]}}, "Materials": {"M": {"type": "Lambert", "albedo": "T"}},
        "Aggregate": "BVH", "Shapes": [{"type": "Sphere", "radius": 1, "material": "M"}]}


def load_variant(tmp_path, mutate):
    s = json.loads(json.dumps(BASE))
    mutate(s)
    p = tmp_path / "v.json"
    p.write_text(json.dumps(s))
    return rrt.HostScene.load(str(p))


@pytest.mark.parametrize("mutate,message", [
    (lambda s: s.pop("Camera"), "Required key Camera is missing."),
    (lambda s: s.pop("Aggregate"), "Required key Aggregate is missing."),
    (lambda s: s.__setitem__("Aggregate", "Grid"), "Unknown Aggregate 'type' Grid given."),
    (lambda s: s["Shapes"][0].__setitem__("material", "Nope"), "Missing Material Nope for Sphere."),
    (lambda s: s["Materials"]["M"].__setitem__("albedo", "Nope"), "Missing Texture Nope for Lambert."),
    (lambda s: s["Materials"]["M"].__setitem__("bump_map", "Nope"), "Missing bump map Texture Nope for Lambert."),
    (lambda s: s["Materials"]["M"].__setitem__("type", "Phong"), "Unsupported material type: Phong"),
    (lambda s: s["Textures"]["T"].__setitem__("type", "Marble"), "Unsupported texture type: Marble"),
    (lambda s: s["Shapes"][0].__setitem__("type", "Cylinder"), "Unknown Shape 'type' Cylinder given."),
    (lambda s: s["Textures"].__setitem__("U", {"type": "Turbulence", "scale": 1, "depth": 2, "omega": 1.5}), "Turbulence omega is greater than 1."),
    (lambda s: s["Camera"].__setitem__("position", [0, 0]), "Deserializing in to Point3 requires a Vec of length 3!"),
    (lambda s: s["Shapes"][0].__setitem__("transform", {"scale": [0, 1, 1]}), "Singular matrix"),
    (lambda s: s.__setitem__("Shapes", {}), "'Shapes' is not a JSON array."),
])
def test_loader_errors_mirror_the_reference(tmp_path, mutate, message):
    with pytest.raises(rrt.RrtError) as e:
        load_variant(tmp_path, mutate)
    assert e.value.code == 2 and message in e.value.message
    # the test-side reader refuses the same files
    s = json.loads(json.dumps(BASE))
    mutate(s)
    (tmp_path / "w.json").write_text(json.dumps(s))
    with pytest.raises(Exception):
        orc.OracleScene.from_file(str(tmp_path / "w.json"))


def test_mesh_rules(tmp_path):
    # multi-object .obj with per-object indices, vt indices, a quad (fan-triangulated) and a degenerate triangle error
    (tmp_path / "m.obj").write_text("o A\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nf 1/1 2/2 3/3 4/4\n"
                                    "o B\nv 0 0 1\nv 1 0 1\nv 0 1 1\nf 5 6 7\n")

    def m(s):
        s["Shapes"] = [{"type": "Mesh", "file_path": "m.obj", "enable_backface_culling": False, "material": "M"}]

    hs = load_variant(tmp_path, m)
    slots = hs.slots()
    assert hs.desc.n_slots == 3
    o = orc.OracleScene.from_file(str(tmp_path / "v.json"))
    assert np.array_equal(slots["prim"].astype(np.int32), o.bvh()[2])
    (tmp_path / "m.obj").write_text("o A\nv 0 0 0\nv 1 0 0\nv 2 0 0\nf 1 2 3\n")
    with pytest.raises(rrt.RrtError) as e:
        load_variant(tmp_path, m)
    assert "Error creating Triangle for file m.obj, object A: Triangle texture coordinates are degenerate." in e.value.message
    (tmp_path / "m.obj").write_text("o A\nv 0 0 0\nv 1 0 0\nv 2 1 0\nf 1 2 9\n")
    with pytest.raises(rrt.RrtError) as e:
        load_variant(tmp_path, m)
    assert "Triangle mesh has length 3 but attempted to make a Triangle with indices 0, 1, 8." in e.value.message


def test_a_medium_may_be_the_boundary_of_a_medium(tmp_path):
    """scene.rs:588 hands the boundary to deserialize_shape, so a ConstantMedium may bound a ConstantMedium.  The flattened
    scene holds one record per medium, the outer one pointing at the inner; box, light sampling and the scene's shape count are
    those of the innermost boundary (volume.rs:104-122)."""
    (tmp_path / "t.obj").write_text("o T\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    (tmp_path / "q.obj").write_text("o Q\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 4\n")
    ball = {"type": "Sphere", "radius": 0.5, "material": "M", "transform": {"translate": [2, 0, 0]}}
    tri = {"type": "Mesh", "file_path": "t.obj", "enable_backface_culling": False, "material": "M"}

    def medium(boundary, density, phase="I"):
        return {"type": "ConstantMedium", "boundary": boundary, "density": density, "phase_func": phase}

    def m(s):
        s["Materials"]["I"] = {"type": "Isotropic", "albedo": "T"}
        s["Materials"]["G"] = {"type": "Dielectric", "refractive_index": 1.5}
        s["Shapes"] = [medium(medium(ball, 1.0), 2.0), {"type": "Sphere", "radius": 1, "material": "M"},
                       medium(medium(medium(tri, 3.0), 4.0), 5.0, "G")]

    hs = load_variant(tmp_path, m)
    assert hs.desc.n_slots == 3 and hs.desc.n_spheres == 2 and hs.desc.n_boundary_tris == 1
    media = hs.media()
    assert [(int(r["boundary_kind"]), int(r["boundary_index"]), float(r["density"])) for r in media] == \
        [(1, 0, 1.0), (2, 0, 2.0), (0, 0, 3.0), (2, 2, 4.0), (2, 3, 5.0)]
    slots = hs.slots()
    by_prim = {int(r["prim"]): r for r in slots}
    assert by_prim[0]["info"] >> 30 == 2 and by_prim[0]["aux"] == 1 and by_prim[2]["info"] >> 30 == 2 and by_prim[2]["aux"] == 4
    lights = hs.lights()  # the glass-phased medium is "important" (scene.rs:131-140); its pdf is the triangle's
    assert len(lights) == 1 and lights[0]["kind"] == 0 and np.array_equal(lights[0]["v1"], [1, 0, 0])
    o = orc.OracleScene.from_file(str(tmp_path / "v.json"))
    assert o.num_shapes() == 3 and list(o.important()) == [2]
    assert np.array_equal(slots["prim"].astype(np.int32), o.bvh()[2])
    assert np.array_equal(bits(hs.nodes()["bmin"]), bits(o.bvh()[0][:, :3])) and np.array_equal(bits(hs.nodes()["bmax"]), bits(o.bvh()[0][:, 3:]))

    def too_deep(s):
        m(s)
        s["Shapes"] = [medium(medium(medium(medium(ball, 1.0), 2.0), 3.0), 4.0)]

    with pytest.raises(rrt.RrtError) as e:
        load_variant(tmp_path, too_deep)
    assert e.value.code == 2 and "this build follows 3 of them at most" in e.value.message

    def two_shapes(s):
        m(s)
        s["Shapes"] = [medium(medium({"type": "Mesh", "file_path": "q.obj", "enable_backface_culling": False, "material": "M"}, 1.0), 2.0)]

    with pytest.raises(rrt.RrtError) as e:
        load_variant(tmp_path, two_shapes)
    assert "Only single shapes are allowed for constant mediums at the moment." in e.value.message

    def no_phase(s):
        m(s)
        s["Shapes"] = [medium(medium(ball, 1.0, "Nope"), 2.0)]

    with pytest.raises(rrt.RrtError) as e:
        load_variant(tmp_path, no_phase)
    assert "Missing Phase Function Nope for ConstantMedium." in e.value.message


def test_builder_api_and_list_aggregate():
    hs = rrt.HostScene.new()
    t = hs.add_texture(rrt.lib() and 0, 0.5, 0.5, 0.5)
    light_t = hs.add_texture(0, 4.0, 4.0, 4.0)
    m = hs.add_material(0, t)
    lm = hs.add_material(3, light_t)
    hs.add_sphere(1.0, m, translate=(0, 0, -3))
    hs.add_mesh([[-1, 2, -4], [1, 2, -4], [1, 2, -2], [-1, 2, -2]], [[0, 1, 2], [0, 2, 3]], lm)
    hs.set_camera((0, 0, 1), (0, 0, 0), (0, 1, 0), 60, 1.0, 0.0, 1.0)
    hs.set_logistics(16, 16, 2)
    hs.finalize("List")
    d = hs.desc
    assert d.aggregate_kind == 0 and d.n_nodes == 1 and d.n_slots == 3 and d.n_lights == 2
    assert list(hs.slots()["prim"]) == [0, 1, 2]  # scene order
    with pytest.raises(rrt.RrtError):
        hs.add_material(0, 99)


def test_device_built_aggregates_fail_loudly_without_a_gpu():
    """RRT_AGGREGATE_BVH_GPU / _BVH_CLEAN are built on the device; there is no host fallback behind them."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import synth

    d = synth.slats_desc(n=3, aggregate=1)
    for kind in ("BVH_GPU", "BVH_CLEAN"):
        with pytest.raises(rrt.RrtError, match="BVH build"):
            synth.host_scene(d, kind)
    hs = rrt.HostScene.new()
    assert rrt.lib().rrt_host_finalize(hs._h, 7) != 0  # unknown aggregate kind


def test_synthetic_mesh_bvh_matches_oracle_at_scale():
    """A 20k-triangle soup: the product's in-place range builder and the oracle's recursive builder give the same tree."""
    rng = np.random.default_rng(1)
    n = 20000
    c = rng.uniform(-1, 1, (n, 1, 3))
    v = (c + rng.normal(scale=0.01, size=(n, 3, 3))).reshape(-1, 3)
    idx = np.arange(3 * n, dtype=np.int32).reshape(n, 3)
    hs = rrt.HostScene.new()
    t = hs.add_texture(0, 0.5, 0.5, 0.5)
    m = hs.add_material(0, t)
    hs.add_mesh(v, idx, m)
    hs.set_camera((0, 0, 3), (0, 0, 0), (0, 1, 0), 60, 1.0, 0.0, 1.0)
    hs.set_logistics(16, 16, 1)
    hs.finalize("BVH")
    L = orc.lib()
    h = L.orc_scene_new()
    L.orc_add_texture_constant(h, 0.5, 0.5, 0.5)
    L.orc_add_material(h, 0, 0, -1, 0.0)
    z, one = orc.a32([0, 0, 0]), orc.a32([1, 1, 1])
    assert L.orc_add_mesh(h, z, z, one, len(v), np.ascontiguousarray(v).reshape(-1), 0, np.zeros(2), n, idx.reshape(-1), np.full(3 * n, -1, np.int32), 0, 0, 0) == n
    L.orc_set_logistics(h, 16, 16, 1, 1)
    L.orc_finalize(h, 1)
    nn = L.orc_bvh_num_nodes(h)
    meta = np.zeros((nn, 4), dtype=np.int32)
    boxes = np.zeros((nn, 6), dtype=np.float32)
    leaf = np.zeros(n, dtype=np.int32)
    L.orc_bvh_export(h, boxes.ctypes.data, meta.ctypes.data, leaf.ctypes.data)
    L.orc_scene_free(h)
    nodes = hs.nodes()
    assert len(nodes) == nn and np.array_equal(hs.slots()["prim"].astype(np.int32), leaf)
    assert np.array_equal(nodes["bmin"], boxes[:, :3]) and np.array_equal(nodes["bmax"], boxes[:, 3:])
    assert nn > 1000


def test_image_decoders_survive_hostile_files_under_asan(tmp_path):
    """Truncated files, segment lengths too short for their fields, Huffman / quantisation selectors out of range and seeded
    random byte flips: the JPEG and PNG decoders, built here with AddressSanitizer and UBSan (host code only), must refuse
    or decode every file without touching memory out of bounds (tests/fuzz_image_decoders.cpp)."""
    import shutil
    import subprocess

    from PIL import Image

    if not shutil.which("g++"):
        pytest.skip("no host compiler")
    rng = np.random.default_rng(0)
    Image.fromarray(rng.integers(0, 256, (40, 56, 3), dtype=np.uint8)).save(tmp_path / "sub.jpg", quality=80, subsampling=2)
    Image.fromarray(rng.integers(0, 256, (24, 40, 3), dtype=np.uint8)).save(tmp_path / "c444.jpg", quality=85, subsampling=0)
    Image.fromarray(rng.integers(0, 256, (24, 40), dtype=np.uint8)).save(tmp_path / "grey.jpg", quality=90)
    Image.fromarray(rng.integers(0, 256, (40, 56, 3), dtype=np.uint8)).save(tmp_path / "a.png")
    exe = str(tmp_path / "fuzz_image_decoders")
    subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "fuzz_image_decoders.cpp"),
                           os.path.join(ROOT, "rust_raytracer_amd", "csrc", "host", "image_codecs.cpp"), "-lz", "-o", exe])
    env = dict(os.environ, ASAN_OPTIONS="verify_asan_link_order=0:detect_leaks=0")
    Image.fromarray(rng.integers(0, 256, (40, 56, 3), dtype=np.uint8)).save(tmp_path / "prog.jpg", quality=85, subsampling=0, progressive=True)
    smooth = np.clip(np.add.outer(np.arange(33) * 5.0, np.arange(47) * 3.0) + rng.normal(0, 12, (33, 47)), 0, 255).astype(np.uint8)
    Image.fromarray(smooth).save(tmp_path / "prog_grey.jpg", quality=70, progressive=True, optimize=True)
    Image.fromarray(rng.integers(0, 256, (40, 56, 3), dtype=np.uint8)).save(tmp_path / "prog_sub.jpg", quality=80, subsampling=2, progressive=True)
    _write_png(tmp_path / "adam7.png", rng.integers(0, 256, (21, 19, 3)), 8, 2, 1)
    _write_png(tmp_path / "adam7_bits.png", rng.integers(0, 4, (9, 13, 1)), 2, 0, 1)
    files = ("sub.jpg", "c444.jpg", "grey.jpg", "a.png", "prog.jpg", "prog_grey.jpg", "prog_sub.jpg", "adam7.png", "adam7_bits.png")
    r = subprocess.run([exe] + [str(tmp_path / n) for n in files], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "fuzz ok" in r.stdout


def test_scene_loader_survives_hostile_files_under_asan(tmp_path):
    """A scene file and its .obj file cut short at every length, with every number replaced by hostile spellings (huge, negative,
    not a number, another JSON type), with seeded random damage and with hundreds of thousands of opening brackets: the loader,
    built here from the library's host sources with AddressSanitizer and UBSan, must refuse or load every file without touching
    memory out of bounds, and what it loads must hold only indices in range (tests/fuzz_scene_loader.cpp)."""
    import shutil
    import subprocess

    from PIL import Image

    if not shutil.which("g++"):
        pytest.skip("no host compiler")
    rng = np.random.default_rng(0)
    Image.fromarray(rng.integers(0, 256, (8, 12, 3), dtype=np.uint8)).save(tmp_path / "a.png")
    (tmp_path / "m.obj").write_text("o A\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nf 1/1 2/2 3/3 4/4\n"
                                    "o B\nv 0 0 1\nv 1 0 1\nv 0 1 1\nf 5 6 7\nf -3 -2 -1\n")
    (tmp_path / "t.obj").write_text("o T\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    ball = {"type": "Sphere", "radius": 0.5, "material": "M", "transform": {"translate": [2, 0, 0], "rotate": [10, 20, 30], "scale": [1, 2, 1]}}

    def medium(boundary, density):
        return {"type": "ConstantMedium", "boundary": boundary, "density": density, "phase_func": "I"}

    scene = {
        "Logistics": {"resolution_x": 8, "resolution_y": 8, "samples": 1, "use_importance_sampling": True},
        "Camera": {"position": [0, 0, 1], "look_at": [0, 0, 0], "up": [0, 1, 0], "fov": 90, "aspect_ratio": 1, "aperture": 0.1, "focus_distance": 1},
        "Textures": {"T": {"type": "Constant", "color": [1, 1, 1]}, "P": {"type": "Image", "image_path": "a.png"}, "N": {"type": "Noise", "scale": 2.0},
                     "U": {"type": "Turbulence", "scale": 1.0, "depth": 3, "omega": 0.5},
                     "C": {"type": "Checker", "repeat": 4.0, "odd": {"type": "Constant", "color": [1, 1, 1]},
                           "even": {"type": "Checker", "repeat": 2, "odd": {"type": "Image", "image_path": "a.png"}, "even": {"type": "Test"}}}},
        "Materials": {"M": {"type": "Lambert", "albedo": "C", "bump_map": "N"}, "I": {"type": "Isotropic", "albedo": "U"},
                      "G": {"type": "Dielectric", "refractive_index": 1.5}, "L": {"type": "DiffuseLight", "emission": "T"},
                      "K": {"type": "Metal", "albedo": "P", "roughness": 0.2}},
        "Aggregate": "BVH",
        "Shapes": [ball, {"type": "Sphere", "radius": 1, "material": "G"},
                   {"type": "Mesh", "file_path": "m.obj", "enable_backface_culling": False, "material": "L", "transform": {"translate": [0, 2, 0]}},
                   medium(medium(dict(ball), 1.0), 2.0),
                   medium({"type": "Mesh", "file_path": "t.obj", "enable_backface_culling": True, "material": "K"}, 0.5)]}
    (tmp_path / "s.json").write_text(json.dumps(scene, indent=1))
    exe = str(tmp_path / "fuzz_scene_loader")
    host = os.path.join(ROOT, "rust_raytracer_amd", "csrc", "host")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "fuzz_scene_loader.cpp")]
                          + [os.path.join(host, f) for f in ("scene_json.cpp", "scene_host.cpp", "image_io.cpp", "image_codecs.cpp")]
                          + [os.path.join(ROOT, "rust_raytracer_amd", "csrc", "rrt_error.cpp"), "-lz", "-o", exe])
    env = dict(os.environ, ASAN_OPTIONS="verify_asan_link_order=0:detect_leaks=0")
    r = subprocess.run([exe, str(tmp_path / "s.json"), str(tmp_path / "m.obj")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "fuzz ok" in r.stdout


def test_scene_description_whose_links_are_no_tree_is_refused():
    """rrt_scene_create validates a description before it touches the device (so this needs no GPU): a node with two parents,
    a node no parent names, a child index out of range and a slot range past the end are all refused with a message."""
    import synth

    hs = synth.host_scene(synth.terrain_desc(12, seed=1))
    base = hs.desc
    nodes = hs.nodes()
    inner = np.nonzero((nodes["b"] & 0x80000000) == 0)[0]
    assert len(inner) >= 3

    def create_with(mutated):
        buf = (rrt._Node * len(mutated)).from_buffer_copy(mutated.tobytes())
        d = rrt._SceneDesc()
        C.memmove(C.byref(d), C.byref(base), C.sizeof(rrt._SceneDesc))
        d.nodes = C.cast(buf, C.POINTER(rrt._Node))
        h = C.c_void_p()
        rc = rrt.lib().rrt_scene_create(C.byref(d), C.byref(h))
        return rc, rrt.lib().rrt_last_error().decode()

    # the right child of the root re-used as the right child of the root's left child: two parents (and one node orphaned)
    m = nodes.copy()
    left = 1
    assert (m["b"][left] & 0x80000000) == 0
    m["a"][left] = m["a"][0]
    rc, msg = create_with(m)
    assert rc == 1 and ("two nodes" in msg or "not reachable" in msg), msg
    m = nodes.copy()
    m["a"][0] = len(nodes) + 5
    rc, msg = create_with(m)
    assert rc == 1 and "out of bounds" in msg
    m = nodes.copy()
    leaf = np.nonzero((nodes["b"] & 0x80000000) != 0)[0][-1]
    m["a"][leaf] = base.n_slots
    rc, msg = create_with(m)
    assert rc == 1 and "slot range" in msg


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_threaded_scene_preparation_is_race_free_and_equals_one_thread(tmp_path, sanitizer):
    """Large trees are prepared for upload by several threads (rrt_scene_prepare.cpp).  tests/check_prepare_threads.cpp prepares a
    131 071-node tree with one thread and with many, under ThreadSanitizer and under AddressSanitizer + UBSan: no report, and
    the two results are the same bytes."""
    import subprocess

    exe = str(tmp_path / "check_prepare_threads")
    csrc = os.path.join(ROOT, "rust_raytracer_amd", "csrc")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=" + sanitizer, "-fno-sanitize-recover=undefined", "-D__HIP_PLATFORM_AMD__",
                           "-I", "/opt/rocm/include", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "check_prepare_threads.cpp"),
                           os.path.join(csrc, "rrt_scene_prepare.cpp"), os.path.join(csrc, "rrt_error.cpp"), "-lpthread", "-o", exe])
    env = dict(os.environ, ASAN_OPTIONS="verify_asan_link_order=0:detect_leaks=0", TSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "threads ok" in r.stdout and "WARNING" not in r.stderr


def test_what_a_scene_would_upload_is_worked_out_without_a_device():
    """rrt_scene_check runs the CPU half of rrt_scene_create alone: the records, the kernel variant and the traversal choices of the
    shipped scenes and of synthetic ones, with no GPU in the box."""
    import orc
    import synth

    F_SPHERE, F_MEDIUM, F_TEX, F_BUMP = 1, 2, 4, 8
    c = rrt.HostScene.load(orc.asset("Cornell_Box_With_Sphere.json")).check()
    # one leaf (the reference's sweep does not split it), its lights among its primitives: the harvesting query, one stack entry
    assert c["node_records"] == 1 and c["harvest"] and not c["deep"] and not c["clean"] and c["stack_depth"] == 1
    assert c["features"] == F_SPHERE and c["plain_records"] > 0 and c["guard_entries"] >= 5
    # the sphere gets a record of its own and is tested directly; the triangles share records two by two, and the sifted records are even
    hs = rrt.HostScene.load(orc.asset("Cornell_Box_With_Sphere.json"))
    n_tri = int((hs.slots()["info"] >> 30 == 0).sum())
    assert (n_tri + 1) // 2 + 1 <= c["pair_records"] <= n_tri + 2 and (c["pair_records"] - 1) % 2 == 0
    g = rrt.HostScene.load(orc.asset("Globe.json")).check()
    assert g["features"] & F_TEX and g["features"] & F_BUMP and g["plain_records"] == 0
    e = rrt.HostScene.load(orc.asset("Example_Spec.json")).check()
    assert e["features"] & F_MEDIUM and not e["harvest"] and not e["deep"]
    assert len({c["desc_hash"], g["desc_hash"], e["desc_hash"]}) == 3
    # a mesh: the reference's tree is deep (the streaming kernels, no guard table); as a List it is one leaf of many pair records
    t = synth.host_scene(synth.terrain_desc(24, seed=2))
    d = t.check()
    assert d["deep"] and d["guard_entries"] == 0 and d["node_records"] == t.desc.n_nodes and d["stack_depth"] <= t.desc.bvh_depth
    assert d["pair_records"] >= (t.desc.n_slots + 1) // 2
    # the same description twice: the same hash; a NULL layout only checks
    assert synth.host_scene(synth.terrain_desc(24, seed=2)).check()["desc_hash"] == d["desc_hash"]
    assert rrt.lib().rrt_scene_check(C.byref(t.desc), None) == 0


def test_rust_bindings_agree_with_the_header_and_the_python_bindings():
    """rrt-sys/src/lib.rs cannot be compiled here (no Rust toolchain), so it is read as text: every function the header declares
    must be declared there with the same number of parameters, and every struct's size and field offsets written there must be the
    ones ctypes computes for the Python mirror of the same struct (the C side holds the same numbers in static_asserts)."""
    header = open(os.path.join(ROOT, "include", "rrt.h")).read()
    rust = open(os.path.join(ROOT, "rrt-sys", "src", "lib.rs")).read()
    header_nc = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    decl = {m.group(1): m.group(2) for m in re.finditer(r"\b(rrt_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", header_nc)}
    assert len(decl) >= 30
    rust_nc = re.sub(r"//[^\n]*", "", rust)
    rdecl = {m.group(1): m.group(2) for m in re.finditer(r"pub fn (rrt_[a-z0-9_]+)\s*\(([^)]*)\)", rust_nc)}
    for name, args in decl.items():
        assert name in rdecl, "rrt-sys does not declare %s" % name
        n_c = 0 if args.strip() in ("", "void") else args.count(",") + 1
        n_r = 0 if not rdecl[name].strip() else rdecl[name].count(":")
        assert n_c == n_r, "%s: %d parameters in rrt.h, %d in rrt-sys" % (name, n_c, n_r)
    mirrors = {"rrt_node": rrt._Node, "rrt_slot": rrt._Slot, "rrt_tri_shade": rrt._TriShade, "rrt_sphere": rrt._Sphere, "rrt_medium": rrt._Medium,
               "rrt_material": rrt._Material, "rrt_texture": rrt._Texture, "rrt_light": rrt._Light, "rrt_camera": rrt._Camera,
               "rrt_scene_desc": rrt._SceneDesc, "rrt_render_params": rrt._RenderParams, "rrt_stats": rrt._Stats}
    for name, cls in mirrors.items():
        body = re.search(r"pub struct %s \{(.*?)\n\}" % name, rust, flags=re.S).group(1)
        fields = re.findall(r"pub (\w+):[^,]*,\s*// offset (\d+)", body)
        assert [f.rstrip("_") for f, _ in fields] == [f[0] for f in cls._fields_], name
        for (fname, off), cf in zip(fields, cls._fields_):
            assert int(off) == getattr(cls, cf[0]).offset, "%s.%s" % (name, fname)
        size = int(re.search(r"size_of::<%s>\(\) == (\d+)" % name, rust).group(1))
        assert size == C.sizeof(cls), name
    layout = open(os.path.join(ROOT, "rust_raytracer_amd", "csrc", "rrt_abi_layout.cpp")).read()
    for name, cls in mirrors.items():
        assert "RRT_SIZE(%s, %d)" % (name, C.sizeof(cls)) in layout
        for cf in cls._fields_:
            assert "RRT_AT(%s, %s, %d)" % (name, cf[0], getattr(cls, cf[0]).offset) in layout, (name, cf[0])
