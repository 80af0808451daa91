"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bars: primitive id and hit/miss bit-exact; hit distance bit-exact (tolerance 0; the stated allowance is
1e-5 relative); hit properties, camera rays, texture values, elementary functions and random streams
bit-exact; rendered radiance sums bit-exact against the oracle's loop-form trace at equal seed.
"""
import numpy as np
import pytest

import orc
import rays as raygen
import rust_raytracer_amd as rrt
import synth

pytestmark = pytest.mark.gpu

SCENES = ["Cornell_Box.json", "Cornell_Box_With_Sphere.json", "Materials.json", "Example_Spec.json", "Globe.json"]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def both(name, **ov):
    o = orc.OracleScene.from_file(orc.asset(name), {k: v for k, v in ov.items()})
    g = rrt.Scene.load(orc.asset(name), width=ov.get("resolution_x", 0), height=ov.get("resolution_y", 0), samples=ov.get("samples", 0))
    return o, g


@pytest.mark.parametrize("name", SCENES)
def test_intersect_batch_bit_exact(gpu, name):
    o, g = both(name, resolution_x=64, resolution_y=64)
    ro, rd = raygen.ray_batch(o, 64, 64, n_random=4096, seed=7)
    p_ref, t_ref = o.intersect(ro, rd)
    for mode in (rrt.AABB_TIGHT, rrt.AABB_REFERENCE):
        p, t = g.hit(ro, rd, aabb_mode=mode)
        assert np.array_equal(p, p_ref), "primitive ids differ in mode %d: %d rays" % (mode, (p != p_ref).sum())
        assert np.array_equal(bits(t), bits(t_ref)), "hit distances differ in mode %d" % mode
    assert (p_ref >= 0).sum() > 100


@pytest.mark.parametrize("name", SCENES)
def test_hit_properties_bit_exact(gpu, name):
    o, g = both(name, resolution_x=64, resolution_y=64)
    ro, rd = raygen.ray_batch(o, 64, 64, n_random=2048, seed=3)
    p, t = o.intersect(ro, rd)
    ref = o.hit_properties(ro, rd, p, t)
    inp = np.zeros((len(ro), 8), dtype=np.uint32)
    inp[:, 0:3] = bits(ro)
    inp[:, 3:6] = bits(rd)
    inp[:, 6] = p.astype(np.uint32)
    inp[:, 7] = bits(t)
    got = g.probe(1, 0, inp, 14)
    assert np.array_equal(bits(got), bits(ref))


@pytest.mark.parametrize("name", ["Cornell_Box_With_Sphere.json", "Materials.json"])
def test_camera_rays_bit_exact(gpu, name):
    o, g = both(name, resolution_x=96, resolution_y=64)
    rng = np.random.default_rng(1)
    n = 5000
    px = rng.integers(0, 96, n).astype(np.uint32)
    py = rng.integers(0, 64, n).astype(np.uint32)
    sm = rng.integers(0, 4096, n).astype(np.uint32)
    ro, rd = o.camera_rays(96, 64, 11, px, py, sm)
    got = g.probe(0, 11, np.stack([px, py, sm], axis=1), 6)
    assert np.array_equal(bits(got[:, :3]), bits(ro)) and np.array_equal(bits(got[:, 3:]), bits(rd))


def test_elementary_functions_and_rng_bit_exact(gpu):
    o, g = both("Cornell_Box.json")
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-7, 7, 20000), rng.uniform(-1, 1, 20000), rng.uniform(-5000, 5000, 5000), rng.integers(0, 1 << 24, 20000) / 16777216.0,
                        [0.0, 1.0, -1.0, 0.5, -0.5, 1e-30, 3.14159274, 1.57079637]]).astype(np.float32)
    y = rng.uniform(-3, 3, len(x)).astype(np.float32)
    y[:100] = 0.0
    got = g.probe(3, 0, np.stack([bits(x), bits(y)], axis=1), 5)
    L = orc.lib()
    for col, which in enumerate([0, 1, 2, 3, 4]):
        ref = np.zeros(len(x), dtype=np.float32)
        L.orc_unit_detmath(which, len(x), x, y, ref)
        assert np.array_equal(bits(got[:, col]), bits(ref)), "function %d differs" % which
    pix = rng.integers(0, 1 << 31, 3000).astype(np.uint32)
    sm = rng.integers(0, 1 << 16, 3000).astype(np.uint32)
    got = g.probe(4, 99, np.stack([pix, sm], axis=1), 8)
    ref = np.zeros(8, dtype=np.float32)
    for i in range(0, 3000, 37):
        L.orc_unit_rng(99, int(pix[i]), int(sm[i]), 8, ref, 0, None, None)
        assert np.array_equal(bits(got[i]), bits(ref))


@pytest.mark.parametrize("name", ["Materials.json", "Example_Spec.json", "Globe.json"])
def test_texture_values_bit_exact(gpu, name):
    o, g = both(name)
    rng = np.random.default_rng(2)
    n = 4000
    uv = rng.uniform(0, 1, (n, 2)).astype(np.float32)
    p = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    for tex in range(len(o.desc["textures"])):
        ref = o.texture_values(tex, uv, p)
        got = g.probe(2, tex, np.concatenate([bits(uv), bits(p)], axis=1), 3)
        assert np.array_equal(bits(got), bits(ref)), "texture %d differs" % tex


RENDER_CASES = [
    ("Cornell_Box.json", 48, 48, 8, rrt.BG_BLACK),
    ("Cornell_Box_With_Sphere.json", 64, 48, 8, rrt.BG_BLACK),
    ("Materials.json", 64, 36, 8, rrt.BG_SKY),
    ("Example_Spec.json", 40, 30, 6, rrt.BG_SKY),
    ("Globe.json", 48, 48, 6, rrt.BG_SKY),
]


@pytest.mark.parametrize("name,W,H,spp,bg", RENDER_CASES)
def test_render_bit_exact(gpu, name, W, H, spp, bg):
    o, g = both(name, resolution_x=W, resolution_y=H, samples=spp)
    ref, st_ref = o.render(W, H, spp, seed=5, bg=bg)
    for mode in (rrt.AABB_TIGHT, rrt.AABB_REFERENCE):
        img, st = g.render(seed=5, background=bg, aabb_mode=mode)
        same = (bits(img) == bits(ref)) | (np.isnan(img) & np.isnan(ref))
        assert same.all(), "%d of %d values differ (mode %d)" % ((~same).sum(), same.size, mode)
        assert st["rays"] == int(st_ref[0]) and st["paths"] == W * H * spp


@pytest.mark.parametrize("name,bg", [("Cornell_Box_With_Sphere.json", rrt.BG_BLACK), ("Cornell_Box.json", rrt.BG_BLACK), ("Example_Spec.json", rrt.BG_SKY)])
def test_render_bit_exact_on_a_million_paths(gpu, name, bg):
    """A larger sample of paths (rare events: NaN pixels of 0 * L / 0, grazing self-hits, rays that leave through the open
    front after many bounces) - still bit for bit."""
    W, H, spp = 160, 128, 48
    o, g = both(name, resolution_x=W, resolution_y=H, samples=spp)
    ref, st_ref = o.render(W, H, spp, seed=17, bg=bg, threads=min(16, len(__import__("os").sched_getaffinity(0))))
    img, st = g.render(seed=17, background=bg)
    same = (bits(img) == bits(ref)) | (np.isnan(img) & np.isnan(ref))
    assert same.all(), "%d of %d values differ" % ((~same).sum(), same.size)
    assert st["rays"] == int(st_ref[0]) and st["paths"] == W * H * spp


def test_large_odd_frame_and_ragged_batches(gpu):
    """A frame of 1003 x 701 pixels (no multiple of the 8x8 wave tile or of the 32x32 sharding tile) of the textured, bump-mapped
    globe; ray batches of 0, 1, 63, 64, 65 and 1000 rays."""
    W, H, spp = 1003, 701, 2
    o, g = both("Globe.json", resolution_x=W, resolution_y=H, samples=spp)
    ref, st_ref = o.render(W, H, spp, seed=23, bg=rrt.BG_SKY, threads=min(16, len(__import__("os").sched_getaffinity(0))))
    img, st = g.render(seed=23, background=rrt.BG_SKY)
    assert np.array_equal(bits(img), bits(ref)) and st["rays"] == int(st_ref[0]) and st["paths"] == W * H * spp
    ro, rd = raygen.ray_batch(o, 64, 64, n_random=300, seed=1)
    for n in (0, 1, 63, 64, 65, 1000):
        p_ref, t_ref = o.intersect(ro[:n], rd[:n]) if n else (np.zeros(0, np.int32), np.zeros(0, np.float32))
        p, t = g.hit(ro[:n], rd[:n])
        assert np.array_equal(p, p_ref) and np.array_equal(bits(t), bits(t_ref))


def test_render_sample_ranges_and_tiles(gpu):
    """first_sample continues the per-pixel stream; interleaved tiles reproduce the full frame bit for bit."""
    o, g = both("Cornell_Box_With_Sphere.json", resolution_x=80, resolution_y=72, samples=8)
    full, _ = g.render(seed=3)
    a, _ = g.render(samples=3, first_sample=0, seed=3)
    b, _ = g.render(samples=5, first_sample=3, seed=3)
    ra, _ = o.render(80, 72, 3, first_sample=0, seed=3)
    rb, _ = o.render(80, 72, 5, first_sample=3, seed=3)
    assert np.array_equal(bits(a), bits(ra)) and np.array_equal(bits(b), bits(rb))
    tiled = np.zeros_like(full)
    for k in range(3):
        g.render(seed=3, tile_count=3, tile_index=k, out=tiled)
    assert np.array_equal(bits(tiled), bits(full))


@pytest.mark.parametrize("cull", [False, True, "soup"])
def test_mesh_with_a_deep_tree_bit_exact(gpu, cull):
    """A 20 000-triangle displaced grid (7 000-node tree, leaves of up to 54 triangles, lanes in different leaves), and a soup
    of 3 000 triangles of very different sizes: closest hits in both traversal modes and a small render against the oracle's
    reference-mode traversal."""
    d = synth.soup_desc(width=40, height=40) if cull == "soup" else synth.terrain_desc(100, seed=2, width=40, height=40, samples=2, cull=cull)
    o = orc.OracleScene(d)
    g = rrt.Scene(synth.host_scene(d))
    ro, rd = raygen.ray_batch(o, 40, 40, n_random=6000, seed=9)
    p_ref, t_ref = o.intersect(ro, rd)
    assert (p_ref >= 0).sum() > 2000
    for mode in (rrt.AABB_TIGHT, rrt.AABB_REFERENCE):
        p, t = g.hit(ro, rd, aabb_mode=mode)
        assert np.array_equal(p, p_ref), "primitive ids differ in mode %d: %d rays" % (mode, (p != p_ref).sum())
        assert np.array_equal(bits(t), bits(t_ref))
    ref, st_ref = o.render(40, 40, 2, seed=1)
    for mode in (rrt.AABB_TIGHT, rrt.AABB_REFERENCE):
        img, st = g.render(seed=1, aabb_mode=mode)
        same = (bits(img) == bits(ref)) | (np.isnan(img) & np.isnan(ref))
        assert same.all(), "%d of %d values differ (mode %d)" % ((~same).sum(), same.size, mode)
        assert st["rays"] == int(st_ref[0])
    if cull is False:
        # a batch large enough for the ray-batch kernel to give every wave a run of several rays per lane (lanes fetch the
        # next ray when their query ends), with a ragged end: the same rays over and over must give the same answers
        reps = 2_200_000 // len(ro) + 1
        big_o, big_d = np.tile(ro, (reps, 1))[:-37], np.tile(rd, (reps, 1))[:-37]
        p, t = g.hit(big_o, big_d)
        assert np.array_equal(p, np.tile(p_ref, reps)[:-37]) and np.array_equal(bits(t), bits(np.tile(t_ref, reps)[:-37]))


@pytest.mark.parametrize("media", [0, 6])
def test_scaled_spheres_outside_their_reference_boxes_bit_exact(gpu, media):
    """Sphere::get_bounding_box (shape.rs:129-139) ignores the scale, so scaled spheres stick out of their boxes and the
    reference misses part of them; both traversal modes must reproduce exactly those hits and misses.  With media the same
    tree also holds ConstantMedium shapes, whose hit distance is a keyed random draw."""
    d = synth.spheres_desc(media=media)
    o = orc.OracleScene(d)
    g = rrt.Scene(synth.host_scene(d))
    ro, rd = raygen.ray_batch(o, 48, 40, n_random=8000, seed=4)
    p_ref, t_ref = o.intersect(ro, rd)
    p_list, _ = orc.OracleScene(dict(d, aggregate=0)).intersect(ro, rd)
    assert (p_ref != p_list).sum() > 10, "the scene should make the BVH lose hits a plain list finds"
    for mode in (rrt.AABB_TIGHT, rrt.AABB_REFERENCE):
        p, t = g.hit(ro, rd, aabb_mode=mode)
        assert np.array_equal(p, p_ref), "primitive ids differ in mode %d: %d rays" % (mode, (p != p_ref).sum())
        assert np.array_equal(bits(t), bits(t_ref))
    ref, st_ref = o.render(48, 40, 4, seed=2, bg=rrt.BG_SKY)
    for mode in (rrt.AABB_TIGHT, rrt.AABB_REFERENCE):
        img, st = g.render(seed=2, background=rrt.BG_SKY, aabb_mode=mode)
        same = (bits(img) == bits(ref)) | (np.isnan(img) & np.isnan(ref))
        assert same.all(), "%d of %d values differ (mode %d)" % ((~same).sum(), same.size, mode)
        assert st["rays"] == int(st_ref[0])


@pytest.mark.parametrize("n,aggregate", [(40, 1), (4, 1), (6, 0)])
def test_media_bounded_by_media_bit_exact(gpu, n, aggregate):
    """A ConstantMedium may bound a ConstantMedium (scene.rs:588): the outer one asks the inner one twice (volume.rs:66-73) and
    each answer is a draw of its own.  Two and three media deep, in a deep tree (the streaming kernel), a shallow one and a List
    (the pooled and the one-path-per-lane kernels), in every box mode, and in the clean tree against the oracle's List."""
    d = synth.spheres_desc(n=n, media=6, nested=True)
    d["aggregate"] = aggregate
    nested = [i for i, s in enumerate(d["shapes"]) if s["type"] == "medium" and s["boundary"]["type"] == "medium"]
    deepest = [i for i in nested if d["shapes"][i]["boundary"]["boundary"]["type"] == "medium"]
    assert len(nested) == 4 and len(deepest) == 2
    o = orc.OracleScene(d)
    hs = synth.host_scene(d)
    assert len(hs.media()) == 6 + 4 + 2
    g = rrt.Scene(hs)
    ro, rd = raygen.ray_batch(o, 48, 40, n_random=8000, seed=4)
    rng = np.random.default_rng(9)
    for i in nested:  # and rays aimed at the nested media from all around, some starting inside
        b = d["shapes"][i]
        while b["type"] == "medium":
            b = b["boundary"]
        centre = np.float32(b["transform"][0])
        start = (centre + rng.normal(0.0, 1.0, (700, 3)) * rng.choice([0.3, 2.0, 5.0], (700, 1))).astype(np.float32)
        aim = (centre + rng.uniform(-0.4, 0.4, (700, 3))).astype(np.float32)
        ro, rd = np.concatenate([ro, start]), np.concatenate([rd, aim - start])
    for t_min in (0.001, -1.0):
        p_ref, t_ref = o.intersect(ro, rd, t_min=t_min)
        assert np.isin(p_ref, nested).sum() > 400 and np.isin(p_ref, deepest).sum() > 25, "too few rays stop inside the nested media to tell"
        for mode in (rrt.AABB_TIGHT, rrt.AABB_REFERENCE, rrt.AABB_EXHAUSTIVE):
            p, t = g.hit(ro, rd, t_min=t_min, aabb_mode=mode)
            assert np.array_equal(p, p_ref), "primitive ids differ in mode %d: %d rays" % (mode, (p != p_ref).sum())
            assert np.array_equal(bits(t), bits(t_ref))
    ref, st_ref = o.render(48, 40, 6, seed=2, bg=rrt.BG_SKY)
    for pooled in (True, False):
        g.set_pooling(pooled)
        for mode in (rrt.AABB_TIGHT, rrt.AABB_REFERENCE, rrt.AABB_EXHAUSTIVE):
            img, st = g.render(samples=6, seed=2, background=rrt.BG_SKY, aabb_mode=mode)
            same = (bits(img) == bits(ref)) | (np.isnan(img) & np.isnan(ref))
            assert same.all(), "%d of %d values differ (pooled %s, mode %d)" % ((~same).sum(), same.size, pooled, mode)
            assert st["rays"] == int(st_ref[0])
    o_list = orc.OracleScene(dict(d, aggregate=0))
    g_clean = rrt.Scene(synth.host_scene(d, "BVH_CLEAN"))
    p_ref, t_ref = o_list.intersect(ro, rd)
    p, t = g_clean.hit(ro, rd)
    assert np.array_equal(p, p_ref) and np.array_equal(bits(t), bits(t_ref))
    ref, st_ref = o_list.render(48, 40, 6, seed=2, bg=rrt.BG_SKY)
    img, st = g_clean.render(samples=6, seed=2, background=rrt.BG_SKY)
    same = (bits(img) == bits(ref)) | (np.isnan(img) & np.isnan(ref))
    assert same.all() and st["rays"] == int(st_ref[0])


@pytest.mark.parametrize("aggregate,n", [(0, 40), (1, 40), (0, 7)])
def test_one_big_leaf_with_ties_bit_exact(gpu, aggregate, n):
    """Stacked quads whose boxes most rays cross (the candidate list of a wave overflows and is flushed in parts), shapes
    duplicated in place (exact ties: the earlier shape wins), spheres mixed in; as a List and as the reference's BVH; also
    with a negative lower bound, where hits behind the origin count."""
    d = synth.slats_desc(n=n, aggregate=aggregate)
    o = orc.OracleScene(d)
    g = rrt.Scene(synth.host_scene(d))
    ro, rd = raygen.ray_batch(o, 40, 32, n_random=6000, seed=11)
    rng = np.random.default_rng(3)
    k = 6000
    zo = np.stack([rng.uniform(-1.5, 1.5, k), rng.uniform(-1.5, 1.5, k), rng.uniform(-3.0, 3.0, k)], axis=1).astype(np.float32)
    zd = np.stack([rng.normal(0, 0.15, k), rng.normal(0, 0.15, k), rng.choice([-1.0, 1.0], k)], axis=1).astype(np.float32)
    ro, rd = np.concatenate([ro, zo]), np.concatenate([rd, zd])
    for t_min in (0.001, -2.5):
        p_ref, t_ref = o.intersect(ro, rd, t_min=t_min)
        assert (p_ref >= 0).sum() > 4000
        for mode in (rrt.AABB_TIGHT, rrt.AABB_REFERENCE):
            p, t = g.hit(ro, rd, t_min=t_min, aabb_mode=mode)
            assert np.array_equal(p, p_ref), "primitive ids differ in mode %d: %d rays" % (mode, (p != p_ref).sum())
            assert np.array_equal(bits(t), bits(t_ref))
    ref, st_ref = o.render(40, 32, 3, seed=4, bg=rrt.BG_SKY)
    for mode in (rrt.AABB_TIGHT, rrt.AABB_REFERENCE):
        img, st = g.render(seed=4, background=rrt.BG_SKY, aabb_mode=mode)
        same = (bits(img) == bits(ref)) | (np.isnan(img) & np.isnan(ref))
        assert same.all(), "%d of %d values differ (mode %d)" % ((~same).sum(), same.size, mode)
        assert st["rays"] == int(st_ref[0])


@pytest.mark.parametrize("case", ["terrain", "terrain39k", "soup", "slats", "spheres", "coincident", "tiny"])
def test_clean_bvh_returns_what_the_list_aggregate_returns(gpu, case):
    """RRT_AGGREGATE_BVH_CLEAN (a tree of its own, built on the device) against the oracle's LIST aggregate: nearest hit, equal
    distances to the shape that is first in scene order, bit-exact distances; rendered frames bit-exact too."""
    if case == "terrain":
        d = synth.terrain_desc(60, seed=3, width=40, height=40, samples=2)
    elif case == "terrain39k":  # a deeper tree; ray batches only (the List oracle tests every triangle for every ray)
        d = synth.terrain_desc(140, seed=2, width=16, height=16, samples=1)
    elif case == "soup":
        d = synth.soup_desc()  # boxes that overlap heavily, triangles from a thousandth of the scene to all of it
    elif case == "slats":
        d = synth.slats_desc(n=40, aggregate=0)  # shapes duplicated in place: exact ties
    elif case == "tiny":
        d = synth.slats_desc(n=1, aggregate=0, spheres=0)  # three quads: the whole tree is one leaf link, no inner node at all
    elif case == "coincident":
        d = synth.slats_desc(n=6, aggregate=0, spheres=1)  # 14 copies of one quad: a leaf no sweep can split (28 triangles)
        d["shapes"] = d["shapes"][:2] + [dict(d["shapes"][2], material=i % 3) for i in range(14)] + d["shapes"][2:]
    else:
        d = synth.spheres_desc(media=6)
    o = orc.OracleScene(dict(d, aggregate=0))
    g = rrt.Scene(synth.host_scene(d, "BVH_CLEAN"))
    W, H, spp = d["logistics"]["resolution_x"], d["logistics"]["resolution_y"], d["logistics"]["samples"]
    ro, rd = raygen.ray_batch(o, W, H, n_random=600 if case == "terrain39k" else 5000, seed=13)
    for t_min in (0.001, -1.5):
        p_ref, t_ref = o.intersect(ro, rd, t_min=t_min)
        assert (p_ref >= 0).sum() > (200 if case == "tiny" else 2000)
        p, t = g.hit(ro, rd, t_min=t_min)
        assert np.array_equal(p, p_ref), "primitive ids differ: %d rays" % (p != p_ref).sum()
        assert np.array_equal(bits(t), bits(t_ref))
    if case == "terrain39k":
        return
    ref, st_ref = o.render(W, H, spp, seed=6, bg=rrt.BG_SKY)
    img, st = g.render(seed=6, background=rrt.BG_SKY)
    same = (bits(img) == bits(ref)) | (np.isnan(img) & np.isnan(ref))
    assert same.all(), "%d of %d values differ" % ((~same).sum(), same.size)
    assert st["rays"] == int(st_ref[0])


@pytest.mark.parametrize("tree", ["BVH", "BVH_CLEAN"])
def test_stack_split_between_lds_and_private_memory_changes_nothing(gpu, tree):
    """rrt_scene_set_lds_stack: with only one or three stack entries in LDS nearly every push and pop goes to the lane's private
    memory; the frame must stay bit-identical to the default split's (and so to the oracle's, which other tests check)."""
    d = synth.terrain_desc(70, seed=4, width=48, height=40, samples=3)
    g = rrt.Scene(synth.host_scene(d, tree))
    ref, st_ref = g.render(seed=3, background=rrt.BG_SKY)
    ro, rd = raygen.ray_batch(orc.OracleScene(d), 48, 40, n_random=3000, seed=5)
    p_ref, t_ref = g.hit(ro, rd)
    assert (p_ref >= 0).sum() > 1000
    for entries in (1, 3, 64, 0):
        g.set_lds_stack(entries)
        p, t = g.hit(ro, rd)  # the ray-batch kernel splits its stack the same way
        assert np.array_equal(p, p_ref) and np.array_equal(bits(t), bits(t_ref))
        img, st = g.render(seed=3, background=rrt.BG_SKY)
        same = (bits(img) == bits(ref)) | (np.isnan(img) & np.isnan(ref))
        assert same.all(), "%d of %d values differ with %d entries in LDS" % ((~same).sum(), same.size, entries)
        assert st["rays"] == st_ref["rays"]
    with pytest.raises(rrt.RrtError):
        g.set_lds_stack(65)


def test_light_densities_come_from_the_shape_not_from_the_tree(gpu):
    """Single-leaf scene whose light (a scaled sphere) sticks out of the leaf's box: the reference's tree misses it there, but
    Sphere::pdf (shape.rs:141-147) asks the sphere itself; the harvesting kernel must reproduce both."""
    d = synth.lone_light_desc()
    o = orc.OracleScene(d)
    hs = synth.host_scene(d)
    assert hs.desc.n_nodes == 1 and hs.desc.n_lights == 1
    g = rrt.Scene(hs)
    W, H, spp = 48, 40, 16
    ro, rd = raygen.ray_batch(o, W, H, n_random=4000, seed=21)
    p_ref, t_ref = o.intersect(ro, rd)
    p_list, _ = orc.OracleScene(dict(d, aggregate=0)).intersect(ro, rd)
    assert (p_ref != p_list).sum() > 10, "the leaf's box should hide part of the sphere from the tree"
    ref, st_ref = o.render(W, H, spp, seed=9)
    for mode in (rrt.AABB_TIGHT, rrt.AABB_REFERENCE):
        p, t = g.hit(ro, rd, aabb_mode=mode)
        assert np.array_equal(p, p_ref) and np.array_equal(bits(t), bits(t_ref))
        img, st = g.render(seed=9, aabb_mode=mode)
        same = (bits(img) == bits(ref)) | (np.isnan(img) & np.isnan(ref))
        assert same.all(), "%d of %d values differ (mode %d)" % ((~same).sum(), same.size, mode)
        assert st["rays"] == int(st_ref[0])


@pytest.mark.parametrize("aggregate,nodes", [(0, 1), (1, 3)])
def test_a_leaf_whose_only_sphere_is_the_light(gpu, aggregate, nodes):
    """A Lambert floor (two triangles) under one stretched emissive sphere.  As a List the scene is a single leaf whose only sphere
    is tested by the rays' own lanes while the harvesting kernel takes the light's density (Sphere::pdf, shape.rs:141-147) from
    that very test; as a BVH the sphere has a leaf to itself.  Hits, frames and ray counts must be the oracle's in both box-test
    modes, with the pooled schedule and with one path per lane."""
    d = dict(synth.floor_and_sphere_light_desc(), aggregate=aggregate)
    o = orc.OracleScene(d)
    hs = synth.host_scene(d)
    assert hs.desc.n_nodes == nodes and hs.desc.n_lights == 1
    g = rrt.Scene(hs)
    W, H, spp = 48, 40, 16
    ro, rd = raygen.ray_batch(o, W, H, n_random=4000, seed=22)
    p_ref, t_ref = o.intersect(ro, rd)
    assert (p_ref == 2).sum() > 50 and (p_ref == 0).sum() + (p_ref == 1).sum() > 200, "rays should meet the sphere and the floor"
    ref, st_ref = o.render(W, H, spp, seed=5)
    for pooled in (True, False):
        g.set_pooling(pooled)
        for mode in (rrt.AABB_TIGHT, rrt.AABB_REFERENCE, rrt.AABB_EXHAUSTIVE):
            p, t = g.hit(ro, rd, aabb_mode=mode)
            assert np.array_equal(p, p_ref) and np.array_equal(bits(t), bits(t_ref)), "hits differ (pooled %s, mode %d)" % (pooled, mode)
            img, st = g.render(seed=5, aabb_mode=mode)
            same = (bits(img) == bits(ref)) | (np.isnan(img) & np.isnan(ref))
            assert same.all(), "%d of %d values differ (pooled %s, mode %d)" % ((~same).sum(), same.size, pooled, mode)
            assert st["rays"] == int(st_ref[0])


def test_scene_file_with_a_device_built_or_clean_tree(gpu):
    """rrt_host_scene_load_as: the asset files rendered with the tree built on the device (bit-identical to the host build) and
    with the clean tree (the oracle's List aggregate; on these scenes also the BVH's frame)."""
    name, W, H, spp = "Cornell_Box.json", 40, 40, 4
    import sceneio
    o_list = orc.OracleScene(dict(sceneio.read_scene(orc.asset(name), {"resolution_x": W, "resolution_y": H, "samples": spp}), aggregate=0))
    ref_list, _ = o_list.render(W, H, spp, seed=2)
    host, _ = rrt.Scene.load(orc.asset(name), width=W, height=H, samples=spp).render(seed=2)
    dev, _ = rrt.Scene.load(orc.asset(name), width=W, height=H, samples=spp, tree="BVH_GPU").render(seed=2)
    clean, _ = rrt.Scene.load(orc.asset(name), width=W, height=H, samples=spp, tree="BVH_CLEAN").render(seed=2)
    assert np.array_equal(bits(host), bits(dev))
    same = (bits(clean) == bits(ref_list)) | (np.isnan(clean) & np.isnan(ref_list))
    assert same.all()


def _same_records(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes()


@pytest.mark.parametrize("case", ["terrain20k", "terrain180k", "soup", "spheres", "slats", "cornell", "tiny2", "tiny6", "tiny8"])
def test_gpu_built_bvh_equals_the_host_built_one(gpu, case):
    """RRT_AGGREGATE_BVH_GPU must give the reference's tree bit for bit: node boxes (signed zeros included), cut axes, child
    links, leaf ranges and the order of the shapes in the leaves, also where box centres tie (the sort must be stable)."""
    import sceneio
    if case == "terrain20k":
        d = synth.terrain_desc(100, seed=2)
    elif case == "terrain180k":
        d = synth.terrain_desc(300, seed=5)
    elif case == "soup":
        d = synth.soup_desc(n=20000)
    elif case == "spheres":
        d = synth.spheres_desc(media=6)
    elif case == "slats":
        d = synth.slats_desc(n=40, aggregate=1)
    elif case == "cornell":
        d = sceneio.read_scene(orc.asset("Cornell_Box_With_Sphere.json"))
    else:
        d = synth.slats_desc(n=2 if case == "tiny8" else 1, aggregate=1, spheres=0)  # 4 or 3 quads
        if case == "tiny2":
            d["shapes"] = d["shapes"][:1]  # two triangles: a leaf without any sorting (aggregate.rs:204)
    host = synth.host_scene(d, "BVH")
    dev = synth.host_scene(d, "BVH_GPU")
    assert len(host.nodes()) == len(dev.nodes())
    assert _same_records(host.nodes(), dev.nodes())
    assert _same_records(host.slots(), dev.slots())
    assert host.desc.bvh_depth == dev.desc.bvh_depth


def test_render_is_independent_of_the_pass_split_and_of_odd_sizes(gpu):
    """Frames that are no multiple of the 8x8 wave tile, sample counts that are no multiple of the 64-sample chunk, and
    staging budgets that force 1, 5 and 70 passes all give the oracle's bits."""
    W, H, spp = 45, 19, 70
    o, g = both("Cornell_Box_With_Sphere.json", resolution_x=W, resolution_y=H, samples=spp)
    ref, st_ref = o.render(W, H, spp, seed=8)
    for limit in (0, W * H * 12 * 16, W * H * 12):
        g.set_staging_limit(limit)
        img, st = g.render(seed=8)
        assert np.array_equal(bits(img), bits(ref)), "staging limit %d" % limit
        assert st["rays"] == int(st_ref[0]) and st["paths"] == W * H * spp
    g.set_staging_limit(0)
    one, _ = g.render(width=1, height=1, samples=3, seed=8)
    assert one.shape == (1, 1, 3) and np.isfinite(one).all()


@pytest.mark.parametrize("aggregate", ["BVH", "List", "BVH_GPU", "BVH_CLEAN"])
def test_empty_scene(gpu, aggregate):
    """No shapes at all (an empty BVH has no nodes, aggregate.rs:339-341): every path is one ray that sees the background."""
    hs = rrt.HostScene.new()
    hs.add_material(0, hs.add_texture(0, 0.5, 0.5, 0.5))
    hs.set_camera((0, 0, 1), (0, 0, 0), (0, 1, 0), 60, 1.5, 0.0, 1.0)
    hs.set_logistics(24, 16, 3)
    hs.finalize(aggregate)
    sc = rrt.Scene(hs)
    img, st = sc.render(seed=1, background=rrt.BG_SKY)
    assert st["rays"] == st["paths"] == 24 * 16 * 3 and np.isfinite(img).all() and img.min() > 1.0
    p, _ = sc.hit(np.zeros((5, 3), np.float32), np.ones((5, 3), np.float32))
    assert (p == -1).all()


def test_fast_reciprocal_is_exact_for_every_float(gpu):
    assert rrt.selftest_reciprocal() == 0


def test_resolve_rgb8(gpu):
    rng = np.random.default_rng(0)
    img = rng.uniform(0, 40, (33, 47, 3)).astype(np.float32)
    img[0, 0] = [np.nan, -1.0, 1e9]
    img[1, 1] = [0.0, 16.0, 15.999]
    assert np.array_equal(rrt.resolve_rgb8(img, 16), orc.resolve_rgb8(img, 16))


@pytest.mark.parametrize("name,W,H,spp", [("Cornell_Box_With_Sphere.json", 200, 136, 20), ("Cornell_Box.json", 333, 222, 9), ("Cornell_Box.json", 64, 64, 300)])
def test_pooled_and_lockstep_schedules_give_the_oracle_frame(gpu, name, W, H, spp):
    """Shallow scenes are traced by k_trace_pool (128 paths of a wave pooled in LDS, one kind of work per step) unless
    rrt_scene_set_pooling(0) asks for k_trace (one path per lane): both must give the oracle's frame bit for bit, in both box-test
    modes, over ragged frames (partial 8x8 and 32x32 tiles), across item boundaries (more samples than a 64-sample chunk) and
    with interleaved tiles."""
    o, g = both(name, resolution_x=W, resolution_y=H, samples=spp)
    ref, st_ref = o.render(W, H, spp, seed=12, threads=min(16, len(__import__("os").sched_getaffinity(0))))
    for pooled in (True, False):
        g.set_pooling(pooled)
        for mode in (rrt.AABB_TIGHT, rrt.AABB_REFERENCE):
            img, st = g.render(seed=12, aabb_mode=mode)
            same = (bits(img) == bits(ref)) | (np.isnan(img) & np.isnan(ref))
            assert same.all(), "%d of %d values differ (pooled %s, mode %d)" % ((~same).sum(), same.size, pooled, mode)
            assert st["rays"] == int(st_ref[0]) and st["paths"] == W * H * spp
        tiled = np.zeros_like(ref)
        for k in range(3):
            g.render(seed=12, tile_count=3, tile_index=k, out=tiled)
        assert ((bits(tiled) == bits(ref)) | (np.isnan(tiled) & np.isnan(ref))).all()
    g.set_pooling(True)
    g.set_staging_limit(W * H * 12 * 7)  # seven samples per pass: many short launches of the persistent grid
    img, st = g.render(seed=12)
    assert ((bits(img) == bits(ref)) | (np.isnan(img) & np.isnan(ref))).all() and st["rays"] == int(st_ref[0])


@pytest.mark.parametrize("name,W,H,spp,bg", [("Materials.json", 150, 90, 12, rrt.BG_SKY), ("Globe.json", 131, 77, 8, rrt.BG_SKY),
                                             ("Example_Spec.json", 120, 72, 10, rrt.BG_SKY), ("Example_Spec.json", 64, 40, 70, rrt.BG_BLACK)])
def test_pooled_and_lockstep_schedules_on_textured_scenes_and_media(gpu, name, W, H, spp, bg):
    """The pooled schedule's textured and full variants (image / procedural textures, bump maps, textured lights, a constant
    medium with its keyed draw, metal and glass) against the one-path-per-lane kernel and the oracle."""
    o, g = both(name, resolution_x=W, resolution_y=H, samples=spp)
    ref, st_ref = o.render(W, H, spp, seed=21, bg=bg, threads=min(16, len(__import__("os").sched_getaffinity(0))))
    for pooled in (True, False):
        g.set_pooling(pooled)
        for mode in (rrt.AABB_TIGHT, rrt.AABB_REFERENCE):
            img, st = g.render(seed=21, background=bg, aabb_mode=mode)
            same = (bits(img) == bits(ref)) | (np.isnan(img) & np.isnan(ref))
            assert same.all(), "%d of %d values differ (pooled %s, mode %d)" % ((~same).sum(), same.size, pooled, mode)
            assert st["rays"] == int(st_ref[0]) and st["paths"] == W * H * spp


def test_pooled_schedule_with_scaled_spheres_and_media(gpu):
    d = synth.spheres_desc(media=6)
    o = orc.OracleScene(d)
    g = rrt.Scene(synth.host_scene(d))
    ref, st_ref = o.render(48, 40, 6, seed=8, bg=rrt.BG_SKY)
    for pooled in (True, False):
        g.set_pooling(pooled)
        img, st = g.render(samples=6, seed=8, background=rrt.BG_SKY)
        same = (bits(img) == bits(ref)) | (np.isnan(img) & np.isnan(ref))
        assert same.all() and st["rays"] == int(st_ref[0]), pooled


def test_pooled_schedule_with_a_depth_limit_and_many_lights(gpu):
    """max_depth below the default (paths cut off by the limit see the background), and a scene of more lights than the
    harvesting query handles (the pooled kernel's plain, non-harvesting variant; light-or-lobe draws made in the trace step)."""
    o, g = both("Cornell_Box_With_Sphere.json", resolution_x=72, resolution_y=56, samples=10)
    for depth in (0, 1, 3):
        ref, st_ref = o.render(72, 56, 10, seed=2, bg=rrt.BG_SKY, max_depth=depth)
        for pooled in (True, False):
            g.set_pooling(pooled)
            img, st = g.render(seed=2, background=rrt.BG_SKY, max_depth=depth)
            same = (bits(img) == bits(ref)) | (np.isnan(img) & np.isnan(ref))
            assert same.all() and st["rays"] == int(st_ref[0]), (depth, pooled)
    d = synth.slats_desc(n=6, aggregate=1, spheres=2)
    d["shapes"] += [dict(d["shapes"][-1], transform=([0.3 * i, 0.0, 0.1 * i], [0.0, 0.0, 0.0], [0.5, 1.0, 0.5])) for i in range(1, 4)]  # four light quads
    o2 = orc.OracleScene(d)
    hs = synth.host_scene(d)
    assert hs.desc.n_lights > 3  # the light quads plus every Metal and Dielectric shape (material.rs:162, 209)
    g2 = rrt.Scene(hs)
    ref, st_ref = o2.render(40, 32, 6, seed=5, bg=rrt.BG_SKY)
    for pooled in (True, False):
        g2.set_pooling(pooled)
        img, st = g2.render(samples=6, seed=5, background=rrt.BG_SKY)
        same = (bits(img) == bits(ref)) | (np.isnan(img) & np.isnan(ref))
        assert same.all() and st["rays"] == int(st_ref[0]), pooled


@pytest.mark.parametrize("tree", ["BVH", "BVH_CLEAN"])
def test_deep_tree_frames_do_not_depend_on_how_the_work_is_handed_out(gpu, tree):
    """The deep-tree kernel's waves are persistent and take (tile, samples) items from eight counters, one per band of the frame,
    stealing from the next band when theirs is used up; on large frames the tiles of a band are numbered column by column.  None of
    it may show in the image: a ragged frame against the oracle, the same frame from interleaved tile subsets and from many small
    passes, and a frame large enough for the column order against its two tile subsets."""
    d = synth.terrain_desc(100, seed=4, width=200, height=136, samples=3)
    g = rrt.Scene(synth.host_scene(d, tree))
    o = orc.OracleScene(d if tree == "BVH" else dict(d, aggregate=0))  # a clean tree answers like the reference's List
    ref, st_ref = o.render(200, 136, 3, seed=6, threads=min(16, len(__import__("os").sched_getaffinity(0))))
    full, st = g.render(seed=6)
    same = (bits(full) == bits(ref)) | (np.isnan(full) & np.isnan(ref))
    assert same.all(), "%d of %d values differ" % ((~same).sum(), same.size)
    assert st["rays"] == int(st_ref[0]) and st["paths"] == 200 * 136 * 3
    tiled = np.zeros_like(full)
    for k in range(3):
        g.render(seed=6, tile_count=3, tile_index=k, out=tiled)
    assert np.array_equal(bits(tiled), bits(full))
    g.set_staging_limit(200 * 136 * 12)  # one sample per pass
    again, st2 = g.render(seed=6)
    assert np.array_equal(bits(again), bits(full)) and st2["rays"] == st["rays"]
    g.set_staging_limit(16 << 30)
    W, H = 2048, 1600  # 64 x 50 tiles of 32 x 32: bands of 7 rows = 448 tiles, numbered by columns
    big, st_big = g.render(width=W, height=H, samples=1, seed=2)
    halves = np.zeros_like(big)
    rays = 0
    for k in range(2):
        _, s = g.render(width=W, height=H, samples=1, seed=2, tile_count=2, tile_index=k, out=halves)
        rays += s["rays"]
    assert ((bits(halves) == bits(big)) | (np.isnan(halves) & np.isnan(big))).all() and rays == st_big["rays"] and st_big["paths"] == W * H
