"""The 10 M-triangle mesh configuration at FULL size under a check (seed-1 displaced grid of 2236 x 2236 quads, the scene the
4K @ 4096 spp mesh figures are quoted on).  Index packings that only matter at this scale - pair numbers in leaf links, slot
numbers in 64-bit minimum keys, 2^23+ pair records, stacks deeper than the part kept in LDS - are exercised here and nowhere
else.  Takes a few minutes: the oracle builds its own tree of 10 M shapes (aggregate.rs:187-322, on up to 16 threads), and its
List aggregate tests every triangle for every ray.
"""
import gc
import time

import numpy as np
import pytest

import orc
import rays as raygen
import rust_raytracer_amd as rrt
import synth

pytestmark = pytest.mark.gpu

GRID = 2236  # 2 * 2236^2 = 9 999 392 triangles + 2 of the light


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _run(note):
    W = H = 32  # the oracle's frame costs about 2 s of a host core per pixel (below): 64 x 64 took 390 s on 16 cores, 40 x 40 210 s
    d = synth.terrain_desc(GRID, seed=1, width=W, height=H, samples=1)
    note("mesh generated")
    o = orc.OracleScene(d)
    assert o.num_shapes() == 2 * GRID * GRID + 2
    note("oracle: SAH tree over 10 M shapes built")

    # (a) the reference's tree, built on the device, against the oracle's own build and its reference-mode walk
    hs = synth.host_scene(d, "BVH_GPU")
    note("device build of the reference's tree")
    boxes, meta, leaf = o.bvh()
    nodes = hs.nodes()
    assert len(nodes) == len(meta)
    assert np.array_equal(nodes["bmin"], boxes[:, :3]) and np.array_equal(nodes["bmax"], boxes[:, 3:])
    is_leaf = (nodes["b"] & 0x80000000) != 0
    assert np.array_equal(is_leaf.astype(np.int32), meta[:, 0])
    inner = ~is_leaf
    assert np.array_equal(nodes["b"][inner], meta[inner, 1].astype(np.uint32))
    assert np.array_equal(nodes["a"][inner] - np.nonzero(inner)[0], meta[inner, 2].astype(np.uint32))
    assert np.array_equal(hs.slots()["prim"].astype(np.int32), leaf)
    del boxes, meta, leaf, nodes, is_leaf, inner
    note("tree identical to the oracle's (%d nodes)" % hs.desc.n_nodes)
    g = rrt.Scene(hs)
    note("reference tree uploaded")
    # The reference's walk (its box test on its origin-containing boxes) visits about 700 000 nodes and tests 2.8 M triangles for
    # ONE camera ray of this mesh - 0.45 s on a host core - so the oracle side is fed in portions, with a progress line each.
    ro, rd = raygen.ray_batch(o, W, H, n_random=1300, seed=31, max_corners=300)
    assert len(ro) >= 4096
    note("ray batch of %d rays (camera, interior, secondary, axis-parallel, aimed at box corners)" % len(ro))
    p_ref = np.zeros(len(ro), dtype=np.int32)
    t_ref = np.zeros(len(ro), dtype=np.float32)
    for lo in range(0, len(ro), 512):
        p_ref[lo:lo + 512], t_ref[lo:lo + 512] = o.intersect(ro[lo:lo + 512], rd[lo:lo + 512])
        note("oracle: reference-mode walk of rays %d..%d" % (lo, min(lo + 512, len(ro))))
    assert (p_ref >= 0).sum() > 1500 and p_ref.max() > 5_000_000
    for mode in (rrt.AABB_TIGHT, rrt.AABB_REFERENCE):
        p, t = g.hit(ro, rd, aabb_mode=mode)
        assert np.array_equal(p, p_ref), "primitive ids differ in mode %d: %d rays" % (mode, (p != p_ref).sum())
        assert np.array_equal(bits(t), bits(t_ref)), "hit distances differ in mode %d" % mode
        note("%d rays on the reference tree bit-exact, AABB mode %d" % (len(ro), mode))
    # (c) a 32 x 32 frame at 1 spp: radiance sums and ray count
    ref = np.zeros((H, W, 3), dtype=np.float32)
    rays_ref = 0
    for y0 in range(0, H, 4):
        part, st_part = o.render(W, H, 1, seed=3, threads=16, rows=(y0, y0 + 4))
        ref[y0:y0 + 4] = part[y0:y0 + 4]
        rays_ref += int(st_part[0])
        note("oracle: rows %d..%d of the %d x %d @ 1 frame" % (y0, y0 + 4, W, H))
    img, st = g.render(seed=3)
    same = (bits(img) == bits(ref)) | (np.isnan(img) & np.isnan(ref))
    assert same.all(), "%d of %d values differ" % ((~same).sum(), same.size)
    assert st["rays"] == rays_ref and st["paths"] == W * H
    note("%d x %d @ 1 frame on the reference tree bit-exact, %d rays" % (W, H, st["rays"]))
    # (no frame with RRT_AABB_REFERENCE here: with the reference's box test alone a wave needs about 15 s for 64 rays of this
    # mesh - 700 000 dependent node visits each - and the ray batch above has already walked the tree in that mode)
    g.close()
    hs.close()
    del g, hs
    gc.collect()

    # (b) the clean tree against the oracle's List aggregate (every triangle tested for every ray)
    hc = synth.host_scene(d, "BVH_CLEAN")
    gcl = rrt.Scene(hc)
    note("clean tree built and uploaded")
    o.close()
    o_list = orc.OracleScene(dict(d, aggregate=0))
    rng = np.random.default_rng(5)
    pick = rng.choice(len(ro), 384, replace=False)
    lo, ld = ro[pick], rd[pick]
    p_list, t_list = o_list.intersect(lo, ld)
    note("oracle: 384 rays against all 10 M triangles (List)")
    assert (p_list >= 0).sum() > 150
    p, t = gcl.hit(lo, ld)
    assert np.array_equal(p, p_list), "primitive ids differ: %d rays" % (p != p_list).sum()
    assert np.array_equal(bits(t), bits(t_list))
    # the clean tree's frame: same ray count and sums as the List aggregate's on an 8 x 8 frame
    ref, st_ref = o_list.render(8, 8, 1, seed=4, threads=16)
    note("oracle: 8 x 8 @ 1 frame with the List aggregate")
    img, st = gcl.render(width=8, height=8, samples=1, seed=4)
    same = (bits(img) == bits(ref)) | (np.isnan(img) & np.isnan(ref))
    assert same.all() and st["rays"] == int(st_ref[0])
    # and a full-size sanity pass of the clean tree: 4.2 M camera-like rays, every hit distance finite, every id a valid shape
    big = np.tile(ro, (1024, 1))
    bigd = np.tile(rd, (1024, 1))
    p_big, t_big = gcl.hit(big, bigd)
    p_one, t_one = gcl.hit(ro, rd)
    assert np.array_equal(p_big.reshape(1024, -1), np.tile(p_one, (1024, 1))) and np.array_equal(bits(t_big).reshape(1024, -1), np.tile(bits(t_one), (1024, 1)))
    note("clean tree: List parity on 384 rays + frame; %d rays self-consistent" % len(big))


def test_ten_million_triangles_reference_tree_and_clean_tree(gpu, capsys):
    t0 = time.time()

    def note(msg):  # progress goes to the real stdout: a run of minutes must not look hung
        with capsys.disabled():
            print("[c5] %-66s %6.1f s" % (msg, time.time() - t0), flush=True)

    _run(note)
