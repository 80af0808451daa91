"""The padded-box shortcuts of the traversal against the mode that has none.

RRT_AABB_TIGHT culls nodes by the padded boxes their primitives occupy, and on small scenes both TIGHT and REFERENCE sift the
pairs of a leaf by padded pair boxes before the exact tests.  RRT_AABB_EXHAUSTIVE uses the reference's node test alone and
sends every primitive of a visited leaf through the exact test, so it equals the reference by construction.  The rays here are
built for the one place where a shortcut could lose a hit: along triangle outlines, at grazing angles, where Moeller-Trumbore's
own rounding accepts points slightly OUTSIDE the triangle (aggregate.rs:324-401 tests every shape of a leaf, shape.rs:335-372).
On shallow trees (the sifting query: at most 16 nodes, Lists included) a ray that runs closer to a triangle's plane than the
padding covers loses its shortcuts (the grazing guard), so there all three modes must return the oracle's answer bit for bit;
deep trees have no guard and keep the weaker contract spelled out in the test."""
import numpy as np
import pytest

import orc
import rays as raygen
import rust_raytracer_amd as rrt
import synth

pytestmark = pytest.mark.gpu

MODES = (rrt.AABB_TIGHT, rrt.AABB_REFERENCE, rrt.AABB_EXHAUSTIVE)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def scene_pair(case):
    if case.endswith(".json"):
        o = orc.OracleScene.from_file(orc.asset(case), {"resolution_x": 32, "resolution_y": 32})
        g = rrt.Scene.load(orc.asset(case), width=32, height=32)
        return o, g
    d = {"slats": lambda: synth.slats_desc(n=24, aggregate=1), "slats_list": lambda: synth.slats_desc(n=24, aggregate=0),
         "soup": lambda: synth.soup_desc(n=400), "terrain": lambda: synth.terrain_desc(12, amplitude=0.02)}[case]()
    return orc.OracleScene(d), rrt.Scene(synth.host_scene(d))


def outside_in_exact_arithmetic(slots, ro, rd, which, prim_slot_hit):
    """Of the rays whose accepted hit is the triangle they were aimed at: how many meet that triangle's plane OUTSIDE the
    triangle when the intersection is worked out in float64, and how far outside (in-plane distance) at most."""
    s = which[prim_slot_hit]
    v0 = slots["v0"][s].astype(np.float64)
    e1 = slots["e1"][s].astype(np.float64)
    e2 = slots["e2"][s].astype(np.float64)
    o = ro[prim_slot_hit].astype(np.float64)
    d = rd[prim_slot_hit].astype(np.float64)
    p = np.cross(d, e2)
    det = np.einsum("ij,ij->i", e1, p)
    ok = det != 0.0
    tv = o - v0
    with np.errstate(divide="ignore", invalid="ignore"):
        u = np.einsum("ij,ij->i", tv, p) / det
        q = np.cross(tv, e1)
        v = np.einsum("ij,ij->i", d, q) / det
        out = ok & ((u < 0.0) | (v < 0.0) | (u + v > 1.0))
        excess = np.maximum(np.maximum(-u, -v), u + v - 1.0)  # in barycentric units
    scale = np.maximum(np.linalg.norm(e1, axis=1), np.linalg.norm(e2, axis=1))
    return int(out.sum()), float((excess * scale)[out].max()) if out.any() else 0.0


def misses_grown_box(slots, s, o, d, pad):
    """float64 slab test: True where the ray o + t d (t >= 0) does not touch the box of triangle slot s grown by pad."""
    v0 = slots["v0"][s].astype(np.float64)
    v1 = v0 + slots["e1"][s].astype(np.float64)
    v2 = v0 + slots["e2"][s].astype(np.float64)
    lo = np.minimum(np.minimum(v0, v1), v2) - pad
    hi = np.maximum(np.maximum(v0, v1), v2) + pad
    o = o.astype(np.float64)
    d = d.astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        t0 = (lo - o) / d
        t1 = (hi - o) / d
    near = np.where(d != 0.0, np.minimum(t0, t1), np.where((o >= lo) & (o <= hi), -np.inf, np.inf))
    far = np.where(d != 0.0, np.maximum(t0, t1), np.where((o >= lo) & (o <= hi), np.inf, -np.inf))
    return ~(np.maximum(near.max(axis=1), 0.0) <= far.min(axis=1))


CASES = ["Cornell_Box_With_Sphere.json", "Cornell_Box.json", "Example_Spec.json", "slats", "slats_list", "soup", "terrain"]

SAFE_ANGLE = 0.01  # deep trees: rad off the triangle's plane from which the padding (1e-5 of the extent) covers the exact test's rounding


@pytest.mark.parametrize("case", CASES)
def test_rays_along_triangle_outlines_at_grazing_angles(gpu, case):
    """What the three modes return for rays along triangle outlines:
    * EXHAUSTIVE equals the oracle on every ray, whatever the angle;
    * shallow trees: so do TIGHT and REFERENCE, on every ray (padded boxes plus the grazing guard);
    * deep trees: TIGHT and REFERENCE equal it on every ray that leaves the plane of the triangle it was aimed at by 0.01 rad
      or more; closer to the plane the reference's arithmetic reports "hits" on triangles the ray passes BESIDE (the rounding
      error of u, v grows as 1/angle and outruns any fixed padding) and TIGHT's node boxes drop exactly those: every ray it
      answers differently is one that, in float64, does not touch the reported triangle's box grown by the padding (nearly
      every one - a dropped artefact can also change what the reference's walk prunes afterwards), and they are few."""
    o, g = scene_pair(case)
    deep = g.host.desc.n_nodes > 16  # the cooperative-leaf walk, whose TIGHT mode has no grazing guard
    slots = g.host.slots()
    bb = o.shape_bboxes()
    extent = float(np.abs(bb[np.isfinite(bb).all(axis=1)]).max())
    pad = 1e-5 * extent
    ro, rd, which, angle = raygen.triangle_outline_grazers(slots, extent, seed=5)
    assert len(ro) > 20000
    slot_of_prim = np.full(int(slots["prim"].max()) + 1, -1, dtype=np.int64)
    slot_of_prim[slots["prim"]] = np.arange(len(slots))
    for t_min in (0.001, 0.0):
        p_ref, t_ref = o.intersect(ro, rd, t_min=t_min)
        assert (p_ref >= 0).sum() > len(ro) // 4
        for mode in MODES:
            p, t = g.hit(ro, rd, t_min=t_min, aabb_mode=mode)
            bad = (p != p_ref) | (bits(t) != bits(t_ref))
            if mode == rrt.AABB_EXHAUSTIVE or not deep:
                assert not bad.any(), "mode %d, t_min %g: %d of %d rays differ from the oracle" % (mode, t_min, bad.sum(), len(ro))
                continue
            steep = angle >= SAFE_ANGLE
            assert not (bad & steep).any(), "mode %d, t_min %g: %d rays at %g rad or more off the plane differ from the oracle" % (
                mode, t_min, (bad & steep).sum(), SAFE_ANGLE)
            i = np.nonzero(bad)[0]
            print("%s mode %d t_min %g: %d of %d rays differ, all closer than %g rad to a triangle's plane" % (case, mode, t_min, len(i), len(ro), SAFE_ANGLE))
            assert len(i) <= len(ro) // 500
            # each of them: the oracle reports a triangle whose grown box the ray does not touch
            assert (p_ref[i] >= 0).all()
            s = slot_of_prim[p_ref[i]]
            assert ((slots["info"][s] >> 30) == 0).all()
            beside = misses_grown_box(slots, s, ro[i], rd[i], 0.99 * pad)
            assert beside.sum() * 4 >= len(i) * 3  # (a dropped rounding artefact can also change which subtrees the reference's walk prunes afterwards)
    # the batch must contain what it was built for: hits the rounding of the exact test lets in from outside the triangle
    prim_of_slot = slots["prim"].astype(np.int64)
    aimed_hit = (p_ref >= 0) & (prim_of_slot[which] == p_ref)
    n_out, far = outside_in_exact_arithmetic(slots, ro, rd, which, aimed_hit)
    print("%s: %d rays, %d hit the triangle aimed at, %d of them from outside it in float64" % (case, len(ro), aimed_hit.sum(), n_out))
    assert n_out > 0


@pytest.mark.parametrize("case", ["Cornell_Box_With_Sphere.json", "Materials.json", "Example_Spec.json", "slats"])
def test_exhaustive_mode_renders_the_oracle_frame(gpu, case):
    o, g = scene_pair(case)
    W, H, spp = (32, 32, 8)
    ref, st_ref = o.render(W, H, spp, seed=3, bg=rrt.BG_SKY)
    for mode in MODES:
        img, st = g.render(width=W, height=H, samples=spp, seed=3, background=rrt.BG_SKY, aabb_mode=mode)
        same = (bits(img) == bits(ref)) | (np.isnan(img) & np.isnan(ref))
        assert same.all(), "%d of %d values differ (mode %d)" % ((~same).sum(), same.size, mode)
        assert st["rays"] == int(st_ref[0])


def _awkward_desc(flat):
    """slats as a List, plus the triangles the guard's bound is weakest for: a sliver (its two edges at v0 almost parallel), a
    triangle a thousand times the size of the rest, a tiny one and - flat = True - a triangle without area (which makes the
    library give up every shortcut: no ray can be trusted to stay clear of a plane that has no direction)."""
    d = synth.slats_desc(n=12, aggregate=0, spheres=2)
    ident_r, ident_s = [0.0, 0.0, 0.0], [1.0, 1.0, 1.0]
    one = np.array([[0, 1, 2]], dtype=np.int32)

    def tri(v, material=0):
        v = np.asarray(v, dtype=np.float64)
        return {"type": "mesh", "object": {"vertices": v, "tex": np.zeros((0, 2)), "vidx": one, "tidx": np.full_like(one, -1)},
                "cull": False, "material": material, "transform": ([0.0, 0.0, 0.0], ident_r, ident_s)}

    d["shapes"].insert(3, tri([[-1.0, 0.3, 0.4], [1.0, 0.3, 0.4], [0.9, 0.3001, 0.45]]))          # sliver
    if flat:
        d["shapes"].insert(5, tri([[0.2, 0.2, 0.1], [0.4, 0.4, 0.3], [0.6, 0.6, 0.5]], material=1))  # three points on a line
    d["shapes"].insert(7, tri([[-900.0, -3.0, -700.0], [800.0, -3.0, -900.0], [100.0, -3.5, 950.0]]))  # far larger than the rest
    d["shapes"].insert(9, tri([[0.1, -0.2, 1.0], [0.1003, -0.2, 1.0], [0.1, -0.1996, 1.0002]], material=1))  # tiny
    return d


@pytest.mark.parametrize("flat", [False, True])
def test_guard_with_slivers_giants_and_odd_ray_lengths(gpu, flat):
    """The guard's bound scales with the angle between a triangle's edges, the ratio of their lengths, the triangle's size against
    the scene's and the ray's distance from the scene; and a direction whose squared length leaves the float range must not
    fool its arithmetic.  All three modes against the oracle on grazing rays of every length, from near and from far."""
    d = _awkward_desc(flat)
    o, g = orc.OracleScene(d), rrt.Scene(synth.host_scene(d))
    assert g.host.desc.n_nodes == 1
    slots = g.host.slots()
    extent = 4.0  # aim at the small triangles as closely as at the giant
    ro, rd, which, angle = raygen.triangle_outline_grazers(slots, extent, seed=11, max_triangles=64, per_edge=4)
    rng = np.random.default_rng(3)
    variants = [(ro, rd)]
    for scale in (1e-18, 1e-21, 1e12, 1e18):  # squared lengths that underflow to denormals or zero, and that overflow
        variants.append((ro, (rd.astype(np.float64) * scale).astype(np.float32)))
    far = (ro.astype(np.float64) - rd.astype(np.float64) / np.linalg.norm(rd, axis=1, keepdims=True) * rng.choice([1e3, 1e5, 1e7], (len(ro), 1))).astype(np.float32)
    variants.append((far, rd))
    axis = rd.copy()
    axis[np.arange(len(axis)), rng.integers(0, 3, len(axis))] = 0.0  # directions inside a coordinate plane
    variants.append((ro, axis))
    total = 0
    for vo, vd in variants:
        vo, vd = np.ascontiguousarray(vo), np.ascontiguousarray(vd)
        for t_min in (0.001, 0.0):
            p_ref, t_ref = o.intersect(vo, vd, t_min=t_min)
            total += int((p_ref >= 0).sum())
            for mode in MODES:
                p, t = g.hit(vo, vd, t_min=t_min, aabb_mode=mode)
                bad = (p != p_ref) | (bits(t) != bits(t_ref))
                assert not bad.any(), "mode %d, t_min %g: %d of %d rays differ from the oracle (first: ray %d)" % (
                    mode, t_min, bad.sum(), len(vo), int(np.nonzero(bad)[0][0]))
    assert total > len(ro)
    # and a frame through the render kernels (their rays take one precomputed reach, or work out their own)
    ref, _ = o.render(32, 24, 6, seed=2, bg=rrt.BG_SKY)
    for mode in MODES:
        img, _ = g.render(width=32, height=24, samples=6, seed=2, background=rrt.BG_SKY, aabb_mode=mode)
        same = (bits(img) == bits(ref)) | (np.isnan(img) & np.isnan(ref))
        assert same.all(), "%d of %d values differ (mode %d)" % ((~same).sum(), same.size, mode)
