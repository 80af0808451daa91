"""Hand-checkable known answers for the CPU oracle (the reference ships no tests of its own)."""
import ctypes as C
import math

import numpy as np
import pytest

import orc

L = orc.lib()
a32 = orc.a32
FMAX = float(np.finfo(np.float32).max)


def tri_hit(v0, v1, v2, o, d, cull=False, tmin=0.001, tmax=FMAX):
    t = C.c_float(0)
    r = L.orc_unit_triangle_hit(a32(v0), a32(v1), a32(v2), int(cull), a32(o), a32(d), tmin, tmax, C.byref(t))
    return (t.value if r else None)


def sphere_hit(o, d, radius=1.0, tr=(0, 0, 0), rot=(0, 0, 0), scl=(1, 1, 1), tmin=0.001, tmax=FMAX):
    t = C.c_float(0)
    r = L.orc_unit_sphere_hit(a32(tr), a32(rot), a32(scl), radius, a32(o), a32(d), tmin, tmax, C.byref(t))
    assert r >= 0
    return (t.value if r else None)


def box(bmin, bmax, o, d, tmin=0.001, tmax=FMAX, mode=0):
    return bool(L.orc_unit_aabb_intersect(a32(bmin), a32(bmax), a32(o), a32(d), tmin, tmax, mode))


def test_triangle_moller_trumbore():
    v0, v1, v2 = (0, 0, 0), (1, 0, 0), (0, 1, 0)  # counter-clockwise seen from +z
    assert tri_hit(v0, v1, v2, (0.25, 0.25, 1), (0, 0, -1)) == pytest.approx(1.0)
    assert tri_hit(v0, v1, v2, (0.25, 0.25, 2), (0, 0, -2)) == pytest.approx(1.0)  # direction is not normalised: t scales
    assert tri_hit(v0, v1, v2, (0.75, 0.75, 1), (0, 0, -1)) is None  # u + v > 1
    assert tri_hit(v0, v1, v2, (-0.1, 0.2, 1), (0, 0, -1)) is None
    assert tri_hit(v0, v1, v2, (0.25, 0.25, 1), (1, 0, 0)) is None  # parallel
    # from behind: hit without culling, culled with it (shape.rs:345-352)
    assert tri_hit(v0, v1, v2, (0.25, 0.25, -1), (0, 0, 1)) == pytest.approx(1.0)
    assert tri_hit(v0, v1, v2, (0.25, 0.25, -1), (0, 0, 1), cull=True) is None
    assert tri_hit(v0, v1, v2, (0.25, 0.25, 1), (0, 0, -1), cull=True) == pytest.approx(1.0)
    # t range is open on both sides
    assert tri_hit(v0, v1, v2, (0.25, 0.25, 1), (0, 0, -1), tmax=1.0) is None
    assert tri_hit(v0, v1, v2, (0.25, 0.25, 1), (0, 0, -1), tmin=1.0) is None
    # edges u == 0 and v == 0 are inside
    assert tri_hit(v0, v1, v2, (0.0, 0.5, 1), (0, 0, -1)) == pytest.approx(1.0)
    assert tri_hit(v0, v1, v2, (0.5, 0.0, 1), (0, 0, -1)) == pytest.approx(1.0)


def test_sphere_quadratic():
    assert sphere_hit((0, 0, 5), (0, 0, -1)) == pytest.approx(4.0)
    assert sphere_hit((0, 0, 0), (0, 0, -1)) == pytest.approx(1.0)  # from inside: far root
    assert sphere_hit((0, 0, 5), (0, 0, 1)) is None
    assert sphere_hit((1, 0, 5), (0, 0, -1)) is None  # tangent: discriminant == 0 is a miss (strict >)
    assert sphere_hit((3, 0, 5), (0, 0, -1), tr=(3, 0, 0)) == pytest.approx(4.0)
    # the ray is not renormalised in local space: scale 2 sphere of radius 1 is hit at world distance 3
    assert sphere_hit((0, 0, 5), (0, 0, -1), scl=(2, 2, 2)) == pytest.approx(3.0)
    # negative radius hits like |r|
    assert sphere_hit((0, 0, 5), (0, 0, -1), radius=-1.0) == pytest.approx(4.0)
    assert sphere_hit((0, 0, 5), (0, 0, -1), tmax=4.0) == pytest.approx(6.0) or sphere_hit((0, 0, 5), (0, 0, -1), tmax=4.0) is None


def test_aabb_reference_quirks():
    b0, b1 = (1, 1, 1), (2, 2, 2)
    assert box(b0, b1, (0, 0, 0), (1, 1, 1))
    assert not box(b0, b1, (0, 0, 0), (-1, -1, -1))  # behind the ray
    assert not box(b0, b1, (0, 0, 0), (1, 1, 1), tmax=0.5)  # entry beyond t_max
    # disjoint slabs: x in [3,4], y in [1,2] is rejected, x in [1,2], y in [3,4] passes (aggregate.rs:491-503 compares
    # an exit with the ENTRY bound only)
    assert not box((3, 1, -9), (4, 2, 9), (0, 0, 0), (1, 1, 0.001))
    assert box((1, 3, -9), (2, 4, 9), (0, 0, 0), (1, 1, 0.001))
    assert not box((1, 3, -9), (2, 4, 9), (0, 0, 0), (1, 1, 0.001), mode=1)  # the complete test rejects both
    # zero-thickness boxes (the Cornell walls) are hit
    assert box((0, 0, 5), (1, 1, 5), (0.5, 0.5, 0), (0, 0, 1))
    assert box((0, 0, 5), (1, 1, 5), (0.5, 0.5, 0), (0, 0, 1), mode=1)
    # zero direction component with the origin outside that slab misses; inside it (0 * inf = NaN) passes
    assert not box(b0, b1, (0, 1.5, 1.5), (0, 0, 1))
    assert box(b0, b1, (1.5, 1.5, 0), (0, 0, 1)) and box(b0, b1, (1.5, 1.5, 0), (0, 0, 1), mode=1)
    assert box(b0, b1, (1.0, 1.5, 0), (0, 0, 1))  # origin on the slab plane


def test_perlin_noise():
    for p in [(0, 0, 0), (1, 2, 3), (-4, 7, 250), (255, 255, 255)]:
        assert L.orc_unit_noise(*map(float, p)) == 0.0  # gradient noise vanishes on the lattice
    vals = [L.orc_unit_noise(x * 0.37 + 0.1, x * 0.11 - 3.3, 0.5 + x * 0.23) for x in range(400)]
    assert max(map(abs, vals)) <= 1.5 and np.std(vals) > 0.1
    # period 256 in every axis (utils.rs:122-124)
    assert L.orc_unit_noise(0.3, 0.6, 0.9) == pytest.approx(L.orc_unit_noise(256.3, 0.6, 0.9), abs=1e-4)
    assert L.orc_unit_turbulence(0.3, 0.6, 0.9, 0, 0.5) == 0.0
    assert L.orc_unit_turbulence(0.3, 0.6, 0.9, 1, 0.5) == pytest.approx(abs(L.orc_unit_noise(0.3, 0.6, 0.9)))


def test_matrix_transform_inverse():
    m = np.zeros(16, dtype=np.float32)
    L.orc_unit_transform_matrix(a32([1, 2, 3]), a32([30, 45, 60]), a32([2, 3, 4]), m)
    inv = np.zeros(16, dtype=np.float32)
    assert L.orc_unit_matrix_inverse(m, inv) == 1
    assert np.allclose(m.reshape(4, 4).astype(np.float64) @ inv.reshape(4, 4).astype(np.float64), np.eye(4), atol=1e-5)
    # T * Rx * Ry * Rz * S applied to (1,0,0): scale, rotate z then y then x, translate
    def rot(axis, deg):
        c, s = math.cos(math.radians(deg)), math.sin(math.radians(deg))
        r = np.eye(3)
        i, j = [(1, 2), (2, 0), (0, 1)][axis]
        r[i, i], r[i, j], r[j, i], r[j, j] = c, -s, s, c
        return r
    expect = rot(0, 30) @ rot(1, 45) @ rot(2, 60) @ np.diag([2, 3, 4]) @ np.array([1.0, 0, 0]) + np.array([1, 2, 3])
    assert np.allclose(m.reshape(4, 4)[:3, :3] @ np.array([1, 0, 0]) + m.reshape(4, 4)[:3, 3], expect, atol=1e-5)
    sing = np.zeros(16, dtype=np.float32)
    assert L.orc_unit_matrix_inverse(sing, inv) == 0  # "Singular matrix"


def test_elementary_functions_accuracy():
    rng = np.random.default_rng(0)
    x = rng.uniform(-50, 50, 20000).astype(np.float32)
    out = np.zeros_like(x)
    z = np.zeros_like(x)
    L.orc_unit_detmath(0, len(x), x, z, out)
    assert np.abs(out - np.sin(x.astype(np.float64))).max() < 2e-6
    L.orc_unit_detmath(1, len(x), x, z, out)
    assert np.abs(out - np.cos(x.astype(np.float64))).max() < 2e-6
    u = rng.uniform(-1, 1, 20000).astype(np.float32)
    L.orc_unit_detmath(2, len(u), u, z, out)
    assert np.abs(out - np.arcsin(u.astype(np.float64))).max() < 2e-6
    p = (rng.integers(1, 1 << 24, 20000) / 16777216.0).astype(np.float32)
    L.orc_unit_detmath(3, len(p), p, z, out)
    assert np.abs(out - np.log(p.astype(np.float64))).max() < 2e-6
    y = rng.uniform(-3, 3, 20000).astype(np.float32)
    L.orc_unit_detmath(4, len(x), x, y, out)  # atan2(x, y)
    assert np.abs(out - np.arctan2(x.astype(np.float64), y.astype(np.float64))).max() < 2e-6
    one = np.array([0.0], dtype=np.float32)
    L.orc_unit_detmath(3, 1, one, one, out[:1])
    assert out[0] == -np.inf


def test_random_stream():
    f = np.zeros(4096, dtype=np.float32)
    L.orc_unit_rng(1, 2, 3, len(f), f, 0, None, None)
    assert f.min() >= 0.0 and f.max() < 1.0 and abs(f.mean() - 0.5) < 0.02
    assert np.all(f * 16777216.0 == np.floor(f * 16777216.0))  # multiples of 2^-24 (rand 0.7.3 Standard f32)
    g = np.zeros(4096, dtype=np.float32)
    L.orc_unit_rng(1, 2, 4, len(g), g, 0, None, None)
    assert abs(np.corrcoef(f, g)[0, 1]) < 0.05 and f[0] != g[0]
    idx = C.c_uint32(99)
    counts = np.zeros(3)
    for s in range(3000):
        L.orc_unit_rng(7, 5, s, 0, f, 3, C.byref(idx), None)
        counts[idx.value] += 1
    assert counts.min() > 900


def test_samplers():
    out = np.zeros(12, dtype=np.float32)
    ys = []
    for s in range(500):
        L.orc_unit_samplers(3, 9, s, 90.0, 400.0 ** 2, out)
        disk, sph, cosd, tos = out[0:3], out[3:6], out[6:9], out[9:12]
        assert disk[1] >= 0 and abs(disk[0] ** 2 + disk[1] ** 2 - 1) < 1e-5 and disk[2] == 0  # upper unit semicircle (utils.rs:10-14)
        assert abs(np.linalg.norm(sph) - 1) < 1e-5 and sph[1] >= 0  # +y hemisphere only (utils.rs:16-24)
        assert abs(np.linalg.norm(cosd) - 1) < 1e-5 and cosd[2] >= 0
        assert abs(np.linalg.norm(tos) - 1) < 1e-5 and tos[2] >= math.sqrt(1 - 90.0 ** 2 / 400.0 ** 2) - 1e-5
        ys.append(cosd[2])
    assert abs(np.mean(ys) - 2.0 / 3.0) < 0.03  # E[cos] under a cosine-weighted density


def test_dielectric_helpers():
    refl = np.zeros(3, dtype=np.float32)
    refr = np.zeros(3, dtype=np.float32)
    s = C.c_float(0)
    v = np.array([math.sin(0.5), -math.cos(0.5), 0], dtype=np.float32)
    L.orc_unit_reflect_refract(v, a32([0, 1, 0]), 1.0 / 1.5, refl, refr, C.byref(s), math.cos(0.5))
    assert np.allclose(refl, [v[0], -v[1], 0], atol=1e-6)
    assert math.asin(refr[0] / np.linalg.norm(refr)) == pytest.approx(math.asin(math.sin(0.5) / 1.5), abs=1e-5)  # Snell
    r0 = ((1 - 1 / 1.5) / (1 + 1 / 1.5)) ** 2
    assert s.value == pytest.approx(r0 + (1 - r0) * (1 - math.cos(0.5)) ** 5, rel=1e-5)


def test_camera_corners_and_v_offset():
    sc = orc.OracleScene.from_file(orc.asset("Cornell_Box.json"), {"resolution_x": 100, "resolution_y": 100})
    cam = sc.camera()
    origin, llc, hor, ver = cam[0:3], cam[3:6], cam[6:9], cam[9:12]
    assert np.allclose(origin, [278, 273, -800])
    assert np.allclose(llc + 0.5 * hor + 0.5 * ver - origin, [0, 0, 10], atol=1e-3)  # looks down +z at focus distance 10
    ro, rd = sc.camera_rays(100, 100, 0, [0, 99, 0], [0, 0, 99], [0, 0, 0])
    assert np.allclose(ro, origin)
    # main.rs:228 uses (res_y - y): row 0 samples v in [1, 1 + 1/H), i.e. one pixel above the frustum's top edge
    top = llc + ver - origin
    assert rd[0][1] >= top[1] and rd[0][1] <= top[1] + ver[1] / 100 + 1e-4
    assert rd[2][1] >= (llc + 0.01 * ver - origin)[1] - 1e-4
    # u = +x of the image maps to -x of the world for this camera (u = up x w with w = -z)
    assert rd[1][0] < rd[0][0]


def test_scene_level_facts():
    sc = orc.OracleScene.from_file(orc.asset("Cornell_Box.json"))
    assert sc.num_shapes() == 36 and list(sc.important()) == [10, 11]  # the two light triangles
    boxes, meta, leaf = sc.bvh()
    assert sorted(leaf.tolist()) == list(range(36))
    assert np.all(boxes[:, :3] <= 0) and np.all(boxes[:, 3:] >= 0)  # every node box contains the origin (aggregate.rs:423-428)
    s2 = orc.OracleScene.from_file(orc.asset("Cornell_Box_With_Sphere.json"))
    assert list(s2.important()) == [10, 11, 24]  # light + glass sphere (Dielectric is "important")
    s3 = orc.OracleScene.from_file(orc.asset("Materials.json"))
    assert len(s3.important()) == 0  # use_importance_sampling: false
    # closest hit straight down the Cornell box axis: back wall at z = 555
    p, t = sc.intersect([[278, 450, -800]], [[0, 0, 1]])  # above both boxes
    assert p[0] >= 0 and t[0] == pytest.approx(1355.0, rel=1e-6)


def _medium_scene(densities):
    """A unit sphere at the origin bounding a medium of densities[0], which bounds a medium of densities[1], and so on."""
    shape = {"type": "sphere", "radius": 1.0, "material": 0, "transform": ([0.0, 0.0, 0.0], [0.0, 0.0, 0.0], [1.0, 1.0, 1.0])}
    for rho in densities:
        shape = {"type": "medium", "boundary": shape, "density": rho, "phase": 1}
    return orc.OracleScene({
        "textures": [{"type": 0, "color": (0.5, 0.5, 0.5)}],
        "materials": [{"type": 0, "albedo": 0, "bump": -1, "param": 0.0}, {"type": 4, "albedo": 0, "bump": -1, "param": 0.0}],
        "shapes": [shape], "aggregate": 0,
        "camera": {"position": [0, 0, 5], "look_at": [0, 0, 0], "up": [0, 1, 0], "fov": 40, "aspect_ratio": 1.0, "aperture": 0.0, "focus_distance": 1.0},
        "logistics": {"resolution_x": 8, "resolution_y": 8, "samples": 1, "use_importance_sampling": False}})


def _reference_medium(rng, n, t_in, t_out, densities, t_min, t_max):
    """ConstantMedium::hit (volume.rs:65-102) for n independent questions, in float64 with numpy's generator; densities[0] is
    the innermost medium's, whose boundary answers t_in, then t_out.  Returns distances, NaN where the answer is None."""
    def ask(level, lo, hi):
        if level < 0:  # the sphere: first root inside (lo, hi), else the second
            first = np.where((t_in > lo) & (t_in < hi), t_in, np.where((t_out > lo) & (t_out < hi), t_out, np.nan))
            return np.broadcast_to(first, (n,)).astype(np.float64)
        big = np.finfo(np.float32).max
        t1 = ask(level - 1, np.full(n, -big), np.full(n, big))
        t2 = ask(level - 1, t1 + np.finfo(np.float32).eps, np.full(n, big))
        t1 = np.maximum(t1, lo)
        t2 = np.minimum(t2, hi)
        ok = t1 < t2  # False where either is NaN
        t1 = np.maximum(t1, 0.0)
        dist = -np.log(rng.random(n)) / densities[level]  # unit direction
        return np.where(ok & (dist <= t2 - t1), t1 + dist, np.nan)
    return ask(len(densities) - 1, np.full(n, t_min), np.full(n, t_max))


@pytest.mark.parametrize("densities", [(0.8,), (3.0, 0.6), (1.5, 2.0, 0.9)])
def test_constant_medium_distances_follow_the_reference_algorithm(densities):
    """The reference draws from a generator that cannot be seeded (volume.rs:95), so a medium's answers are pinned as a
    distribution: how often a ray through the sphere's centre is stopped, and the mean and spread of where, against the
    algorithm written out independently (float64, numpy's generator) - for a plain medium also against the closed form
    1 - exp(-density * chord).  A medium that bounds a medium is asked twice per question put to the outer one and must draw
    afresh each time."""
    sc = _medium_scene(densities)
    n = 200000
    o = np.tile(np.float32([0, 0, 3]), (n, 1))
    d = np.tile(np.float32([0, 0, -1]), (n, 1))
    prim, t = sc.intersect(o, d, seed=11)
    hit = prim >= 0
    assert np.all(prim[hit] == 0) and np.all((t[hit] >= 2.0) & (t[hit] <= 4.0))
    ref = _reference_medium(np.random.default_rng(5), 1000000, 2.0, 4.0, densities, 0.001, float(np.finfo(np.float32).max))
    p_ref = np.mean(~np.isnan(ref))
    if len(densities) == 1:
        assert abs(p_ref - (1.0 - math.exp(-densities[0] * 2.0))) < 2e-3
    sigma = math.sqrt(p_ref * (1 - p_ref) * (1 / n + 1e-6))
    assert abs(hit.mean() - p_ref) < 4.5 * sigma, (hit.mean(), p_ref)
    spread = np.nanstd(ref)
    assert abs(t[hit].mean() - np.nanmean(ref)) < 4.5 * spread * math.sqrt(1 / hit.sum() + 1e-6)
    assert abs(t[hit].std() - spread) < 0.01
    # every ray drew for itself: a repeated batch repeats, another seed does not
    prim2, t2 = sc.intersect(o, d, seed=11)
    assert np.array_equal(prim, prim2) and np.array_equal(t, t2)
    prim3, _ = sc.intersect(o, d, seed=12)
    assert (prim3 != prim).mean() > 0.05


def test_resolve_saturates():
    img = np.zeros((1, 4, 3), dtype=np.float32)
    img[0, 0] = [15.0, 0.25, 1.0]
    img[0, 1] = [np.nan, -1.0, 0.0]
    img[0, 2] = [0.5, 0.5, 0.5]
    out = orc.resolve_rgb8(img, 1)
    assert out[0, 0].tolist() == [255, 127, 255]  # (sqrt(15)*255.99 = 991) saturates; sqrt(.25)*255.99 = 127.99 -> 127
    assert out[0, 1].tolist() == [0, 0, 0]  # NaN -> 0
    assert out[0, 2].tolist() == [181, 181, 181]
