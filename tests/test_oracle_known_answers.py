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


def test_resolve_saturates():
    img = np.zeros((1, 4, 3), dtype=np.float32)
    img[0, 0] = [15.0, 0.25, 1.0]
    img[0, 1] = [np.nan, -1.0, 0.0]
    img[0, 2] = [0.5, 0.5, 0.5]
    out = orc.resolve_rgb8(img, 1)
    assert out[0, 0].tolist() == [255, 127, 255]  # (sqrt(15)*255.99 = 991) saturates; sqrt(.25)*255.99 = 127.99 -> 127
    assert out[0, 1].tolist() == [0, 0, 0]  # NaN -> 0
    assert out[0, 2].tolist() == [181, 181, 181]
