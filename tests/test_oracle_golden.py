"""The oracle against (a) what the reference itself produced — the gallery render img/cornell_box.png, recorded as
statistics in golden/gallery_stats.json — and (b) its own committed ray/render fixtures (regression pins)."""
import json
import os

import numpy as np
import pytest

import orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SCENES = {"Cornell_Box": orc.BG_BLACK, "Cornell_Box_With_Sphere": orc.BG_BLACK, "Materials": orc.BG_SKY, "Example_Spec": orc.BG_SKY,
          "Globe": orc.BG_SKY}


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("name", list(SCENES))
def test_ray_fixture(name):
    g = np.load(os.path.join(GOLD, "rays_%s.npz" % name))
    sc = orc.OracleScene.from_file(orc.asset(name + ".json"), {"resolution_x": 64, "resolution_y": 64})
    prim, t = sc.intersect(g["o"], g["d"])
    assert np.array_equal(prim, g["prim"]) and np.array_equal(bits(t), bits(g["t"]))
    # the complete slab test never changes an answer, it only visits fewer nodes
    prim2, t2, st2 = sc.intersect(g["o"], g["d"], tight=True, want_stats=True)
    _, _, st = sc.intersect(g["o"], g["d"], want_stats=True)
    assert np.array_equal(prim2, prim) and np.array_equal(bits(t2), bits(t)) and st2[1] <= st[1]


@pytest.mark.parametrize("name", list(SCENES))
def test_render_fixture(name):
    ref = np.load(os.path.join(GOLD, "render_%s.npy" % name))
    sc = orc.OracleScene.from_file(orc.asset(name + ".json"), {"resolution_x": 64, "resolution_y": 64, "samples": 8})
    img, _ = sc.render(64, 64, 8, seed=2024, bg=SCENES[name])
    same = (bits(img) == bits(ref)) | (np.isnan(img) & np.isnan(ref))
    assert same.all()
    # independent of the thread count and of the AABB variant
    img1, _ = sc.render(64, 64, 8, seed=2024, bg=SCENES[name], threads=1, flags=orc.FLAG_TIGHT_AABB)
    assert ((bits(img1) == bits(ref)) | (np.isnan(img1) & np.isnan(ref))).all()


def test_recursive_and_loop_trace_agree():
    """The loop form multiplies throughput down the path; the reference multiplies radiance up the recursion
    (aggregate.rs:46-89).  Same samples, same decisions: per-pixel sums agree to f32 rounding of the products."""
    sc = orc.OracleScene.from_file(orc.asset("Cornell_Box_With_Sphere.json"), {"resolution_x": 64, "resolution_y": 64, "samples": 16})
    a, sa = sc.render(64, 64, 16, seed=9)
    b, sb = sc.render(64, 64, 16, seed=9, flags=orc.FLAG_RECURSIVE)
    assert int(sa[0]) == int(sb[0])  # same number of rays: identical decisions
    ok = ~(np.isnan(a) | np.isnan(b))
    assert np.array_equal(np.isnan(a), np.isnan(b))
    assert np.allclose(a[ok], b[ok], rtol=2e-5, atol=1e-6)


def test_gallery_cornell_box_statistics():
    """500x500 Cornell box as shipped (black background, light sampling on) against the reference's 1000-spp gallery image.
    Light pixels are excluded (the gallery was written by a rustc whose `as u8` wrapped: light = 223), borders must be
    identical, block means must agree within Monte-Carlo noise at 48 spp."""
    with open(os.path.join(GOLD, "gallery_stats.json")) as f:
        g = json.load(f)
    spp = 48
    sc = orc.OracleScene.from_file(orc.asset("Cornell_Box.json"))
    img, _ = sc.render(500, 500, spp, seed=1)
    rgb = orc.resolve_rgb8(img, spp).astype(np.float64)
    zero_rows = [r for r in range(500) if rgb[r].max() == 0]
    zero_cols = [c for c in range(500) if rgb[:, c].max() == 0]
    assert zero_rows == g["zero_rows"] and zero_cols == g["zero_cols"]
    mean = np.nan_to_num(img / spp)
    not_light = mean.max(axis=2) < 1.0
    lin = np.minimum(mean, 1.0)  # the 8-bit image clips at 1
    B = g["block"]
    blocks = (lin * not_light[..., None]).reshape(500 // B, B, 500 // B, B, 3).mean(axis=(1, 3))
    ref_blocks = np.asarray(g["block_means_linear_not_light"])
    assert np.abs(blocks - ref_blocks).max() < 0.02  # worst block straddles the light edge; 0.0026 at 200 spp
    assert np.sqrt(((blocks - ref_blocks) ** 2).mean()) < 0.003
    assert np.allclose(lin[not_light].mean(axis=0), g["mean_linear_not_light"], rtol=0.02)


def sky_gallery_check(rgb8, ref, mean_tol=0.1, block_rmse_tol=1e-3, block_max_tol=0.012, black_rel_tol=None):
    """An 8-bit 500x500 render against the statistics of one of the reference's sky-lit gallery images: mean RGB8 within
    `mean_tol` code values, 25-pixel block means of the linear image (v/255.99)^2 within `block_rmse_tol` (RMSE) and
    `block_max_tol` (worst block), count of exactly-black pixels within `black_rel_tol` (relative)."""
    rgb = rgb8.astype(np.float64)
    assert rgb.shape == (500, 500, 3)
    B = ref["block"]
    assert np.abs(rgb.reshape(-1, 3).mean(axis=0) - np.asarray(ref["mean_rgb8"])).max() < mean_tol
    lin = (rgb / 255.99) ** 2
    blocks = lin.reshape(500 // B, B, 500 // B, B, 3).mean(axis=(1, 3))
    diff = blocks - np.asarray(ref["block_means_linear"])
    assert np.sqrt((diff ** 2).mean()) < block_rmse_tol
    assert np.abs(diff).max() < block_max_tol
    if black_rel_tol is not None:
        black = int((rgb.max(axis=2) == 0).sum())
        assert abs(black - ref["black_pixels"]) <= black_rel_tol * ref["black_pixels"], (black, ref["black_pixels"])


def test_gallery_globe_statistics():
    """assets/Globe.json (sphere uv, image texture, bump map) with the sky background of main.rs:259-267 against the reference's
    img/globe.png (1000 spp): at 64 spp the oracle's mean RGB8 is within 0.006 and the block RMSE 1.7e-4 linear."""
    with open(os.path.join(GOLD, "gallery_stats.json")) as f:
        ref = json.load(f)["globe"]
    spp = 64
    sc = orc.OracleScene.from_file(orc.asset("Globe.json"))
    img, _ = sc.render(500, 500, spp, seed=1, bg=orc.BG_SKY)
    sky_gallery_check(orc.resolve_rgb8(img, spp), ref, mean_tol=0.05, block_rmse_tol=5e-4, block_max_tol=0.006)


def test_gallery_materials_statistics():
    """assets/Materials.json (checker, Lambert, Metal 0.1, Dielectric, the sphere of radius -0.45; importance sampling off) with
    the sky background against the reference's img/materials.png, at its 200 spp: mean RGB8 within 0.015, block RMSE 5.4e-4
    linear, 41 362 exactly-black pixels (the black checker squares) against the reference's 41 366."""
    with open(os.path.join(GOLD, "gallery_stats.json")) as f:
        ref = json.load(f)["materials"]
    spp = 200
    sc = orc.OracleScene.from_file(orc.asset("Materials.json"))
    img, _ = sc.render(500, 500, spp, seed=1, bg=orc.BG_SKY)
    sky_gallery_check(orc.resolve_rgb8(img, spp), ref, mean_tol=0.1, block_rmse_tol=1e-3, block_max_tol=0.012, black_rel_tol=0.002)
