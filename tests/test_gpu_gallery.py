"""The HIP path against what the reference itself produced: its three gallery renders (Readme.md:5-8), as statistics in
golden/gallery_stats.json, rendered on the GPU at the gallery's resolution and at matched (Materials) or sufficient samples.

Tolerances (stated per check in test_oracle_golden.sky_gallery_check and below): mean RGB8 within 0.1 code values, 25-pixel
block means within 1e-3 RMSE of the linear image (Cornell: 0.0015, its worst block 0.02 at the light's edge), mean linear radiance
of the Cornell box within 0.5 %, black pixel counts within 0.2 %, black border rows / columns identical.
"""
import json
import os

import numpy as np
import pytest

import orc
import rust_raytracer_amd as rrt
from test_oracle_golden import GOLD, sky_gallery_check

pytestmark = pytest.mark.gpu


def stats():
    with open(os.path.join(GOLD, "gallery_stats.json")) as f:
        return json.load(f)


def test_gallery_globe_on_the_gpu(gpu):
    spp = 1000  # assets/Globe.json's own sample count
    g = rrt.Scene.load(orc.asset("Globe.json"))
    assert (g.width, g.height, g.samples) == (500, 500, spp)
    img, _ = g.render(seed=1, background=rrt.BG_SKY)
    sky_gallery_check(rrt.resolve_rgb8(img, spp), stats()["globe"], mean_tol=0.05, block_rmse_tol=3e-4, block_max_tol=0.004)


def test_gallery_materials_on_the_gpu(gpu):
    spp = 200  # assets/Materials.json's own sample count
    g = rrt.Scene.load(orc.asset("Materials.json"))
    assert (g.width, g.height, g.samples) == (500, 500, spp)
    img, _ = g.render(seed=1, background=rrt.BG_SKY)
    sky_gallery_check(rrt.resolve_rgb8(img, spp), stats()["materials"], mean_tol=0.1, block_rmse_tol=1e-3, block_max_tol=0.012,
                      black_rel_tol=0.002)


def test_gallery_cornell_box_on_the_gpu(gpu):
    """assets/Cornell_Box.json as shipped (500 x 500, 1000 spp, black background, light sampling on) against img/cornell_box.png:
    identical black border rows and columns (the one-pixel `res_y - y` shift of main.rs:228 included), block means of the
    non-light pixels within Monte-Carlo noise of two 1000-spp renders."""
    g_ref = stats()
    spp = 1000
    g = rrt.Scene.load(orc.asset("Cornell_Box.json"))
    assert (g.width, g.height, g.samples) == (500, 500, spp)
    img, _ = g.render(seed=1)
    rgb = rrt.resolve_rgb8(img, spp).astype(np.float64)
    assert [r for r in range(500) if rgb[r].max() == 0] == g_ref["zero_rows"]
    assert [c for c in range(500) if rgb[:, c].max() == 0] == g_ref["zero_cols"]
    # through the same 8-bit quantisation as the gallery file; light pixels (the gallery's read 223: its rustc's `as u8` wrapped) are
    # masked by value on both sides, which leaves the antialiased pixels of the light's edge as the worst block (0.013 at 200 spp)
    not_light = rgb.max(axis=2) < 200
    lin = (rgb / 255.99) ** 2
    B = g_ref["block"]
    blocks = (lin * not_light[..., None]).reshape(500 // B, B, 500 // B, B, 3).mean(axis=(1, 3))
    ref_blocks = np.asarray(g_ref["block_means_linear_not_light"])
    assert np.abs(blocks - ref_blocks).max() < 0.02
    assert np.sqrt(((blocks - ref_blocks) ** 2).mean()) < 0.0015  # the oracle at 200 spp: 0.00137
    assert np.allclose(lin[not_light].mean(axis=0), g_ref["mean_linear_not_light"], rtol=0.005)  # the oracle at 200 spp: within 0.002
