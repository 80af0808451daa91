#!/usr/bin/env python3
"""Regenerates the committed fixtures under tests/golden/.

* gallery_stats.json — numbers measured on the reference's own gallery renders img/cornell_box.png, img/globe.png and
  img/materials.png (the only reference-produced outputs; the last two were made with the sky background that main.rs:259-267
  keeps commented out).  Needs /root/reference, so it is only regenerated where that exists (`--gallery-only` stops there).
* rays_<scene>.npz   — fixed ray batches with the oracle's (prim_id, t) answers.
* render_<scene>.npy — 64x64 oracle renders (radiance sums) at a fixed seed.
The reference has no tests or golden vectors of its own and cannot be built here (no Rust toolchain), so the
ray/render fixtures are produced by the CPU oracle and pin it (and the GPU path) against regressions.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orc  # noqa: E402
import rays as raygen  # noqa: E402

SCENES = {
    "Cornell_Box": (orc.BG_BLACK, 8),
    "Cornell_Box_With_Sphere": (orc.BG_BLACK, 8),
    "Materials": (orc.BG_SKY, 8),
    "Example_Spec": (orc.BG_SKY, 8),
    "Globe": (orc.BG_SKY, 8),
}


def gallery_stats():
    from PIL import Image

    ref = np.asarray(Image.open("/root/reference/img/cornell_box.png").convert("RGB")).astype(np.float64)
    lin = (ref / 255.99) ** 2
    zero_rows = [int(r) for r in range(ref.shape[0]) if ref[r].max() == 0]
    zero_cols = [int(c) for c in range(ref.shape[1]) if ref[:, c].max() == 0]
    B = 25
    not_light = ref.max(axis=2) < 200
    blocks = (lin * not_light[..., None]).reshape(500 // B, B, 500 // B, B, 3).mean(axis=(1, 3))
    return {
        "source": "img/cornell_box.png of the reference (500x500, assets/Cornell_Box.json at 1000 spp, black background)",
        "mean_rgb8": ref.reshape(-1, 3).mean(axis=0).tolist(),
        "mean_linear": lin.reshape(-1, 3).mean(axis=0).tolist(),
        "mean_linear_not_light": lin[not_light].mean(axis=0).tolist(),
        "zero_rows": zero_rows, "zero_cols": zero_cols,
        "block": B, "block_means_linear_not_light": np.round(blocks, 6).tolist(),
        "light_pixel_value": int(ref[70, 250, 0]),
    }


def sky_gallery_stats(name, scene, spp):
    """Statistics of a gallery render made with the sky background the reference keeps commented out (main.rs:259-267):
    means, 25-pixel block means (8-bit and linear) and the number of exactly-black pixels."""
    from PIL import Image

    ref = np.asarray(Image.open("/root/reference/img/%s.png" % name).convert("RGB")).astype(np.float64)
    assert ref.shape == (500, 500, 3)
    lin = (ref / 255.99) ** 2
    B = 25
    return {
        "source": "img/%s.png of the reference (500x500, assets/%s at %d spp, sky background of main.rs:259-267)" % (name, scene, spp),
        "mean_rgb8": ref.reshape(-1, 3).mean(axis=0).tolist(),
        "mean_linear": lin.reshape(-1, 3).mean(axis=0).tolist(),
        "black_pixels": int((ref.max(axis=2) == 0).sum()),
        "block": B,
        "block_means_rgb8": np.round(ref.reshape(500 // B, B, 500 // B, B, 3).mean(axis=(1, 3)), 4).tolist(),
        "block_means_linear": np.round(lin.reshape(500 // B, B, 500 // B, B, 3).mean(axis=(1, 3)), 6).tolist(),
    }


def main():
    if os.path.exists("/root/reference/img/cornell_box.png"):
        stats = gallery_stats()
        stats["globe"] = sky_gallery_stats("globe", "Globe.json", 1000)
        stats["materials"] = sky_gallery_stats("materials", "Materials.json", 200)
        with open(os.path.join(HERE, "gallery_stats.json"), "w") as f:
            json.dump(stats, f)
    if "--gallery-only" in sys.argv:
        return
    for name, (bg, spp) in SCENES.items():
        sc = orc.OracleScene.from_file(orc.asset(name + ".json"), {"resolution_x": 64, "resolution_y": 64, "samples": spp})
        ro, rd = raygen.ray_batch(sc, 64, 64, n_random=768, seed=11)
        prim, t = sc.intersect(ro, rd)
        np.savez_compressed(os.path.join(HERE, "rays_%s.npz" % name), o=ro, d=rd, prim=prim, t=t)
        img, st = sc.render(64, 64, spp, seed=2024, bg=bg)
        np.save(os.path.join(HERE, "render_%s.npy" % name), img)
        print(name, len(ro), "rays", int((prim >= 0).sum()), "hits; render rays", int(st[0]))


if __name__ == "__main__":
    main()
