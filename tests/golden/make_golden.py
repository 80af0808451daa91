#!/usr/bin/env python3
"""Regenerates the committed fixtures under tests/golden/.

* gallery_stats.json — numbers measured on the reference's own gallery render img/cornell_box.png (the only
  reference-produced output usable as a pin; needs /root/reference, so it is only regenerated where that exists).
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


def main():
    if os.path.exists("/root/reference/img/cornell_box.png"):
        with open(os.path.join(HERE, "gallery_stats.json"), "w") as f:
            json.dump(gallery_stats(), f)
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
