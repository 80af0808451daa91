"""A device short of memory: the per-sample staging buffer of a render shrinks to what can be had (more, smaller passes) and
the frame stays the same bits - sample order is restored by the accumulation, whatever the split (main.rs:244-245)."""
import numpy as np
import pytest
import torch

import orc
import rust_raytracer_amd as rrt

pytestmark = pytest.mark.gpu


def test_render_when_the_staging_buffer_does_not_fit(gpu):
    W = H = 512
    spp = 512  # 512 x 512 x 12 B x 512 samples = 1.6 GB of staging in one pass
    g = rrt.Scene.load(orc.asset("Cornell_Box_With_Sphere.json"), width=W, height=H, samples=spp)
    ref, st_ref = g.render(seed=6)
    g.close()
    g = rrt.Scene.load(orc.asset("Cornell_Box_With_Sphere.json"), width=W, height=H, samples=spp)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    keep = 600 << 20  # leave about 0.6 GB: room for 128-sample passes at most
    hog = torch.empty(free - keep, dtype=torch.uint8, device="cuda") if free > keep else None
    try:
        free_now, _ = torch.cuda.mem_get_info()
        assert free_now < W * H * 12 * spp, "the test could not make the device short of memory"
        img, st = g.render(seed=6)
    finally:
        del hog
        torch.cuda.empty_cache()
    assert np.array_equal(np.ascontiguousarray(img).view(np.uint32), np.ascontiguousarray(ref).view(np.uint32))
    assert st["rays"] == st_ref["rays"] and st["paths"] == W * H * spp
