"""rrt_render_device_noalloc (the render entry point that allocates nothing) against rrt_render_device, through the C ABI.

Two calls in a row on one scene handle, the second on a smaller frame, so that the second finds the first call's records in
the shared staging area; then a frame whose samples need several passes of the area, a tile-sharded call, and a frame beyond
the area's limit.  Bit-exact against the allocating entry point (which other tests hold bit-exact against the oracle).
"""
import ctypes as C

import numpy as np
import pytest

import orc
import rust_raytracer_amd as rrt

pytestmark = pytest.mark.gpu


def bits(t):
    return t.cpu().numpy().view(np.uint32)


def render_both(torch, scene, W, H, spp, **kw):
    dev = torch.device("cuda", 0)
    out = []
    for noalloc in (False, True):
        img = torch.full((H, W, 3), 7.0, dtype=torch.float32, device=dev)  # stale values the render must replace
        cnt = torch.zeros(2, dtype=torch.int64, device=dev)
        scene.render_device(img.data_ptr(), cnt.data_ptr(), torch.cuda.current_stream().cuda_stream, width=W, height=H, samples=spp,
                            out_floats=img.numel(), noalloc=noalloc, **kw)
        torch.cuda.synchronize()
        out.append((img, cnt))
    return out


def test_noalloc_render_equals_the_allocating_render(gpu):
    import torch

    torch.cuda.set_device(0)
    scene = rrt.Scene.load(orc.asset("Cornell_Box_With_Sphere.json"), width=96, height=80, samples=24)
    # first call: 96 x 80 @ 24; second call, at once, on a smaller frame with fewer samples
    for W, H, spp, seed in ((96, 80, 24, 5), (40, 24, 5, 6)):
        (ref, ref_cnt), (got, got_cnt) = render_both(torch, scene, W, H, spp, seed=seed)
        same = (bits(got) == bits(ref)) | (torch.isnan(got) & torch.isnan(ref)).cpu().numpy()
        assert same.all(), "%d of %d values differ at %dx%d" % ((~same).sum(), same.size, W, H)
        assert torch.equal(got_cnt, ref_cnt) and int(got_cnt[1]) == W * H * spp
    # the smaller frame against the oracle as well
    o = orc.OracleScene.from_file(orc.asset("Cornell_Box_With_Sphere.json"), {"resolution_x": 40, "resolution_y": 24, "samples": 5})
    ref_o, st_o = o.render(40, 24, 5, seed=6)
    assert np.array_equal(bits(got), ref_o.view(np.uint32)) and int(got_cnt[0]) == int(st_o[0])


def test_noalloc_render_allocates_nothing_and_splits_large_frames(gpu):
    import torch

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    # 1024 x 768 x 12 B = 9.4 MB per sample: 20 samples need two passes of the 128 MiB area (13 samples fit)
    W, H, spp = 1024, 768, 20
    scene = rrt.Scene.load(orc.asset("Materials.json"), width=W, height=H, samples=spp)
    img = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    ref = torch.zeros_like(img)
    cnt = torch.zeros(2, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    kw = dict(width=W, height=H, samples=spp, seed=3, background=rrt.BG_SKY)
    scene.render_device(img.data_ptr(), cnt.data_ptr(), stream, noalloc=True, **kw)  # warm-up: code object load, first launch
    torch.cuda.synchronize()
    free_before, _ = torch.cuda.mem_get_info()
    for _ in range(3):
        scene.render_device(img.data_ptr(), cnt.data_ptr(), stream, noalloc=True, **kw)
    torch.cuda.synchronize()
    free_after, _ = torch.cuda.mem_get_info()
    assert free_after == free_before, "device memory in use changed by %d bytes across no-alloc renders" % (free_before - free_after)
    scene.render_device(ref.data_ptr(), 0, stream, **kw)
    torch.cuda.synchronize()
    assert np.array_equal(bits(img), bits(ref))
    # interleaved tiles: each call touches only its own pixels
    tiled = torch.zeros_like(img)
    for k in range(3):
        scene.render_device(tiled.data_ptr(), 0, stream, noalloc=True, tile_count=3, tile_index=k, **kw)
    torch.cuda.synchronize()
    assert np.array_equal(bits(tiled), bits(ref))


def test_noalloc_render_refuses_frames_beyond_the_staging_area(gpu):
    import torch

    torch.cuda.set_device(0)
    scene = rrt.Scene.load(orc.asset("Cornell_Box.json"), width=64, height=64, samples=1)
    W, H = 4096, 2731  # 11 186 176 pixels: 1 366 more than fit
    img = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
    with pytest.raises(rrt.RrtError) as e:
        scene.render_device(img.data_ptr(), 0, 0, width=W, height=H, samples=1, noalloc=True)
    assert e.value.code == 4 and "staging" in e.value.message
    scene.render_device(img.data_ptr(), 0, 0, width=W, height=2730, samples=1, noalloc=True)  # 4096 x 2730 just fits
    torch.cuda.synchronize()
    assert torch.isfinite(img[:2730]).all()
