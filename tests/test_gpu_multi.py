"""rrt_render_multi (several scene handles, one host thread each, shares gathered on the first handle's device) through the C
ABI.  A one-GPU box can still run every line of it: two or three handles on the same device take the multi-device path (own
frame and stream per handle, copy to the first device's landing buffer, add in handle order); with several GPUs the handles go to
different devices."""
import numpy as np
import pytest

import orc
import rust_raytracer_amd as rrt

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def handles(name, n, **ov):
    host = rrt.HostScene.load(orc.asset(name), width=ov["W"], height=ov["H"], samples=ov["spp"])
    ndev = rrt.device_count()
    return [rrt.Scene(host, device=i % ndev) for i in range(n)]


def test_one_handle_equals_rrt_render(gpu):
    W, H, spp = 72, 56, 12
    (g,) = handles("Cornell_Box_With_Sphere.json", 1, W=W, H=H, spp=spp)
    ref, st_ref = g.render(seed=4)
    for shard in (rrt.SHARD_SAMPLES, rrt.SHARD_TILES):
        img, st = rrt.render_multi([g], shard=shard, seed=4)
        same = (bits(img) == bits(ref)) | (np.isnan(img) & np.isnan(ref))
        assert same.all() and st["rays"] == st_ref["rays"] and st["paths"] == W * H * spp
        assert st["seconds"] > 0 and st["trace_seconds"] > 0 and st["accumulate_seconds"] > 0 and st["copy_seconds"] > 0
    assert st_ref["trace_seconds"] > 0 and st_ref["seconds"] >= st_ref["trace_seconds"]


@pytest.mark.parametrize("n", [2, 3])
def test_tiles_over_several_handles_are_bit_identical_to_one(gpu, n):
    W, H, spp = 100, 70, 9  # 4 x 3 tiles of 32 x 32, ragged at both edges
    gs = handles("Cornell_Box_With_Sphere.json", n, W=W, H=H, spp=spp)
    ref, st_ref = gs[0].render(seed=7)
    img = np.full((H, W, 3), 123.0, dtype=np.float32)  # stale values: every pixel must be overwritten
    img, st = rrt.render_multi(gs, shard=rrt.SHARD_TILES, seed=7, out=img)
    same = (bits(img) == bits(ref)) | (np.isnan(img) & np.isnan(ref))
    assert same.all(), "%d values differ" % (~same).sum()
    assert st["rays"] == st_ref["rays"] and st["paths"] == W * H * spp and st["reduce_seconds"] > 0


@pytest.mark.parametrize("n,spp", [(2, 10), (3, 10), (3, 2)])
def test_sample_ranges_over_several_handles(gpu, n, spp):
    """Handle i traces a contiguous range of the samples (the remainder goes to the first handles; with fewer samples than handles
    the last ones trace nothing); the frame is the f32 sum of the partial frames in handle order - compared bit for bit with
    that sum formed from separate single-handle renders of the same ranges - and the oracle's frame up to that re-association."""
    W, H = 64, 48
    gs = handles("Cornell_Box_With_Sphere.json", n, W=W, H=H, spp=spp)
    img, st = rrt.render_multi(gs, shard=rrt.SHARD_SAMPLES, seed=9, samples=spp)
    base, extra = divmod(spp, n)
    expect, first, rays = None, 0, 0
    for i in range(n):
        mine = base + (1 if i < extra else 0)
        if mine:
            part, st_part = gs[0].render(samples=mine, first_sample=first, seed=9)
            rays += st_part["rays"]
        else:
            part = np.zeros((H, W, 3), dtype=np.float32)
        expect = part if expect is None else expect + part
        first += mine
    same = (bits(img) == bits(expect)) | (np.isnan(img) & np.isnan(expect))
    assert same.all(), "%d values differ" % (~same).sum()
    assert st["rays"] == rays and st["paths"] == W * H * spp
    o = orc.OracleScene.from_file(orc.asset("Cornell_Box_With_Sphere.json"), {"resolution_x": W, "resolution_y": H, "samples": spp})
    ref, st_ref = o.render(W, H, spp, seed=9)
    ok = np.isfinite(ref) & np.isfinite(img)
    assert st["rays"] == int(st_ref[0]) and np.allclose(img[ok], ref[ok], rtol=1e-5, atol=1e-6)  # same samples, sums re-associated


def test_accumulate_and_argument_checks(gpu):
    W, H = 40, 40
    gs = handles("Cornell_Box.json", 2, W=W, H=H, spp=6)
    a, _ = rrt.render_multi(gs, shard=rrt.SHARD_TILES, seed=2, samples=4)
    b, _ = rrt.render_multi(gs, shard=rrt.SHARD_TILES, seed=2, samples=2, first_sample=4, out=a.copy(), accumulate=True)
    ref, _ = gs[0].render(seed=2, samples=6)
    same = (bits(b) == bits(ref)) | (np.isnan(b) & np.isnan(ref))
    assert same.all()
    with pytest.raises(rrt.RrtError):
        rrt.render_multi([gs[0], gs[0]], seed=1)  # one render at a time per handle
    with pytest.raises(rrt.RrtError):
        rrt.render_multi(gs, shard=5)


def test_buffers_kept_between_calls_more_handles_than_tiles_and_mismatched_handles(gpu):
    """The handles keep their frames, landing buffers, streams and events from call to call: a larger frame after a smaller one (the
    buffers grow), then a smaller one again (they are reused), each bit-identical to one handle; a frame of ONE tile over three
    handles (two of them own nothing and push nothing); and handles that hold different scenes are refused."""
    gs = handles("Cornell_Box_With_Sphere.json", 3, W=48, H=40, spp=5)
    for W, H, spp, shard in ((48, 40, 5, rrt.SHARD_TILES), (130, 97, 3, rrt.SHARD_TILES), (130, 97, 3, rrt.SHARD_SAMPLES), (33, 20, 4, rrt.SHARD_TILES),
                             (32, 32, 6, rrt.SHARD_TILES)):
        img, st = rrt.render_multi(gs, shard=shard, seed=11, width=W, height=H, samples=spp)
        ref, st_ref = gs[0].render(seed=11, width=W, height=H, samples=spp)
        assert st["rays"] == st_ref["rays"] and st["paths"] == W * H * spp
        if shard == rrt.SHARD_TILES:
            same = (bits(img) == bits(ref)) | (np.isnan(img) & np.isnan(ref))
            assert same.all(), "%d values differ at %dx%d" % ((~same).sum(), W, H)
        else:
            ok = np.isfinite(ref) & np.isfinite(img)
            assert np.allclose(img[ok], ref[ok], rtol=1e-5, atol=1e-6)
    other = handles("Cornell_Box.json", 1, W=48, H=40, spp=5)
    with pytest.raises(rrt.RrtError) as e:
        rrt.render_multi([gs[0], other[0]], seed=1)
    assert "same scene" in e.value.message


def test_bench_with_two_ranks_sharing_the_gpu(gpu):
    """bench.py's N > 1 path on real hardware: two ranks under torch.distributed.run on the box's single GPU (they share it and talk
    over gloo - a rehearsal, it measures nothing): render on its own stream, event, collective, resolve, and the strong-scaling
    figures (sample ranges and tiles) must come out, with the ray count of ONE frame for the strong cases."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29517",
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--spp", "32"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=170, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1
    rec = json.loads(line[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and "rehearsal" in rec["config"]
    assert rec["value"] > 0 and rec["roofline"]["kernel"] == "k_trace_pool"
    assert set(rec["strong"]) == {"samples", "tiles"} and all(v["value"] > 0 and v["efficiency_vs_1gpu"] > 0 for v in rec["strong"].values())
