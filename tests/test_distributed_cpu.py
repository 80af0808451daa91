"""The N > 1 path on CPU: two gloo ranks shard one frame (sample ranges, interleaved tiles) and combine it with
the single collective the GPU job uses (a sum for sample ranges, a gather of each rank's own pixels for tiles).  The per-rank renders come from the CPU oracle; what is under test is the sharding
arithmetic and the collective, which are the same code bench.py runs over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
W, H, SPP = 80, 72, 4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import orc
    from rust_raytracer_amd import shard

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = orc.OracleScene.from_file(orc.asset("Cornell_Box_With_Sphere.json"), {"resolution_x": W, "resolution_y": H, "samples": SPP})
    if mode == "samples":
        first, n = shard.sample_range(rank, SPP)
        img, _ = sc.render(W, H, n, first_sample=first, seed=3, threads=2)
    else:
        full, _ = sc.render(W, H, SPP, seed=3, threads=2)
        # (pixels of other ranks' tiles hold stale values, as a tile-sharded GPU render leaves them)
        img = np.where(shard.tile_owner_mask(W, H, world, rank)[..., None], full, np.float32(77)).astype(np.float32)
    frame = torch.from_numpy(np.nan_to_num(img))
    if mode == "tiles":
        shard.gather_tiles(frame, dst=0)
    else:
        shard.reduce_frame(frame, dst=0)
    if rank == 0:
        np.save(os.path.join(out_dir, "frame_%s.npy" % mode), frame.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["samples", "tiles"])
def test_two_rank_sharding(tmp_path, mode):
    sys.path.insert(0, HERE)
    import orc
    from rust_raytracer_amd import shard

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), mode, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / ("frame_%s.npy" % mode))
    sc = orc.OracleScene.from_file(orc.asset("Cornell_Box_With_Sphere.json"), {"resolution_x": W, "resolution_y": H, "samples": SPP})
    if mode == "samples":
        a, _ = sc.render(W, H, SPP, first_sample=0, seed=3)
        b, _ = sc.render(W, H, SPP, first_sample=SPP, seed=3)
        expect = np.nan_to_num(a) + np.nan_to_num(b)
        # and the two ranges together are the 2*SPP-sample render (same streams), up to the order of the final addition
        both, _ = sc.render(W, H, 2 * SPP, seed=3)
        assert np.allclose(np.nan_to_num(both), expect, rtol=1e-5, atol=1e-6)
    else:
        full, _ = sc.render(W, H, SPP, seed=3)
        expect = np.nan_to_num(full)
        masks = [shard.tile_owner_mask(W, H, world, r) for r in range(world)]
        assert np.array_equal(masks[0] ^ masks[1], np.ones((H, W), bool))  # a partition of the frame
    assert np.array_equal(got.view(np.uint32), expect.view(np.uint32))
