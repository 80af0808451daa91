"""Multi-GPU sharding of one frame: one process per GPU, scene replicated, one collective at the end.

Two partitionings (both leave every (pixel, sample) with the same random stream as a single-GPU render):
* sample ranges — rank r renders samples [r*spp, (r+1)*spp) of every pixel; partial sums are added with one reduce.
* interleaved 32x32 tiles — rank r renders tiles whose index % world == r into a zeroed frame; since every pixel is
  written by exactly one rank, the same sum-reduce reassembles the frame bit for bit (x + 0 == x).
"""
import numpy as np

TILE = 32


def sample_range(rank, spp_per_rank):
    """(first_sample, samples) of `rank`."""
    return rank * spp_per_rank, spp_per_rank


def split_samples(rank, world, total_spp):
    """(first_sample, samples) of `rank` when ONE frame of `total_spp` samples per pixel is divided over `world` ranks
    (strong scaling).  The remainder goes to the first ranks, one sample each, so no sample is dropped and the ranges are
    contiguous: rank r starts where rank r-1 ends."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError("rank %d outside a world of %d" % (rank, world))
    base, extra = divmod(total_spp, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def tile_owner_mask(width, height, world, rank):
    """Boolean [H, W] mask of the pixels rank `rank` renders under interleaved 32x32 tiles."""
    tiles_x = (width + TILE - 1) // TILE
    ty, tx = np.meshgrid(np.arange(height) // TILE, np.arange(width) // TILE, indexing="ij")
    return ((ty * tiles_x + tx) % world) == rank


def reduce_frame(frame, dst=0):
    """Sums the per-rank radiance frames onto rank `dst` (RCCL on GPU tensors, gloo on CPU tensors). In place."""
    import torch.distributed as dist

    if dist.is_initialized() and dist.get_world_size() > 1:
        if frame.is_cuda and dist.get_backend() == "gloo":
            # ranks that share a GPU (a rehearsal of the multi-GPU path on a box with fewer GPUs than ranks) talk over gloo, which
            # reduces host tensors: the frame goes through the host, and every rank ends up with the sum
            host = frame.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            frame.copy_(host)
        else:
            dist.reduce(frame, dst=dst, op=dist.ReduceOp.SUM)
    return frame
