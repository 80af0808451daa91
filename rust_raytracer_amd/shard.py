"""Multi-GPU sharding of one frame: one process per GPU, scene replicated, one collective at the end.

Two partitionings (both leave every (pixel, sample) with the same random stream as a single-GPU render):
* sample ranges — rank r renders samples [r*spp, (r+1)*spp) of every pixel; partial sums are added with one reduce.
* interleaved 32x32 tiles — rank r renders tiles whose index % world == r; every pixel is written by exactly one rank, so
  the frame is reassembled by ONE gather of each rank's own pixels, packed (an N-th of the frame per rank), which rank 0
  puts back in place: bit for bit the single-GPU frame.  (A sum-reduce of frames that are zero elsewhere gives the same
  frame, x + 0 == x, but moves N whole frames: reduce_frame() is for sample ranges.)
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


_PIXELS = {}


def tile_pixels(width, height, world, device):
    """Per rank, the flat indices (y * width + x, ascending) of the pixels that rank owns under interleaved 32x32 tiles, as int64
    tensors on `device`; cached."""
    import torch

    key = (width, height, world, str(device))
    if key not in _PIXELS:
        _PIXELS[key] = [torch.from_numpy(np.flatnonzero(tile_owner_mask(width, height, world, r))).to(device) for r in range(world)]
    return _PIXELS[key]


def gather_tiles(frame, dst=0):
    """Reassembles a tile-sharded [H, W, 3] frame on rank `dst`: every rank sends its own pixels, packed (padded to the largest
    share so that the collective's pieces are equal), rank `dst` puts them in place.  In place; one collective.  RCCL on GPU
    tensors, gloo on CPU tensors (and for GPU tensors of ranks that share a GPU, through the host)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return frame
    world, rank = dist.get_world_size(), dist.get_rank()
    H, W = frame.shape[0], frame.shape[1]
    via_host = frame.is_cuda and dist.get_backend() == "gloo"
    work = frame.cpu() if via_host else frame
    pixels = tile_pixels(W, H, world, work.device)
    most = max(len(p) for p in pixels)
    flat = work.view(-1, 3)
    mine = torch.zeros((most, 3), dtype=work.dtype, device=work.device)
    mine[: len(pixels[rank])] = flat.index_select(0, pixels[rank])
    pieces = [torch.empty_like(mine) for _ in range(world)] if rank == dst else None
    dist.gather(mine, pieces, dst=dst)
    if rank == dst:
        for r in range(world):
            if r != dst:
                flat.index_copy_(0, pixels[r], pieces[r][: len(pixels[r])])
        if via_host:
            frame.copy_(work)
    return frame
