"""Multi-GPU framebuffer partition (no reference counterpart).

One process per GPU.  The frame is cut into tile_w x tile_h tiles numbered row-major; rank r owns
tile t iff t % world == r (interleaved, because cost per pixel is very uneven).  Each rank renders
only its tiles (rsrt_set_partition) into a zero-initialised accumulator, then ONE RCCL
reduce(sum) brings the W*H*4 f32 accumulators to rank 0.  Every pixel has exactly one non-zero
contributor, so x + 0 + ... + 0 is exact and the N-GPU image is bit-identical to the 1-GPU image
whatever the reduction order.
"""
import numpy as np

TILE_W = 16
TILE_H = 16


def tile_grid(width, height, tile_w=TILE_W, tile_h=TILE_H):
    return (width + tile_w - 1) // tile_w, (height + tile_h - 1) // tile_h


def tile_owner_map(width, height, world, tile_w=TILE_W, tile_h=TILE_H):
    """[H, W] int32: owning rank of every pixel."""
    tx, _ = tile_grid(width, height, tile_w, tile_h)
    ys, xs = np.mgrid[0:height, 0:width]
    return (((ys // tile_h) * tx + xs // tile_w) % world).astype(np.int32)


def owned_mask(width, height, rank, world, tile_w=TILE_W, tile_h=TILE_H):
    """[H, W] bool: the pixels `rank` renders — the library's own arithmetic (rsrt_partition_mask, pure host code)."""
    import ctypes as C

    from . import state
    mask = np.zeros((height, width), np.uint8)
    n = C.c_uint64(0)
    rc = state.lib().rsrt_partition_mask(width, height, tile_w, tile_h, rank, world, mask.ctypes.data_as(C.c_void_p), C.byref(n))
    if rc != 0:
        raise ValueError("rsrt_partition_mask: bad arguments")
    assert int(mask.sum()) == n.value
    return mask.astype(bool)


def reduce_accumulators(tensor, dst=0, group=None):
    """In-place sum of per-rank accumulators onto rank `dst` through torch.distributed — the CPU (gloo) rehearsal of the
    exchange step and bench.py's fallback; the product's reduce is rsrt_comm_reduce (RCCL inside librsrt)."""
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.reduce(tensor, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return tensor
