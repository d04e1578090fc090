"""Multi-GPU framebuffer partition (no reference counterpart).

One process per GPU.  The frame is cut into tile_w x tile_h tiles; rank r owns tile (tx, ty) iff
(tx + ty * skew) % world == r — interleaved in x (cost per pixel is very uneven), every tile row shifted by
`skew` against the row above so that a rank's tiles form a lattice whatever the frame width (t % world gives
every rank fixed column stripes whenever the tiles per row are a multiple of world: 1920 / 16 = 120, 8 GPUs).
Each rank renders only its tiles (rsrt_set_partition) into a zero-initialised accumulator.  Every pixel has
exactly one owner, so the sum of the accumulators is a GATHER of every rank's own tiles: the exchange step
(rsrt_comm_reduce, RCCL inside librsrt) packs them into a compact buffer — `tile_slots` per rank, the same
number for all ranks — sends that (1 / world of the frame) to the root and scatters it there.  Nothing is
added, so the N-GPU image is bit-identical to the 1-GPU image by construction.

The numpy functions here restate the library's arithmetic (rsrt_partition_owner / _mask / _tiles, pure host
code of librsrt.so) for the CPU tests and the gloo rehearsal.
"""
import math

import numpy as np

TILE_W = 16
TILE_H = 16


def tile_grid(width, height, tile_w=TILE_W, tile_h=TILE_H):
    return (width + tile_w - 1) // tile_w, (height + tile_h - 1) // tile_h


def skew(world):
    """The smallest odd number >= 3 with no factor in common with world."""
    s = 3
    while math.gcd(s, world) != 1:
        s += 2
    return s


def tile_owner_map(width, height, world, tile_w=TILE_W, tile_h=TILE_H):
    """[H, W] int32: owning rank of every pixel."""
    ys, xs = np.mgrid[0:height, 0:width]
    return ((xs // tile_w + (ys // tile_h) * skew(world)) % world).astype(np.int32)


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


def tile_slots(width, height, rank, world, tile_w=TILE_W, tile_h=TILE_H):
    """[n_slots, 2] int64 (tx, ty) of rank's tile slots, -1 for padding — the library's own list (rsrt_partition_tiles)."""
    import ctypes as C

    from . import state
    L = state.lib()
    n = C.c_uint32(0)
    if L.rsrt_partition_tiles(width, height, tile_w, tile_h, rank, world, None, C.byref(n)) != 0:
        raise ValueError("rsrt_partition_tiles: bad arguments")
    out = np.zeros((n.value, 2), np.uint32)
    if L.rsrt_partition_tiles(width, height, tile_w, tile_h, rank, world, out.ctypes.data_as(C.c_void_p), C.byref(n)) != 0:
        raise ValueError("rsrt_partition_tiles: bad arguments")
    res = out.astype(np.int64)
    res[out == 0xFFFFFFFF] = -1
    return res


def tile_slots_numpy(width, height, rank, world, tile_w=TILE_W, tile_h=TILE_H):
    """The same list from the formula: slot j of rank r is tile (((r - ty * skew) mod world) + k * world, ty) with
    ty = j // per_row, k = j % per_row, per_row = ceil(tiles_x / world)."""
    tiles_x, tiles_y = tile_grid(width, height, tile_w, tile_h)
    per_row, s = (tiles_x + world - 1) // world, skew(world)
    out = np.full((tiles_y * per_row, 2), -1, np.int64)
    for j in range(tiles_y * per_row):
        ty, k = divmod(j, per_row)
        tx = (rank - ty * s) % world + k * world
        if tx < tiles_x:
            out[j] = (tx, ty)
    return out


def pack_tiles(image, rank, world, tile_w=TILE_W, tile_h=TILE_H):
    """rank's compact buffer [n_slots, tile_h, tile_w, C] of an [H, W, C] image (zeros in padding / outside the frame):
    what rt_pack_tiles_kernel produces."""
    h, w = image.shape[:2]
    slots = tile_slots(w, h, rank, world, tile_w, tile_h)
    out = np.zeros((len(slots), tile_h, tile_w) + image.shape[2:], image.dtype)
    for j, (tx, ty) in enumerate(slots):
        if tx < 0:
            continue
        blk = image[ty * tile_h:(ty + 1) * tile_h, tx * tile_w:(tx + 1) * tile_w]
        out[j, :blk.shape[0], :blk.shape[1]] = blk
    return out


def unpack_tiles(buffers, width, height, tile_w=TILE_W, tile_h=TILE_H):
    """The frame from the `world` compact buffers (list index = rank): what rt_unpack_tiles_kernel does on the root."""
    world = len(buffers)
    frame = np.zeros((height, width) + buffers[0].shape[3:], buffers[0].dtype)
    for rank, buf in enumerate(buffers):
        for j, (tx, ty) in enumerate(tile_slots(width, height, rank, world, tile_w, tile_h)):
            if tx < 0:
                continue
            y0, x0 = ty * tile_h, tx * tile_w
            hh, ww = min(tile_h, height - y0), min(tile_w, width - x0)
            frame[y0:y0 + hh, x0:x0 + ww] = buf[j, :hh, :ww]
    return frame


def reduce_accumulators(tensor, dst=0, group=None):
    """In-place sum of per-rank accumulators onto rank `dst` through torch.distributed — the dense form of the exchange
    step (RSRT_COMM_MODE=reduce) and bench.py's fallback; the product's exchange is rsrt_comm_reduce (RCCL inside librsrt)."""
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.reduce(tensor, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return tensor


def gather_tiles(image, rank, world, dst=0, group=None, tile_w=TILE_W, tile_h=TILE_H):
    """The product's exchange step rehearsed over torch.distributed (gloo on CPU): every rank packs its tiles of `image`
    ([H, W, C] numpy), the compact buffers are gathered on `dst`, which returns the frame (other ranks: None)."""
    import torch
    import torch.distributed as dist
    mine = torch.from_numpy(pack_tiles(image, rank, world, tile_w, tile_h))
    bufs = [torch.empty_like(mine) for _ in range(world)] if rank == dst else None
    dist.gather(mine, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return unpack_tiles([b.numpy() for b in bufs], image.shape[1], image.shape[0], tile_w, tile_h)
