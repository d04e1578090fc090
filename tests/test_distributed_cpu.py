"""N>1 host logic on CPU: gloo, world_size 2.  Each rank fills only the pixels it owns (values
from the oracle), one reduce(sum) to rank 0 must reproduce the single-process image bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import util


def _worker(rank, world, port, full_path, out_path):
    sys.path.insert(0, util.ROOT)
    from rsoderh_raytracing_amd import partition
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = np.load(full_path)
    h, w = full.shape[:2]
    mine = np.zeros_like(full)
    m = partition.owned_mask(w, h, rank, world)
    mine[m] = full[m]
    t = torch.from_numpy(mine)
    partition.reduce_accumulators(t)
    if rank == 0:
        np.save(out_path, t.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_tile_partition_reduce_is_bit_exact(tmp_path, world):
    import oracle
    import rsoderh_raytracing_amd as R
    sc = R.Scene.load_toml(util.scene_path("house"))
    img, _ = oracle.render(util.oracle_scene(sc), util.oracle_env(util.small_env()), sc.camera_uniform().view(oracle.CAMERA),
                           72, 40, 0, 2, 4)
    full_path, out_path = str(tmp_path / "full.npy"), str(tmp_path / "out.npy")
    np.save(full_path, img)
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, full_path, out_path), nprocs=world, join=True)
    out = np.load(out_path)
    assert np.array_equal(util.bits(out), util.bits(img))


def test_owner_map_covers_every_pixel_once():
    from rsoderh_raytracing_amd import partition
    for w, h, world in [(1920, 1080, 8), (72, 40, 3), (17, 5, 2), (16, 16, 4)]:
        owner = partition.tile_owner_map(w, h, world)
        assert owner.min() >= 0 and owner.max() < world
        total = sum(partition.owned_mask(w, h, r, world).astype(np.int64) for r in range(world))
        assert np.all(total == 1)
    # balance: tile counts per rank differ by at most one
    tx, ty = partition.tile_grid(1920, 1080)
    counts = np.bincount(np.arange(tx * ty) % 8)
    assert counts.max() - counts.min() <= 1
