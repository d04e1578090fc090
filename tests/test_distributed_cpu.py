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


def _gather_worker(rank, world, port, full_path, out_path):
    sys.path.insert(0, util.ROOT)
    from rsoderh_raytracing_amd import partition
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = np.load(full_path)
    h, w = full.shape[:2]
    mine = np.full_like(full, np.nan)  # what a rank does not own must never reach the frame
    m = partition.owned_mask(w, h, rank, world)
    mine[m] = full[m]
    frame = partition.gather_tiles(mine, rank, world)
    if rank == 0:
        np.save(out_path, frame)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_compact_tile_gather_is_bit_exact(tmp_path, world):
    """The product's exchange step (pack own tiles -> gather -> scatter on the root), rehearsed over gloo with the
    library's own tile lists (rsrt_partition_tiles): 1 / world of the bytes of the dense reduce, the same frame."""
    import oracle
    import rsoderh_raytracing_amd as R
    sc = R.Scene.load_toml(util.scene_path("house"))
    img, _ = oracle.render(util.oracle_scene(sc), util.oracle_env(util.small_env()), sc.camera_uniform().view(oracle.CAMERA),
                           72, 40, 0, 2, 4)
    full_path, out_path = str(tmp_path / "full.npy"), str(tmp_path / "out.npy")
    np.save(full_path, img)
    port = 31500 + (os.getpid() % 2000) + world
    mp.spawn(_gather_worker, args=(world, port, full_path, out_path), nprocs=world, join=True)
    out = np.load(out_path)
    assert np.array_equal(util.bits(out), util.bits(img))


def test_tile_slots_of_the_library_match_the_formula_and_tile_the_frame():
    from rsoderh_raytracing_amd import partition
    for w, h, world, tw, th in [(1920, 1080, 8, 16, 16), (72, 40, 3, 16, 16), (17, 5, 2, 16, 16), (100, 60, 4, 32, 8), (200, 120, 6, 16, 16), (64, 64, 1, 16, 16)]:
        tiles_x, tiles_y = partition.tile_grid(w, h, tw, th)
        seen = np.zeros((tiles_y, tiles_x), np.int64)
        lengths = set()
        for r in range(world):
            slots = partition.tile_slots(w, h, r, world, tw, th)
            assert np.array_equal(slots, partition.tile_slots_numpy(w, h, r, world, tw, th))
            lengths.add(len(slots))
            for tx, ty in slots:
                if tx >= 0:
                    seen[ty, tx] += 1
                    assert (tx + ty * partition.skew(world)) % world == r
        assert len(lengths) == 1 and np.all(seen == 1)  # equal buffers for all ranks; every tile exactly once
        # pack -> unpack is the identity on any image
        img = np.arange(h * w * 4, dtype=np.float32).reshape(h, w, 4)
        bufs = [partition.pack_tiles(np.where(partition.owned_mask(w, h, r, world, tw, th)[..., None], img, -1.0), r, world, tw, th) for r in range(world)]
        assert np.array_equal(partition.unpack_tiles(bufs, w, h, tw, th), img)


def test_a_ranks_tiles_are_a_lattice_not_column_stripes():
    """1920 / 16 = 120 tiles per row = 0 mod 8: with t % world every rank would own fixed 16-pixel columns.  The skewed
    ownership gives every rank tiles in every tile column (and every tile row) for 2, 4 and 8 ranks."""
    from rsoderh_raytracing_amd import partition
    for world in (2, 4, 8):
        owner = partition.tile_owner_map(1920, 1080, world)[::16, ::16]
        for r in range(world):
            assert (owner == r).any(axis=0).all() and (owner == r).any(axis=1).all()
        counts = np.bincount(owner.ravel(), minlength=world)
        assert counts.max() - counts.min() <= owner.shape[0]  # within one tile per tile row


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
    # balance: every rank owns the same number of tile slots; real tiles per rank differ by at most one per tile row
    tx, ty = partition.tile_grid(1920, 1080)
    owner = partition.tile_owner_map(1920, 1080, 8)[::16, ::16]
    counts = np.bincount(owner.ravel(), minlength=8)
    assert counts.sum() == tx * ty and counts.max() - counts.min() <= ty


def _bench_worker(rank, world, port, full_path, out_path, corrupt):
    """bench.py's N > 1 bookkeeping over gloo, with the frame the ranks would have rendered supplied by the test: the gathered frame is proved
    against the one-process frame, every rank's milliseconds reach rank 0."""
    sys.path.insert(0, util.ROOT)
    import json
    import bench
    from rsoderh_raytracing_amd import partition
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = np.load(full_path)
    h, w = full.shape[:2]
    mine = np.full_like(full, np.nan)
    m = partition.owned_mask(w, h, rank, world)
    mine[m] = full[m]
    if corrupt and rank == world - 1:  # one rank renders one of ITS pixels wrongly: the proof must say so
        ys, xs = np.nonzero(m)
        mine[ys[0], xs[0], 0] += 1.0
    with bench.watchdog("the first exchange (gloo)", rank, "first frame", seconds=60):
        bench.fault("hang_reduce")
        frame = partition.gather_tiles(mine, rank, world)
    stats = bench.gather_rank_stats(rank, world, 10.0 + rank, 0.5 * rank, dist)

    # the pre-flight choice of the exchange: "a" is wrong on the last rank only, "b" cannot even run on rank 0, "c" is right everywhere
    tried, notes = [], []

    def attempt(cand):
        tried.append(cand)
        if cand == "b" and rank == 0:
            raise RuntimeError("ncclGroupEnd failed: unhandled system error")
        return not (cand == "a" and rank == world - 1)

    def agree(ok):
        t = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t[0]) == 1

    chosen = bench.choose_exchange(["a", "b", "c", "d"], attempt, agree, lambda c, m: notes.append((c, m)))
    nothing = bench.choose_exchange(["a"], attempt, agree)
    if rank == 0:
        proof = bench.frame_proof(frame, full)
        with open(out_path, "w") as f:
            json.dump({"proof": proof, "per_rank": stats, "chosen": chosen, "tried": tried, "notes": notes, "nothing": nothing}, f)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("corrupt", [False, True])
def test_bench_multi_gpu_proof_and_rank_stats_over_gloo(tmp_path, corrupt):
    """What `bench.py --gpus N` adds to its JSON line for N > 1 (`frame_equals_1gpu`, `multi_gpu.per_rank`, `multi_gpu.proof`), computed by
    bench.py's own functions in a world of two over gloo: true for the real gather, false — with the pixel named — when a rank's tile is wrong."""
    import json
    rng = np.random.default_rng(4)
    full = rng.uniform(0, 4, (40, 72, 4)).astype(np.float32)
    full[..., 3] = 1.0
    full_path, out_path = str(tmp_path / "full.npy"), str(tmp_path / "out.json")
    np.save(full_path, full)
    port = 33500 + (os.getpid() % 2000) + int(corrupt)
    mp.spawn(_bench_worker, args=(2, port, full_path, out_path, corrupt), nprocs=2, join=True)
    with open(out_path) as f:
        out = json.load(f)
    assert out["proof"]["frame_equals_1gpu"] is (not corrupt)
    if corrupt:
        assert out["proof"]["differing_pixels"] == 1 and len(out["proof"]["first_differing_pixel_xy"]) == 2
    assert [r["rank"] for r in out["per_rank"]] == [0, 1]
    assert [r["trace_ms"] for r in out["per_rank"]] == [10.0, 11.0] and [r["reduce_ms"] for r in out["per_rank"]] == [0.0, 0.5]
    # the exchange that is timed is the first one EVERY rank found right: not "a" (wrong on rank 1 only), not "b" (raised on rank 0), never "d"
    assert out["chosen"] == "c" and out["tried"] == ["a", "b", "c", "a"] and out["nothing"] is None
    assert out["notes"] == [["b", "ncclGroupEnd failed: unhandled system error"]]


@pytest.mark.parametrize("where", ["hang_init", "hang_reduce"])
def test_bench_watchdog_ends_a_hung_collective_with_exit_code_3(where):
    """A collective that never returns (fault injection: RSRT_BENCH_FAULT) must not run into the driver's timeout with nothing written: the
    watchdog thread prints who waited for what and ends the process with os._exit(3)."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); import bench\n"
            "with bench.watchdog('rsrt_comm_init (ncclCommInitRank)' if %r == 'hang_init' else 'the first exchange (rccl)', 1, 'bring-up'):\n"
            "    bench.fault(%r)\n"
            "print('returned')\n" % (util.ROOT, where, where))
    env = dict(os.environ, RSRT_BENCH_FAULT=where, RSRT_BENCH_WATCHDOG_S="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "returned" not in r.stdout
    assert "WATCHDOG: rank 1" in r.stderr and "NCCL_DEBUG" in r.stderr and ("comm_init" in r.stderr if where == "hang_init" else "first exchange" in r.stderr)
    # ... and without the fault the guarded call simply returns
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RSRT_BENCH_WATCHDOG_S="30"), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "returned" in r.stdout
