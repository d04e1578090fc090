"""Multi-GPU behind the C-ABI (include/rsrt.h: rsrt_comm_*, rsrt_multi_*).

CPU part: symbols, argument checking and the partition arithmetic (pure host code of librsrt.so).
GPU part: on the one-GPU box the world is one rank — the RCCL calls (unique id, ncclCommInitRank / ncclCommInitAll,
grouped ncclReduce) all run for real; wherever more GPUs are visible the same tests run with world 2, 4, 8."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import util
import rsoderh_raytracing_amd as R
from rsoderh_raytracing_amd import _build, partition, state


def gpu_count():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def test_partition_arithmetic_of_the_library_matches_the_numpy_restatement():
    L = state.lib()
    for w, h, world, tw, th in [(1920, 1080, 8, 16, 16), (72, 40, 3, 16, 16), (17, 5, 2, 16, 16), (100, 60, 4, 32, 8), (3840, 2160, 8, 16, 16)]:
        owner = partition.tile_owner_map(w, h, world, tw, th)
        total = np.zeros((h, w), np.int64)
        for r in range(world):
            m = partition.owned_mask(w, h, r, world, tw, th)
            assert np.array_equal(m, owner == r)
            total += m
        assert np.all(total == 1)  # every pixel has exactly one renderer: the reduce adds x + 0 + ... + 0
        for x, y in [(0, 0), (w - 1, h - 1), (w // 2, h // 3)]:
            assert L.rsrt_partition_owner(w, h, tw, th, world, x, y) == owner[y, x]
    assert L.rsrt_partition_owner(64, 64, 16, 16, 2, 64, 0) == 0xFFFFFFFF  # outside the frame
    assert L.rsrt_partition_mask(64, 64, 16, 16, 2, 2, None, None) == 1      # rank >= world
    assert L.rsrt_partition_mask(64, 64, 10, 10, 0, 2, None, None) == 1      # tile of 100 pixels: not whole waves


def test_comm_and_multi_entry_points_check_their_arguments_without_a_gpu():
    L = state.lib()
    ident = C.create_string_buffer(128)
    assert L.rsrt_comm_init(None, 0, 1, ident) == 1
    assert L.rsrt_comm_reduce(None, 0, None, None) == 1
    assert L.rsrt_comm_set_mode(None, 0) == 1
    assert L.rsrt_comm_destroy(None) == 1
    assert L.rsrt_comm_unique_id(None) == 1
    h = C.c_void_p()
    assert L.rsrt_multi_create(None, 0, C.byref(h)) == 1 and b"device list" in L.rsrt_multi_last_error(None)
    devs = (C.c_int * 2)(3, 3)
    assert L.rsrt_multi_create(devs, 2, C.byref(h)) == 1 and b"twice" in L.rsrt_multi_last_error(None)
    assert L.rsrt_multi_size(None) == 0 and L.rsrt_multi_context(None, 0) is None
    assert L.rsrt_multi_render(None, None, 1, 1, 0, 1, 1, 0, 0) == 1
    if gpu_count() == 0:
        devs = (C.c_int * 2)(0, 1)
        assert L.rsrt_multi_create(devs, 2, C.byref(h)) == 2 and b"no HIP device" in L.rsrt_multi_last_error(None)


def build_ranks_demo(tmp_path):
    exe = str(tmp_path / "ranks_demo")
    pkg = os.path.join(util.ROOT, "rsoderh-raytracing_amd")
    _build.build_host()
    _build.build_hip()
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-pthread", "-I", os.path.join(util.ROOT, "include"),
           os.path.join(util.ROOT, "tests", "cpp", "ranks_demo.cpp"), "-o", exe, "-L", pkg, "-lrsrt", "-lrsrt_host",
           "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return exe


def test_cpp_rank_program_compiles_against_the_headers(tmp_path):
    build_ranks_demo(tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_one_process_per_gpu_through_the_c_abi(world, tmp_path):
    """N copies of tests/cpp/ranks_demo.cpp (C-ABI only, no Python, no torch in the data path): unique id by file,
    rsrt_comm_init, tiles rendered per rank, ONE rsrt_comm_reduce — the frame equals the oracle's bit for bit."""
    import oracle
    if gpu_count() < world:
        pytest.skip("needs %d GPUs" % world)
    exe = build_ranks_demo(tmp_path)
    w, h, spp, mb = 200, 120, 4, 8
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([exe, util.scene_path("house"), str(w), str(h), str(spp), str(mb), "256", "128", str(tmp_path / "o.f32"),
                               str(world), str(r), str(tmp_path / "id.bin")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
             for r in range(world)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    sums = np.fromfile(tmp_path / "o.f32", np.float32).reshape(h, w, 4)
    sc = R.Scene.load_toml(util.scene_path("house"))
    envm = R.Environment.synthetic(256, 128)
    ref, st = oracle.render(util.oracle_scene(sc), util.oracle_env(envm), sc.camera_uniform().view(oracle.CAMERA), w, h, 0, spp, mb)
    assert np.array_equal(util.bits(sums), util.bits(ref))
    assert sum(int(o.split("paths ")[1].split()[0]) for o in outs) == st["paths"]


@pytest.mark.gpu
def test_comm_reduce_in_place_and_into_a_frame_buffer():
    """World of one rank with a REAL communicator: in place the accumulator is unchanged, with a receive buffer the
    frame lands there and the accumulator stays what it was (progressive callers keep accumulating into it)."""
    import torch
    sc = R.Scene.load_toml(util.scene_path("default"))
    env = util.small_env()
    st = R.State.new(sc, env, 80, 48)
    st.comm_init(0, 1, R.State.comm_unique_id())
    st.render_samples(3)
    before = st.download()
    st.comm_reduce(0)
    assert np.array_equal(util.bits(st.download()), util.bits(before))
    frame = torch.full((48, 80, 4), -1.0, dtype=torch.float32, device="cuda")
    st.comm_reduce(0, recv_ptr=frame.data_ptr())
    st.synchronize()
    assert np.array_equal(util.bits(frame.cpu().numpy()), util.bits(before))
    s = st.stats()
    assert s["reduce_ms"] > 0
    st.comm_destroy()
    st.comm_reduce(0)  # no communicator: a world of one again, nothing to do
    with pytest.raises(R.RsrtError, match="root"):
        st.comm_reduce(1)
    st.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 4, 8])
def test_device_list_state_equals_the_oracle(n):
    """rsrt_multi_*: one caller, a list of devices; progressive (two render calls), frame and display via the reduce."""
    import oracle
    from rsoderh_raytracing_amd import host
    if gpu_count() < n:
        pytest.skip("needs %d GPUs" % n)
    sc = R.Scene.load_toml(util.scene_path("house"))
    env = util.small_env()
    w, h = 150, 70
    ms = state.MultiState(sc, env, w, h, devices=list(range(n)))
    assert ms.size() == n
    ms.max_bounces = 8
    ms.render_samples(2)
    first = ms.download()          # reduce #1 must not disturb the per-device accumulators ...
    ms.render_samples(3)           # ... that keep accumulating
    img, shown, stats = ms.download(), ms.display_srgb8(), ms.stats()
    ms.close()
    osc, oenv, cam = util.oracle_scene(sc), util.oracle_env(env), sc.camera_uniform().view(oracle.CAMERA)
    ref2, _ = oracle.render(osc, oenv, cam, w, h, 0, 2, 8)
    ref5, ost = oracle.render(osc, oenv, cam, w, h, 0, 5, 8)
    assert np.array_equal(util.bits(first), util.bits(ref2))
    assert np.array_equal(util.bits(img), util.bits(ref5))
    assert np.array_equal(shown, host.display_srgb8(ref5, 5))
    assert (stats["paths"], stats["ext_rays"], stats["shadow_rays"]) == (ost["paths"], ost["ext_rays"], ost["shadow_rays"])


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 3, 8])
def test_device_list_rehearsed_on_one_gpu(n, monkeypatch):
    """N "devices" that are all GPU 0 (RSRT_MULTI_ALLOW_SAME_DEVICE=1): N contexts, N interleaved tile partitions, N
    kernels, and the reduce that rsrt_multi falls back to when RCCL cannot be used (peer copies + adds on devices[0];
    RCCL refuses two ranks on one device).  Everything of the N > 1 path except RCCL itself, on the hardware at hand."""
    import oracle
    from rsoderh_raytracing_amd import host
    monkeypatch.setenv("RSRT_MULTI_ALLOW_SAME_DEVICE", "1")
    sc = R.Scene.load_toml(util.scene_path("house"))
    env = util.small_env()
    w, h = 150, 70
    ms = state.MultiState(sc, env, w, h, devices=[0] * n)
    assert ms.size() == n and not ms.uses_rccl()
    ms.max_bounces = 8
    ms.render_samples(2)
    first = ms.download()
    ms.render_samples(3)
    img, shown, stats = ms.download(), ms.display_srgb8(), ms.stats()
    again = ms.download()  # a second reduce of the same accumulators gives the same frame
    ms.close()
    osc, oenv, cam = util.oracle_scene(sc), util.oracle_env(env), sc.camera_uniform().view(oracle.CAMERA)
    ref2, _ = oracle.render(osc, oenv, cam, w, h, 0, 2, 8)
    ref5, ost = oracle.render(osc, oenv, cam, w, h, 0, 5, 8)
    assert np.array_equal(util.bits(first), util.bits(ref2))
    assert np.array_equal(util.bits(img), util.bits(ref5)) and np.array_equal(util.bits(again), util.bits(ref5))
    assert np.array_equal(shown, host.display_srgb8(ref5, 5))
    assert (stats["paths"], stats["ext_rays"], stats["shadow_rays"]) == (ost["paths"], ost["ext_rays"], ost["shadow_rays"])
