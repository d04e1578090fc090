"""Parity tests proper: the HIP path (through the C-ABI) against the CPU oracle and the committed
golden vectors.  Integer/index work and — because both sides evaluate the same correctly-rounded
f32 expression trees (DESIGN.md "Numerics") — the radiance sums too are compared BIT-EXACT; the
north-star tolerance (1e-3 per-channel RMSE) is asserted as well where stated."""
import ctypes
import os

import numpy as np
import pytest

import oracle
import util
import rsoderh_raytracing_amd as R
from rsoderh_raytracing_amd import partition

pytestmark = pytest.mark.gpu

SCENES = ["house", "default", "cube", "suzanne", "spheres_only"]
RMSE_TOL = 1e-3  # BASELINE.json north_star


def golden(name):
    return np.load(os.path.join(util.ROOT, "tests", "golden", "scene_%s.npz" % name))


def golden_env():
    g = np.load(os.path.join(util.ROOT, "tests", "golden", "env_64x32.npz"))
    return R.Environment(g["rgba"], g["alias"].view(R.types.ALIAS_ENTRY).reshape(-1))


@pytest.fixture(scope="module")
def big_env():
    return R.Environment.synthetic(2048, 1024)


def gpu_render(scene, env, w, h, begin, count, mb, flags=0, partition_args=None):
    st = R.State.new(scene, env, w, h)
    st.max_bounces, st.flags = mb, flags
    if partition_args:
        st.set_partition(*partition_args)
    st.render_range(begin, count)
    img, stats = st.download(), st.stats()
    st.close()
    return img, stats


@pytest.mark.parametrize("name", SCENES)
def test_matches_golden_images_bit_exact(name):
    g = golden(name)
    sc, env = R.Scene.load_toml(util.scene_path(name)), golden_env()
    for key in [k for k in g.files if k.startswith("sum_")]:
        spp, mb = int(key.split("_")[1][:-3]), int(key.split("_")[2][:-1])
        for flags in (0, R.state.FLAG_REFERENCE_TRAVERSAL):
            img, st = gpu_render(sc, env, 64, 64, 0, spp, mb, flags)
            assert np.array_equal(util.bits(img), util.bits(g[key])), (key, flags)
            assert [st["paths"], st["ext_rays"], st["shadow_rays"]] == list(g["rays_%dspp_%db" % (spp, mb)])


def probe_modes(name):
    """Every way rsrt_cast_rays can run a query (include/rsrt.h): traversal 0 threaded / 1 stack / 2 typed leaf loops /
    3 flat (what house, default and cube run in production; suzanne's 968 triangles do not qualify) / 4 fixed-order walk
    / 5 wide walk (4-wide nodes, one ray a lane) / 6 cooperative wide walk (the same nodes, a wave's rays as work items on two LDS stacks: what
    suzanne and anything bigger run), x scene read from global memory or from LDS as the production kernel stages
    it for that traversal (bit 4), x cast_ray / cast_ray_bvh (bit 0)."""
    sels = [0, 1, 2, 4, 5, 6] + ([3] if name != "suzanne" else [])  # (5, 6: the wide walks — every builder-made tree qualifies)
    return [(sel << 1) | lds | bvh_only for sel in sels for lds in (0, 16) for bvh_only in (0, 1)]


@pytest.mark.parametrize("name", SCENES)
def test_ray_batch_matches_golden_bit_exact(name):
    """The committed 1 k-ray hit records through all four traversals, the production ones included."""
    g = golden(name)
    st = R.State.new(R.Scene.load_toml(util.scene_path(name)), golden_env(), 16, 16)
    for mode in probe_modes(name):
        key = "hits_bvh" if mode & 1 else "hits"
        for flags in (0, R.state.FLAG_REFERENCE_TRAVERSAL):
            h = st.cast_rays(g["ray_o"], g["ray_d"], mode, flags)
            assert np.array_equal(np.ascontiguousarray(h).view(np.uint32).reshape(-1, 9), g[key]), (key, mode, flags)
    st.close()


def test_probe_refuses_a_traversal_the_scene_does_not_qualify_for():
    st = R.State.new(R.Scene.load_toml(util.scene_path("suzanne")), golden_env(), 16, 16)
    o, d = np.zeros((4, 3), np.float32), np.tile(np.float32([0, 0, -1]), (4, 1))
    with pytest.raises(R.RsrtError, match="flat traversal"):
        st.cast_rays(o, d, 3 << 1, 0)
    with pytest.raises(R.RsrtError, match="bad arguments"):
        st.cast_rays(o, d, 32, 0)
    with pytest.raises(R.RsrtError, match="bad arguments"):
        st.cast_rays(o, d, 7 << 1, 0)
    st.close()
    # a tree whose boxes do not nest (a leaf box pushed out of its parent's) keeps the fixed-order walk: the wide walk is refused
    sc = R.Scene.load_toml(util.scene_path("default"))
    nodes = sc.bvh_nodes.copy()
    leaf = next(i for i in range(len(nodes)) if nodes[i]["primitives_len"] > 0)
    nodes["bounds_max"][leaf, 0] += 100.0
    bad = R.Scene(sc.materials, sc.spheres, sc.plane_descs, sc.vertices, sc.normals, sc.triangles, sc.camera_desc, planes=sc.planes,
                  primitives=sc.primitives, bvh_nodes=nodes, bvh_depth=sc.bvh_depth)
    st = R.State.new(bad, golden_env(), 16, 16)
    with pytest.raises(R.RsrtError, match="wide walk"):
        st.cast_rays(o, d, 5 << 1, 0)
    with pytest.raises(R.RsrtError, match="cooperative walk"):
        st.cast_rays(o, d, 6 << 1, 0)
    st.cast_rays(o, d, 4 << 1, 0)
    st.close()


@pytest.mark.parametrize("name,w,h,spp,mb", [("house", 160, 90, 8, 8), ("default", 128, 72, 8, 10), ("suzanne", 96, 64, 4, 10),
                                             ("house", 67, 35, 3, 10), ("cube", 128, 72, 8, 10)])
def test_matches_oracle_live(name, w, h, spp, mb, big_env):
    """Seeded live comparison on the full-size 2048x1024 environment (ragged sizes included)."""
    sc = R.Scene.load_toml(util.scene_path(name))
    ref, ost = oracle.render(util.oracle_scene(sc), util.oracle_env(big_env), sc.camera_uniform().view(oracle.CAMERA), w, h, 0, spp, mb)
    img, st = gpu_render(sc, big_env, w, h, 0, spp, mb)
    assert np.all(util.rmse_per_channel(img, ref, spp) <= RMSE_TOL)
    assert np.array_equal(util.bits(img), util.bits(ref))
    assert (st["paths"], st["ext_rays"], st["shadow_rays"]) == (ost["paths"], ost["ext_rays"], ost["shadow_rays"])


@pytest.mark.parametrize("variant,traversal", [("4", "6"), ("2", "6"), ("4", "6-noflat"), ("1", "6-noflat"), ("0", "4"), ("1", "4"), ("2", "4"), ("3", "4"), ("4", "4"), ("2", "4-noflat"), ("4", "4-noflat"), ("2", "3"), ("4", "3"), ("2", "3-noflat"), ("4", "3-noflat"), ("2", "1"), ("4", "1"), ("2", "0")])
def test_every_kernel_variant_is_bit_exact(variant, traversal, big_env, monkeypatch):
    """RSRT_KERNEL: 0 = lockstep megakernel, 1/2/3 = stage-scheduled wave-pool kernel (192/160/128 slots per wave), 4 (the
    default) = one 1024-thread workgroup and one scene copy per CU for scenes that fit LDS, 192 slots per wave;
    RSRT_TRAVERSAL caps the traversal: 6 = product (flat loop for small scenes, cooperative wide walk otherwise; with RSRT_FLAT=0 that
    walk for small scenes too), 4 = the one-ray-a-lane wide walk instead, 3 = the fixed-order walk instead of the wide one, 1 tree walk with per-type leaf loops, 0 generic tree walk.
    Scheduling differs, the per-path arithmetic does not: all must give the oracle's bits."""
    monkeypatch.setenv("RSRT_KERNEL", variant)
    monkeypatch.setenv("RSRT_TRAVERSAL", traversal[0])
    if traversal.endswith("noflat"):
        monkeypatch.setenv("RSRT_FLAT", "0")
    for name, w, h, spp, mb in [("house", 150, 70, 6, 8), ("suzanne", 64, 48, 3, 10)]:
        sc = R.Scene.load_toml(util.scene_path(name))
        ref, ost = oracle.render(util.oracle_scene(sc), util.oracle_env(big_env), sc.camera_uniform().view(oracle.CAMERA), w, h, 0, spp, mb)
        for flags in (0, R.state.FLAG_REFERENCE_TRAVERSAL):
            img, st = gpu_render(sc, big_env, w, h, 0, spp, mb, flags)
            assert np.array_equal(util.bits(img), util.bits(ref)), (name, variant, traversal, flags)
            assert (st["ext_rays"], st["shadow_rays"]) == (ost["ext_rays"], ost["shadow_rays"])


def _axis_parallel_rays(scene, rng, n=4096):
    """Rays with exactly zero direction components (inv_dir = +-inf) starting ON node / primitive bounds,
    so that (bound - origin) * inv_dir hits 0 * inf = NaN in the slab test — the case the branch-free
    slab test must treat exactly like the shader's compare chain."""
    nodes = scene.bvh_nodes
    lo, hi = nodes["bounds_min"], nodes["bounds_max"]
    o = np.zeros((n, 3), np.float32)
    d = np.zeros((n, 3), np.float32)
    for i in range(n):
        k = rng.integers(0, len(nodes))
        corner = np.where(rng.integers(0, 2, 3) == 1, hi[k], lo[k]).astype(np.float32)
        mode = i % 4
        p = corner.copy()
        if mode >= 1:  # move off the corner along some axes, staying on at least one face plane
            ax = rng.integers(0, 3)
            p[ax] += np.float32(rng.normal())
        if mode == 3:
            p[(ax + 1) % 3] += np.float32(rng.normal() * 0.5)
        o[i] = p
        axis = rng.integers(0, 3)
        if i % 3 == 0:      # one non-zero component
            d[i, axis] = rng.choice([-1.0, 1.0])
        elif i % 3 == 1:    # two non-zero components
            v = rng.normal(size=2)
            v /= np.linalg.norm(v)
            d[i, (axis + 1) % 3], d[i, (axis + 2) % 3] = v
        else:               # negative zero in one component
            v = rng.normal(size=3)
            v[axis] = -0.0
            d[i] = v / np.linalg.norm(v)
    return o, d


@pytest.mark.parametrize("name", ["house", "default", "suzanne"])
def test_axis_parallel_rays_on_box_faces_bit_exact(name):
    sc = R.Scene.load_toml(util.scene_path(name))
    osc = util.oracle_scene(sc)
    o, d = _axis_parallel_rays(sc, np.random.default_rng(11))
    st = R.State.new(sc, golden_env(), 16, 16)
    for mode in probe_modes(name):  # all four traversals (flat: these rays take its tree-walk fallback), LDS and global
        ref = oracle.cast_rays(osc, o, d, mode & 1, 0)
        got = st.cast_rays(o, d, mode, 0)
        a = np.ascontiguousarray(got).view(np.uint32).reshape(-1, 9)
        b = ref.view(np.uint32).reshape(-1, 9)
        assert np.array_equal(a, b), (name, mode, int((a != b).any(axis=1).sum()))
    st.close()


def test_random_scenes_bit_exact(big_env):
    """Random spheres / planes / triangles (degenerate and coincident ones included), random cameras."""
    from rsoderh_raytracing_amd import host, types as T
    rng = np.random.default_rng(5)
    for trial in range(6):
        ns, npl, nt = int(rng.integers(0, 8)), int(rng.integers(0, 3)), int(rng.integers(1, 40))
        mats = np.zeros(4, T.MATERIAL)
        mats["color"] = rng.uniform(0.1, 1, (4, 3))
        mats["roughness"] = [1.0, 0.3, 0.0, 0.6]
        mats["metallic"] = [0.0, 1.0, 1.0, 0.5]
        mats["emission"] = [[0, 0, 0], [0, 0, 0], [0, 0, 0], [0.5, 0.2, 0.1]]
        sph = np.zeros(ns, T.SPHERE)
        sph["pos"], sph["radius"], sph["material_id"] = rng.uniform(-3, 3, (ns, 3)), rng.uniform(0.2, 1.0, ns), rng.integers(0, 4, ns)
        pls = np.zeros(npl, T.PLANE_DESC)
        pls["pos"], pls["forward"], pls["right"] = rng.uniform(-4, 0, (npl, 3)), rng.uniform(-6, 6, (npl, 3)), rng.uniform(-6, 6, (npl, 3))
        pls["material_id"] = rng.integers(0, 4, npl)
        verts = np.zeros(3 * nt, T.VEC3)
        verts["v"] = np.round(rng.uniform(-3, 3, (3 * nt, 3)) * 4) / 4  # coarse grid: shared edges, coplanar faces
        norms = np.zeros(3 * nt, T.VEC3)
        nn = rng.normal(size=(3 * nt, 3))
        norms["v"] = nn / np.linalg.norm(nn, axis=1, keepdims=True)
        tri = np.zeros(nt, T.TRIANGLE)
        tri["vertex_0"], tri["vertex_1"], tri["vertex_2"] = np.arange(nt) * 3, np.arange(nt) * 3 + 1, np.arange(nt) * 3 + 2
        tri["normal_0"], tri["normal_1"], tri["normal_2"] = tri["vertex_0"], tri["vertex_1"], tri["vertex_2"]
        tri["material_id"] = rng.integers(0, 4, nt)
        if trial == 0:  # a degenerate (zero-area) triangle and a duplicate one
            verts["v"][0:3] = verts["v"][0]
            verts["v"][3:6] = verts["v"][6:9]
        cam = host.make_camera_desc(rng.uniform(-1, 1, 3) + [0, 1, 5], yaw=rng.uniform(-0.3, 0.3), pitch=rng.uniform(-0.3, 0.1), fov_y=1.2)
        sc = R.Scene(mats, sph, pls, verts, norms, tri, cam)
        ref, ost = oracle.render(util.oracle_scene(sc), util.oracle_env(big_env), sc.camera_uniform().view(oracle.CAMERA), 80, 48, 0, 4, 6)
        img, st = gpu_render(sc, big_env, 80, 48, 0, 4, 6)
        same = util.bits(img) == util.bits(ref)
        assert same.all(), (trial, int((~same).any(axis=2).sum()))
        assert (st["ext_rays"], st["shadow_rays"]) == (ost["ext_rays"], ost["shadow_rays"])


def test_opt_in_pruning_stays_within_tolerance(big_env, monkeypatch):
    """RSRT_FLAG_PRUNE on a scene whose BVH is really walked (suzanne): fewer box / primitive tests, the same rays, an
    image within the north-star tolerance of the exact one (it is NOT exactly result-preserving: include/rsrt.h).
    On the scenes that run the flat loop (house, default, cube) the flag has nothing to act on."""
    sc = R.Scene.load_toml(util.scene_path("suzanne"))
    a, sa = gpu_render(sc, big_env, 160, 90, 0, 8, 10)
    b, sb = gpu_render(sc, big_env, 160, 90, 0, 8, 10, R.state.FLAG_PRUNE)
    assert np.all(util.rmse_per_channel(a, b, 8) <= RMSE_TOL)
    assert (sa["ext_rays"], sa["shadow_rays"]) == (sb["ext_rays"], sb["shadow_rays"])
    # the flag selects the near-child-first tree walk: its steps are counted against the SAME walk without pruning (the wide
    # walk the unflagged render takes counts node visits, four boxes each — another unit)
    monkeypatch.setenv("RSRT_TRAVERSAL", "1")
    a1, sa1 = gpu_render(sc, big_env, 160, 90, 0, 8, 10)
    assert np.array_equal(util.bits(a1), util.bits(a))
    assert 0 < sb["traversal_steps"] < 0.95 * sa1["traversal_steps"], (sa1["traversal_steps"], sb["traversal_steps"])  # measured: 0.917
    monkeypatch.delenv("RSRT_TRAVERSAL")
    house = R.Scene.load_toml(util.scene_path("house"))
    c, sc_ = gpu_render(house, big_env, 96, 54, 0, 4, 8)
    d, sd = gpu_render(house, big_env, 96, 54, 0, 4, 8, R.state.FLAG_PRUNE)
    assert np.array_equal(util.bits(c), util.bits(d)) and sc_["traversal_steps"] == sd["traversal_steps"] == 0


def test_pruning_counterexample_default_flags_are_exact(big_env):
    """house.toml 1920x1080, sample 149 of pixel (851,477): the path t-pruning gets wrong.  Default
    flags must give the literal traversal's value there."""
    sc = R.Scene.load_toml(util.scene_path("house"))
    lit, _ = gpu_render(sc, big_env, 1920, 1080, 149, 1, 8, R.state.FLAG_REFERENCE_TRAVERSAL)
    dflt, _ = gpu_render(sc, big_env, 1920, 1080, 149, 1, 8, 0)
    assert np.array_equal(util.bits(lit), util.bits(dflt))
    assert list(dflt[477, 851, :3]) == [0.0, 0.0, 0.0]


def test_sample_ranges_compose_exactly(big_env):
    """[0,5) then [5,12) into the same accumulator == [0,12) in one call, and == 12 reference frames."""
    sc = R.Scene.load_toml(util.scene_path("house"))
    one, _ = gpu_render(sc, big_env, 96, 54, 0, 12, 8)
    st = R.State.new(sc, big_env, 96, 54)
    st.max_bounces = 8
    st.render_range(0, 5)
    st.render_range(5, 7)
    two = st.download()
    st.clear()
    for _ in range(12):
        st._last_hash = st._scene_hash()
        st.render()  # State::render: one sample per frame
    prog = st.download()
    st.close()
    assert np.array_equal(util.bits(one), util.bits(two))
    assert np.array_equal(util.bits(one), util.bits(prog))


def test_renders_on_different_caller_streams_are_one_chain(big_env):
    """rsrt_accumulator_clear runs on the context's own stream; renders may be put on any caller stream.  The work
    buffers and the accumulator belong to the context, so the library orders every enqueue after the previous one,
    whatever the stream: clear -> render on s1 -> render on s2 -> render on the context's stream == one render."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")  # two caller streams straight from the HIP runtime the library itself uses
    s1, s2 = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipStreamCreate(ctypes.byref(s1)) == 0 and hip.hipStreamCreate(ctypes.byref(s2)) == 0
    sc = R.Scene.load_toml(util.scene_path("house"))
    w, h = 320, 180  # big enough that a render is still running when the next one is enqueued
    ref, _ = oracle.render(util.oracle_scene(sc), util.oracle_env(big_env), sc.camera_uniform().view(oracle.CAMERA), w, h, 0, 7, 8)
    st = R.State.new(sc, big_env, w, h)
    st.max_bounces = 8
    for _ in range(3):  # the second and third round clear an accumulator that is full of samples
        st._last_hash = None
        st.render_samples(3, stream=s1.value)   # hash changed: clear (context stream), then samples 0..2 on s1
        st.render_samples(2, stream=s2.value)   # samples 3..4 on another stream, same sample buffer / path arena
        st.render_samples(2)                          # samples 5..6 on the context's own stream
        img = st.download()
        assert np.array_equal(util.bits(img), util.bits(ref))
    st.close()
    assert hip.hipStreamDestroy(s1) == 0 and hip.hipStreamDestroy(s2) == 0


def test_sample_buffer_passes_do_not_change_the_image(big_env, monkeypatch):
    sc = R.Scene.load_toml(util.scene_path("default"))
    a, _ = gpu_render(sc, big_env, 320, 192, 0, 5, 6)
    monkeypatch.setenv("RSRT_SAMPLE_BUFFER_MB", "1")  # 0.74 MB per sample -> one pass per sample
    b, st = gpu_render(sc, big_env, 320, 192, 0, 5, 6)
    assert st["launches"] > 2
    assert np.array_equal(util.bits(a), util.bits(b))


@pytest.mark.parametrize("world", [2, 8])
def test_tile_partition_union_is_the_full_image(world, big_env):
    """Each 'rank' renders only its tiles; the sum over ranks (what the RCCL reduce computes) is
    bit-identical to the single-GPU image, and no rank touches a pixel it does not own."""
    sc = R.Scene.load_toml(util.scene_path("house"))
    w, h = 200, 120
    full, fst = gpu_render(sc, big_env, w, h, 0, 4, 8)
    total = np.zeros_like(full)
    rays = 0
    for r in range(world):
        part, st = gpu_render(sc, big_env, w, h, 0, 4, 8, partition_args=(r, world, 16, 16))
        m = partition.owned_mask(w, h, r, world)
        assert np.all(part[~m] == 0)
        total += part
        rays += st["ext_rays"] + st["shadow_rays"]
    assert np.array_equal(util.bits(total), util.bits(full))
    assert rays == fst["ext_rays"] + fst["shadow_rays"]


def test_full_size_properties(big_env):
    """BASELINE config 4 geometry (1920x1080, 8 bounces) at 2 spp: linear in samples, alpha 1,
    finite, and the oracle agrees on a sampled window of rows."""
    sc = R.Scene.load_toml(util.scene_path("house"))
    st = R.State.new(sc, big_env, 1920, 1080)
    st.max_bounces = 8
    st.render_range(0, 2)
    a = st.download()
    s = st.stats()
    st.close()
    assert np.isfinite(a).all() and np.all(a[..., 3] == 1.0) and a[..., :3].min() >= 0
    assert s["paths"] == 1920 * 1080 * 2 and s["ext_rays"] >= s["paths"]
    ref, _ = oracle.render(util.oracle_scene(sc), util.oracle_env(big_env), sc.camera_uniform().view(oracle.CAMERA), 1920, 1080, 0, 2, 8)
    assert np.array_equal(util.bits(a), util.bits(ref))


def test_the_complete_baseline_frame_is_bit_exact(big_env):
    """BASELINE config 4 in full — house.toml 1920x1080, 256 spp, 8 bounces, what bench.py times: every one of the
    2,073,600 RGBA32F sums equal to the oracle's, bit for bit, and the same 2,043,817,015 rays.  (The oracle needs
    ~45 s on the GPU box's 16 cores; -O3 twin of the strict build, bit-identical to it: test_oracle_kat.)"""
    sc = R.Scene.load_toml(util.scene_path("house"))
    img, st = gpu_render(sc, big_env, 1920, 1080, 0, 256, 8)
    ref, ost = oracle.render(util.oracle_scene(sc), util.oracle_env(big_env), sc.camera_uniform().view(oracle.CAMERA), 1920, 1080, 0, 256, 8, fast=True)
    assert np.array_equal(util.bits(img), util.bits(ref))
    assert (st["paths"], st["ext_rays"], st["shadow_rays"]) == (ost["paths"], ost["ext_rays"], ost["shadow_rays"])
    assert st["ext_rays"] + st["shadow_rays"] == 2043817015
    assert np.all(util.rmse_per_channel(img, ref, 256) <= RMSE_TOL)  # the north-star tolerance, trivially


def test_stats_window_survives_many_small_calls(big_env):
    """40 one-sample frames (more than the library's event pool) must add up to one 40-sample call."""
    sc = R.Scene.load_toml(util.scene_path("default"))
    st = R.State.new(sc, big_env, 96, 64)
    st.render_range(0, 40)
    one = st.stats()
    st.clear()
    for k in range(40):
        st.render_range(k, 1)
    many = st.stats()
    img_many = st.download()
    st.clear()
    st.render_range(0, 40)
    img_one = st.download()
    st.close()
    for key in ("paths", "ext_rays", "shadow_rays"):
        assert one[key] == many[key], key
    assert many["launches"] == 80 and one["launches"] == 2
    assert many["kernel_ms"] > 0 and abs(many["kernel_ms"] - (many["trace_kernel_ms"] + many["resolve_kernel_ms"])) < 1e-6
    assert np.array_equal(util.bits(img_one), util.bits(img_many))


def test_config5_geometry_partitioned(big_env):
    """BASELINE config 5 geometry: 3840x2160, framebuffer tiled over 8 ranks.  Two of the eight ranks at
    1 spp: every owned pixel equals the oracle's, every other pixel is untouched."""
    sc = R.Scene.load_toml(util.scene_path("house"))
    w, h = 3840, 2160
    ref, _ = oracle.render(util.oracle_scene(sc), util.oracle_env(big_env), sc.camera_uniform().view(oracle.CAMERA), w, h, 0, 1, 8, fast=True)
    for rank in (0, 5):
        part, st = gpu_render(sc, big_env, w, h, 0, 1, 8, partition_args=(rank, 8, 16, 16))
        m = partition.owned_mask(w, h, rank, 8)
        assert np.array_equal(util.bits(part[m]), util.bits(ref[m]))
        assert np.all(part[~m] == 0)
        assert st["paths"] == int(m.sum())


def test_mean_f16_and_alpha(big_env):
    sc = R.Scene.load_toml(util.scene_path("default"))
    st = R.State.new(sc, big_env, 64, 40)
    st.render_samples(3)
    s, m = st.download(), st.download_mean_f16()
    st.close()
    assert np.array_equal(m[..., :3], (s[..., :3] / np.float32(3)).astype(np.float16))
    assert np.all(m[..., 3] == 1.0)


def test_scene_hash_reset_and_environment_switch(big_env):
    sc = R.Scene.load_toml(util.scene_path("default"))
    st = R.State.new(sc, [big_env, golden_env()], 48, 32)
    st.render_samples(2)
    assert st.sample_count == 2
    st.update(environment_index=1)
    st.render()  # scene hash changed -> accumulator cleared, sample index restarts at 0
    assert st.sample_count == 1
    a = st.download()
    st.close()
    ref, _ = oracle.render(util.oracle_scene(sc), util.oracle_env(golden_env()), sc.camera_uniform().view(oracle.CAMERA), 48, 32, 0, 1, 10)
    assert np.array_equal(util.bits(a), util.bits(ref))


def test_zero_bounces_and_zero_samples(big_env):
    sc = R.Scene.load_toml(util.scene_path("default"))
    st = R.State.new(sc, big_env, 32, 16)
    st.max_bounces = 0
    st.render_range(0, 3)
    a = st.download()
    assert np.all(a[..., :3] == 0) and np.all(a[..., 3] == 1)
    st.render_range(3, 0)
    st.close()


def test_invalid_input_is_rejected_not_launched(big_env):
    from rsoderh_raytracing_amd.state import RsrtError
    sc = R.Scene.load_toml(util.scene_path("default"))
    st = R.State(0)
    with pytest.raises(RsrtError, match="no scene"):
        st.camera = sc.camera_uniform()
        st.width, st.height = 8, 8
        st.render_range(0, 1)
    bad = R.Scene.load_toml(util.scene_path("default"))
    bad.triangles = bad.triangles.copy()
    bad.triangles["vertex_0"][0] = 10 ** 6
    with pytest.raises(RsrtError, match="vertex index out of range"):
        st.upload_scene(bad)
    bad = R.Scene.load_toml(util.scene_path("default"))
    bad.bvh_nodes = bad.bvh_nodes.copy()
    bad.bvh_nodes["primitives_or_second_child_index"][0] = 0  # cycle
    with pytest.raises(RsrtError, match="bvh"):
        st.upload_scene(bad)
    st.upload_scene(sc)
    with pytest.raises(RsrtError, match="environment 0 not uploaded"):
        st.render_range(0, 1)
    st.close()


def test_short_reciprocal_is_the_ieee_quotient_for_every_input(big_env):
    """rt_rcp (rt_math.h): v_rcp_f32 + one Newton step replaces the 11-instruction division where the exponent
    allows it.  The device compares it with `1.0f / x` for all 2^32 bit patterns."""
    sc = R.Scene.load_toml(util.scene_path("cube"))
    st = R.State.new(sc, big_env, 16, 16)
    r = st.selftest_numerics()
    st.close()
    assert r["mismatches"] == 0, hex(r["first_bad"])
    assert r["short_path_inputs"] == 2 * 251 * 2 ** 23
    assert r["bare_rcp_wrong"] > 0  # the comparison is live: the bare instruction is NOT correctly rounded


def test_foreign_bvh_with_long_leaves_uses_the_generic_loop(big_env):
    """A host may upload any valid BVH.  Two leaves of ~11 primitives each (more than the 8 the per-type leaf
    masks hold) must take the generic in-order leaf loop and still give the oracle's bits."""
    base = R.Scene.load_toml(util.scene_path("default"))
    nodes, prims = base.bvh_nodes, base.primitives
    leaves = [n for n in nodes if n["primitives_len"] > 0]
    leaves.sort(key=lambda n: int(n["primitives_or_second_child_index"]))
    half = len(prims) // 2
    cut = min((int(n["primitives_or_second_child_index"]) for n in leaves), key=lambda i: abs(i - half))
    assert 8 < cut < len(prims) - 8

    def union(sel):
        return (np.min([n["bounds_min"] for n in sel], axis=0), np.max([n["bounds_max"] for n in sel], axis=0))

    new = np.zeros(3, nodes.dtype)
    new[0]["bounds_min"], new[0]["bounds_max"] = union(leaves)
    new[0]["primitives_or_second_child_index"], new[0]["primitives_len"], new[0]["split_axis"] = 2, 0, 0
    a = [n for n in leaves if int(n["primitives_or_second_child_index"]) < cut]
    b = [n for n in leaves if int(n["primitives_or_second_child_index"]) >= cut]
    new[1]["bounds_min"], new[1]["bounds_max"] = union(a)
    new[1]["primitives_or_second_child_index"], new[1]["primitives_len"] = 0, cut
    new[2]["bounds_min"], new[2]["bounds_max"] = union(b)
    new[2]["primitives_or_second_child_index"], new[2]["primitives_len"] = cut, len(prims) - cut
    sc = R.Scene(base.materials, base.spheres, base.plane_descs, base.vertices, base.normals, base.triangles, base.camera_desc,
                 planes=base.planes, primitives=prims, bvh_nodes=new, bvh_depth=2)
    ref, ost = oracle.render(util.oracle_scene(sc), util.oracle_env(big_env), sc.camera_uniform().view(oracle.CAMERA), 96, 64, 0, 4, 10)
    img, st = gpu_render(sc, big_env, 96, 64, 0, 4, 10)
    assert np.array_equal(util.bits(img), util.bits(ref))
    assert (st["ext_rays"], st["shadow_rays"]) == (ost["ext_rays"], ost["shadow_rays"])


def test_bvh_whose_boxes_do_not_nest_keeps_the_tree_walk(big_env):
    """The flat small-scene loop is only valid when every child box lies inside its parent's.  Shrink an interior
    node's box so that its children stick out: the reference (and the oracle) then skip that subtree for rays that
    miss the shrunken box although they would hit a child — and so must the product."""
    base = R.Scene.load_toml(util.scene_path("default"))
    nodes = base.bvh_nodes.copy()
    interior = [i for i in range(1, len(nodes)) if nodes[i]["primitives_len"] == 0]
    i = interior[0]
    mid = (nodes[i]["bounds_min"] + nodes[i]["bounds_max"]) * np.float32(0.5)
    nodes[i]["bounds_min"] = mid + (nodes[i]["bounds_min"] - mid) * np.float32(0.5)
    nodes[i]["bounds_max"] = mid + (nodes[i]["bounds_max"] - mid) * np.float32(0.5)
    sc = R.Scene(base.materials, base.spheres, base.plane_descs, base.vertices, base.normals, base.triangles, base.camera_desc,
                 planes=base.planes, primitives=base.primitives, bvh_nodes=nodes, bvh_depth=base.bvh_depth)
    ref, ost = oracle.render(util.oracle_scene(sc), util.oracle_env(big_env), sc.camera_uniform().view(oracle.CAMERA), 96, 64, 0, 4, 10)
    intact, _ = oracle.render(util.oracle_scene(base), util.oracle_env(big_env), base.camera_uniform().view(oracle.CAMERA), 96, 64, 0, 4, 10)
    assert not np.array_equal(util.bits(ref), util.bits(intact))  # the shrunken box does change the picture
    img, st = gpu_render(sc, big_env, 96, 64, 0, 4, 10)
    assert np.array_equal(util.bits(img), util.bits(ref))
    assert (st["ext_rays"], st["shadow_rays"]) == (ost["ext_rays"], ost["shadow_rays"])


@pytest.mark.parametrize("traversal", ["6-noflat", "4", "4-noflat", "3", "3-q100", "3-noflat", "1", "0"])
def test_twin_records_in_different_leaves_tie_by_visiting_order(traversal, big_env, monkeypatch):
    """A sphere exists twice, with two materials, and its copies sit in DIFFERENT leaves (the builder would never do that — equal
    centroids share a leaf — so the BVH is edited by hand: the copy is appended to the last leaf, whose box and whose
    ancestors' boxes grow to hold it).  Every hit on it is a tie of equal t between two leaves; the reference keeps the one it
    visits first, which depends on the ray's sign octant, and the winner's material shows.  The flat traversal ("3": the octant's
    tabulated rank; "3-q100": with its triangle loop cut as often as can be), the fixed-order walk and the tree walks must agree
    with the oracle; the twin may NOT be dropped the way a twin inside one leaf is.  The cooperative walk ("6-noflat") folds hits with an
    atomic minimum over (t, record): it must notice the equal t and send such a ray through the exact walk again."""
    from rsoderh_raytracing_amd import types as T
    monkeypatch.setenv("RSRT_TRAVERSAL", traversal[0])
    if traversal.endswith("noflat"):
        monkeypatch.setenv("RSRT_FLAT", "0")
    if traversal.endswith("q100"):
        monkeypatch.setenv("RSRT_FLAT_QUORUM", "100")
    base = R.Scene.load_toml(util.scene_path("default"))
    nodes, prims = base.bvh_nodes.copy(), base.primitives
    k = next(i for i, p in enumerate(prims) if int(p["primitive_type"]) == 0)  # a sphere record ...
    src = int(prims[k]["index"])
    last = max((i for i in range(len(nodes)) if nodes[i]["primitives_len"] > 0), key=lambda i: int(nodes[i]["primitives_or_second_child_index"]))
    assert not (int(nodes[last]["primitives_or_second_child_index"]) <= k < int(nodes[last]["primitives_or_second_child_index"]) + int(nodes[last]["primitives_len"]))  # ... of another leaf
    spheres = np.zeros(len(base.spheres) + 1, T.SPHERE)  # (np.concatenate would drop the struct's padding)
    spheres[:-1] = base.spheres
    spheres[-1] = base.spheres[src]
    spheres[-1]["material_id"] = (int(spheres[src]["material_id"]) + 1) % len(base.materials)
    new_prims = np.concatenate([prims, np.array([(0, len(spheres) - 1)], prims.dtype)])
    lo = np.asarray(spheres[-1]["pos"], np.float32) - np.float32(spheres[-1]["radius"])
    hi = np.asarray(spheres[-1]["pos"], np.float32) + np.float32(spheres[-1]["radius"])
    nodes[last]["primitives_len"] += 1
    parent = {}
    for i in range(len(nodes)):
        if nodes[i]["primitives_len"] == 0:
            parent[i + 1] = i
            parent[int(nodes[i]["primitives_or_second_child_index"])] = i
    i = last
    while True:
        nodes[i]["bounds_min"] = np.minimum(nodes[i]["bounds_min"], lo)
        nodes[i]["bounds_max"] = np.maximum(nodes[i]["bounds_max"], hi)
        if i == 0:
            break
        i = parent[i]
    sc = R.Scene(base.materials, spheres, base.plane_descs, base.vertices, base.normals, base.triangles, base.camera_desc,
                 planes=base.planes, primitives=new_prims, bvh_nodes=nodes, bvh_depth=base.bvh_depth)
    ref, ost = oracle.render(util.oracle_scene(sc), util.oracle_env(big_env), sc.camera_uniform().view(oracle.CAMERA), 128, 80, 0, 4, 10)
    plain, _ = oracle.render(util.oracle_scene(base), util.oracle_env(big_env), base.camera_uniform().view(oracle.CAMERA), 128, 80, 0, 4, 10)
    assert not np.array_equal(util.bits(ref), util.bits(plain))  # the twin does win somewhere
    img, st = gpu_render(sc, big_env, 128, 80, 0, 4, 10)
    same = util.bits(img) == util.bits(ref)
    assert same.all(), (traversal, int((~same).any(axis=2).sum()))
    assert (st["ext_rays"], st["shadow_rays"]) == (ost["ext_rays"], ost["shadow_rays"])


def test_leaves_that_share_records_or_too_many_fallback_records_keep_the_tree_walk(big_env):
    """The flat loop carries a hit as a 6-bit record index and gives every record ONE visiting rank.  A foreign BVH
    whose leaves overlap (a record in two leaves: the reference simply tests it twice), or a scene with more than 64
    spheres behind a 64-record BVH, must fall back to the tree walk and still match the oracle."""
    from rsoderh_raytracing_amd import types as T
    base = R.Scene.load_toml(util.scene_path("default"))
    nodes = base.bvh_nodes.copy()
    leaves = [i for i in range(len(nodes)) if nodes[i]["primitives_len"] > 1]
    b_ = next(i for i in leaves[1:] if nodes[i]["primitives_or_second_child_index"] > 0)
    nodes[b_]["primitives_or_second_child_index"] -= 1  # leaf b now also holds its left neighbour's last record ...
    nodes[b_]["primitives_len"] += 1                    # (boxes untouched: they still nest, only the sharing disqualifies)
    sc = R.Scene(base.materials, base.spheres, base.plane_descs, base.vertices, base.normals, base.triangles, base.camera_desc,
                 planes=base.planes, primitives=base.primitives, bvh_nodes=nodes, bvh_depth=base.bvh_depth)
    ref, ost = oracle.render(util.oracle_scene(sc), util.oracle_env(big_env), sc.camera_uniform().view(oracle.CAMERA), 96, 64, 0, 4, 10)
    img, st = gpu_render(sc, big_env, 96, 64, 0, 4, 10)
    assert np.array_equal(util.bits(img), util.bits(ref))
    assert (st["ext_rays"], st["shadow_rays"]) == (ost["ext_rays"], ost["shadow_rays"])
    # 70 spheres, of which the BVH (and so `primitives`) knows only the first 10: cast_ray's brute-force loop still
    # visits all 70 after a BVH miss (shader.wgsl:583-590), and a hit on sphere 64.. does not fit 6 bits
    spheres = np.zeros(70, T.SPHERE)
    spheres[:10] = base.spheres
    for i in range(10, 70):
        spheres[i]["pos"] = (-6.0 + 0.2 * i, 2.5 + 0.03 * i, -4.0)
        spheres[i]["radius"] = 0.12
        spheres[i]["material_id"] = i % len(base.materials)
    sc2 = R.Scene(base.materials, spheres, base.plane_descs, base.vertices, base.normals, base.triangles, base.camera_desc,
                  planes=base.planes, primitives=base.primitives, bvh_nodes=base.bvh_nodes, bvh_depth=base.bvh_depth)
    ref2, ost2 = oracle.render(util.oracle_scene(sc2), util.oracle_env(big_env), sc2.camera_uniform().view(oracle.CAMERA), 96, 64, 0, 4, 10)
    plain, _ = oracle.render(util.oracle_scene(base), util.oracle_env(big_env), base.camera_uniform().view(oracle.CAMERA), 96, 64, 0, 4, 10)
    assert not np.array_equal(util.bits(ref2), util.bits(plain))  # the extra spheres are seen (through the fallback only)
    img2, st2 = gpu_render(sc2, big_env, 96, 64, 0, 4, 10)
    assert np.array_equal(util.bits(img2), util.bits(ref2))
    assert (st2["ext_rays"], st2["shadow_rays"]) == (ost2["ext_rays"], ost2["shadow_rays"])


@pytest.mark.parametrize("quorum", ["0", "20", "60", "100"])
def test_flat_triangle_vote_does_not_change_the_image(quorum, big_env, monkeypatch):
    """The flat traversal's triangle loop ends by a wave vote (RSRT_FLAT_QUORUM: once fewer than that percentage of the
    lanes that entered it still hold triangles); a ray that is cut short is re-queued with the triangles it has not
    tested and its best hit so far.  Which rays are cut, and how often, must be invisible: 0 never cuts, 100 cuts as
    soon as the first lane is through."""
    monkeypatch.setenv("RSRT_FLAT_QUORUM", quorum)
    sc = R.Scene.load_toml(util.scene_path("house"))
    ref, ost = oracle.render(util.oracle_scene(sc), util.oracle_env(big_env), sc.camera_uniform().view(oracle.CAMERA), 160, 90, 0, 4, 8)
    img, st = gpu_render(sc, big_env, 160, 90, 0, 4, 8)
    assert np.array_equal(util.bits(img), util.bits(ref))
    assert (st["ext_rays"], st["shadow_rays"]) == (ost["ext_rays"], ost["shadow_rays"])


@pytest.mark.parametrize("traversal", ["6-noflat", "4", "4-noflat", "3", "3-q100", "3-noflat", "1", "0"])
def test_coincident_primitives_resolve_ties_like_the_reference(traversal, big_env, monkeypatch):
    """Every primitive exists three times, at the same place, with three different materials, so every hit is a
    tie of equal t between records that usually sit in different leaves.  The reference keeps the first one it
    visits (strict `<`, near-child-first order, which depends on the ray's sign octant); the material of the
    winner is visible in the picture.  Exercises the visiting-order ranks of the flat traversal ("3": this scene has
    27 records) and of the fixed-order walk ("3-noflat"), and the position rule of the typed leaf loops; with a
    TRACE budget of 3 steps every traversal is cut and resumed many times, ties against an incumbent of an earlier
    call included.  Copies that share a leaf (equal centroids: most do) are the case the flat traversal settles at upload —
    a record that repeats an EARLIER record of its leaf bit for bit is left out of the leaf's mask, it could never win —
    so "3" also checks that it is the first copy's material that shows."""
    from rsoderh_raytracing_amd import host, types as T
    monkeypatch.setenv("RSRT_TRAVERSAL", traversal[0])
    monkeypatch.setenv("RSRT_TRACE_BUDGET", "1" if traversal[0] == "4" else "3")  # (the wide walk counts rounds: every ray parks its stack after one)
    if traversal.endswith("noflat"):
        monkeypatch.setenv("RSRT_FLAT", "0")
    if traversal.endswith("q100"):  # the flat traversal's triangle loop cut as often as can be: ties against the hit of an earlier call
        monkeypatch.setenv("RSRT_FLAT_QUORUM", "100")
    rng = np.random.default_rng(23)
    mats = np.zeros(3, T.MATERIAL)
    mats["color"] = [[0.9, 0.1, 0.1], [0.1, 0.9, 0.1], [0.1, 0.1, 0.9]]
    mats["roughness"] = [1.0, 0.4, 0.8]
    mats["metallic"] = [0.0, 0.5, 0.0]
    mats["emission"] = [[0.3, 0, 0], [0, 0.3, 0], [0, 0, 0.3]]
    ns, npl, nt = 2, 1, 6
    sph = np.zeros(3 * ns, T.SPHERE)
    pos, rad = rng.uniform(-2, 2, (ns, 3)), rng.uniform(0.4, 0.9, ns)
    pls = np.zeros(3 * npl, T.PLANE_DESC)
    ppos, pf, pr = np.array([[-4.0, -1.5, -4.0]]), np.array([[0.0, 0.0, 8.0]]), np.array([[8.0, 0.0, 0.0]])
    tv = np.round(rng.uniform(-3, 3, (nt, 3, 3)) * 2) / 2
    verts = np.zeros(9 * nt, T.VEC3)
    norms = np.zeros(9 * nt, T.VEC3)
    tri = np.zeros(3 * nt, T.TRIANGLE)
    order = rng.permutation(3 * nt)  # copies scattered through the arrays
    for copy in range(3):
        sph["pos"][copy::3], sph["radius"][copy::3], sph["material_id"][copy::3] = pos, rad, (copy + 1) % 3
        pls["pos"][copy::3], pls["forward"][copy::3], pls["right"][copy::3], pls["material_id"][copy::3] = ppos, pf, pr, (copy + 2) % 3
        for k in range(nt):
            j = int(order[copy * nt + k])
            verts["v"][3 * j:3 * j + 3] = tv[k]
            n = np.cross(tv[k][1] - tv[k][0], tv[k][2] - tv[k][0])
            norms["v"][3 * j:3 * j + 3] = n / max(np.linalg.norm(n), 1e-6)
            tri[j] = (3 * j, 3 * j + 1, 3 * j + 2, 3 * j, 3 * j + 1, 3 * j + 2, copy)
    cam = host.make_camera_desc([0.3, 1.0, 6.0], yaw=0.05, pitch=-0.1, fov_y=1.2)
    sc = R.Scene(mats, sph, pls, verts, norms, tri, cam)
    ref, ost = oracle.render(util.oracle_scene(sc), util.oracle_env(big_env), sc.camera_uniform().view(oracle.CAMERA), 96, 64, 0, 6, 6)
    img, st = gpu_render(sc, big_env, 96, 64, 0, 6, 6)
    same = util.bits(img) == util.bits(ref)
    assert same.all(), (traversal, int((~same).any(axis=2).sum()))
    assert (st["ext_rays"], st["shadow_rays"]) == (ost["ext_rays"], ost["shadow_rays"])


@pytest.mark.parametrize("hybrid,traversal", [("1", "6"), ("0", "6"), ("1", "4"), ("0", "4"), ("1", "3"), ("0", "3"), ("1", "1"), ("0", "1")])
def test_mid_size_scene_with_nodes_in_lds_or_in_global_memory(hybrid, traversal, big_env, monkeypatch):
    """suzanne (968 triangles): too big for the LDS image; by default what its box steps touch — the pre-order nodes of
    the fixed-order walk, or the nodes and escape links of the tree walk (RSRT_TRAVERSAL=1) — is staged in LDS for one
    1024-thread workgroup per CU (RSRT_HYBRID=0: everything from global memory).  Same bits every way."""
    monkeypatch.setenv("RSRT_HYBRID", hybrid)
    monkeypatch.setenv("RSRT_TRAVERSAL", traversal)
    sc = R.Scene.load_toml(util.scene_path("suzanne"))
    ref, ost = oracle.render(util.oracle_scene(sc), util.oracle_env(big_env), sc.camera_uniform().view(oracle.CAMERA), 120, 68, 0, 4, 10)
    img, st = gpu_render(sc, big_env, 120, 68, 0, 4, 10)
    assert np.array_equal(util.bits(img), util.bits(ref))
    assert (st["ext_rays"], st["shadow_rays"]) == (ost["ext_rays"], ost["shadow_rays"])


def test_big_scene_all_global_is_bit_exact(big_env):
    """The builder-authored suzanne grid (tools/make_big_scene.py: 15,488 triangles, ~8.7 k nodes): nothing of it fits
    the LDS budget, so the fixed-order walk reads nodes and records from global memory.  Reduced size, bit for bit."""
    import sys
    sys.path.insert(0, util.ROOT + "/tools")
    import make_big_scene
    sc = R.Scene.load_toml(make_big_scene.make(4))
    assert len(sc.triangles) == 15488 and len(sc.bvh_nodes) > 8000
    ref, ost = oracle.render(util.oracle_scene(sc), util.oracle_env(big_env), sc.camera_uniform().view(oracle.CAMERA), 200, 112, 0, 3, 10, fast=True)
    img, st = gpu_render(sc, big_env, 200, 112, 0, 3, 10)
    assert np.array_equal(util.bits(img), util.bits(ref))
    assert (st["ext_rays"], st["shadow_rays"]) == (ost["ext_rays"], ost["shadow_rays"])
    # the ray probe on the same scene: fixed-order walk and tree walks agree with the oracle's cast_ray
    rng = np.random.default_rng(5)
    o = rng.uniform(-6, 6, (4096, 3)).astype(np.float32) + np.float32([0, 2, 2])
    d = rng.normal(size=(4096, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    hits = oracle.cast_rays(util.oracle_scene(sc), o, d, 0, 0)
    s2 = R.State.new(sc, util.small_env(), 16, 16)
    for mode in (0 << 1, 2 << 1, 4 << 1, 5 << 1, (5 << 1) | 16, 6 << 1, (6 << 1) | 16):
        got = s2.cast_rays(o, d, mode, 0)
        assert np.array_equal(np.ascontiguousarray(got).view(np.uint32).reshape(-1, 9), hits.view(np.uint32).reshape(-1, 9)), mode
    s2.close()


@pytest.mark.parametrize("levels,budget,hybrid,kernel", [(42, "", "1", "4"), (60, "", "1", "4"), (60, "1", "1", "4"), (60, "1", "0", "4"), (60, "1", "0", "1"), (60, "", "0", "3"),
                                                         (42, "coop", "1", "4"), (60, "coop", "1", "4"), (60, "coop", "0", "4")])
def test_wide_walk_stack_overflow_on_a_chain_tree(levels, budget, hybrid, kernel, big_env, monkeypatch):
    """tests/util.py deck_scene: a hand-built chain tree whose wide form has levels / 3 levels, on which a ray along the deck holds up to
    13 (42 levels) or 19 (60) stack words — more than the walk has registers (test_wide_tree.py shows that on the CPU).  The image, seen
    along the deck, and a batch of probe rays must be the oracle's bit for bit: with the default budget, and with one round per TRACE
    call, so that rays are parked and resumed while words sit in the overflow area.  42 levels: the scene's whole image fits LDS (scene
    view 1); 60: nodes staged in LDS (view 2) or everything in global memory (RSRT_HYBRID=0, view 0) with 160 / 192 / 128-slot pools."""
    if budget == "coop":  # the same trees through the cooperative walk, which has no per-ray stack to overflow (its work stacks are the wave's)
        monkeypatch.setenv("RSRT_TRAVERSAL", "6")
    else:
        monkeypatch.setenv("RSRT_TRAVERSAL", "5")
        if budget:
            monkeypatch.setenv("RSRT_TRACE_BUDGET", budget)
    monkeypatch.setenv("RSRT_HYBRID", hybrid)
    monkeypatch.setenv("RSRT_KERNEL", kernel)
    monkeypatch.setenv("RSRT_FLAT", "0")
    sc = util.deck_scene(levels)
    ref, ost = oracle.render(util.oracle_scene(sc), util.oracle_env(big_env), sc.camera_uniform().view(oracle.CAMERA), 96, 64, 0, 4, 10)
    img, st = gpu_render(sc, big_env, 96, 64, 0, 4, 10)
    assert np.array_equal(util.bits(img), util.bits(ref))
    assert (st["ext_rays"], st["shadow_rays"]) == (ost["ext_rays"], ost["shadow_rays"])
    assert ost["ext_rays"] > 96 * 64 * 4 * 1.2  # (the deck is hit: paths bounce)
    rng = np.random.default_rng(9)
    o = (rng.uniform(-0.5, 0.5, (2048, 3)) + [0, 0, 3]).astype(np.float32)
    d = (rng.normal(size=(2048, 3)) * 0.05 + [0, 0, -1]).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    hits = oracle.cast_rays(util.oracle_scene(sc), o, d, 0, 0)
    assert hits["did_hit"].sum() > 1024
    s2 = R.State.new(sc, util.small_env(), 16, 16)
    for mode in (4 << 1, 5 << 1, 6 << 1):
        got = s2.cast_rays(o, d, mode, 0)
        assert np.array_equal(np.ascontiguousarray(got).view(np.uint32).reshape(-1, 9), hits.view(np.uint32).reshape(-1, 9)), mode
    s2.close()


@pytest.mark.parametrize("budget", ["", "1", "coop"])
def test_scene_with_a_deep_tree_takes_the_wide_walk(budget, big_env, monkeypatch):
    """suzanne on an 8 x 8 grid (61,952 triangles, binary depth 19): the wide tree has eleven levels, more than the walk's eight stack
    registers can always serve, so the scene runs the kernel variant whose stack may overflow into memory (TRAV 5) — on THIS scene no
    ray ever holds more than eight words (the chain tree above is the test of the overflow itself); what is checked here is that a
    scene of this size takes the wide walk at all (a third of the fixed-order walk's steps), bit for bit against the oracle, also with
    one round per TRACE call (RSRT_TRACE_BUDGET=1), and through the probe, in global memory and with the top of the tree staged in LDS."""
    import sys
    sys.path.insert(0, util.ROOT + "/tools")
    import make_big_scene
    if budget == "coop":  # (the product's choice for this scene since round 4: the cooperative walk, which needs no stack variant)
        monkeypatch.setenv("RSRT_TRAVERSAL", "6")
    else:
        monkeypatch.setenv("RSRT_TRAVERSAL", "5")
        if budget:
            monkeypatch.setenv("RSRT_TRACE_BUDGET", budget)
    sc = R.Scene.load_toml(make_big_scene.make(8))
    assert len(sc.triangles) == 61952 and sc.bvh_depth >= 18
    ref, ost = oracle.render(util.oracle_scene(sc), util.oracle_env(big_env), sc.camera_uniform().view(oracle.CAMERA), 160, 90, 0, 2, 10, fast=True)
    img, st = gpu_render(sc, big_env, 160, 90, 0, 2, 10)
    assert np.array_equal(util.bits(img), util.bits(ref))
    assert (st["ext_rays"], st["shadow_rays"]) == (ost["ext_rays"], ost["shadow_rays"])
    wide_steps = st["traversal_steps"] / (st["ext_rays"] + st["shadow_rays"])
    monkeypatch.setenv("RSRT_TRAVERSAL", "3")
    img3, st3 = gpu_render(sc, big_env, 160, 90, 0, 2, 10)
    assert np.array_equal(util.bits(img3), util.bits(ref))
    assert wide_steps < 0.5 * st3["traversal_steps"] / (st3["ext_rays"] + st3["shadow_rays"]), wide_steps
    if budget in ("", "coop"):
        rng = np.random.default_rng(6)
        o = rng.uniform(-12, 12, (4096, 3)).astype(np.float32) + np.float32([0, 2, -8])
        d = rng.normal(size=(4096, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        hits = oracle.cast_rays(util.oracle_scene(sc), o, d, 0, 0)
        s2 = R.State.new(sc, util.small_env(), 16, 16)
        for mode in (4 << 1, 5 << 1, (5 << 1) | 16, 6 << 1, (6 << 1) | 16):
            got = s2.cast_rays(o, d, mode, 0)
            assert np.array_equal(np.ascontiguousarray(got).view(np.uint32).reshape(-1, 9), hits.view(np.uint32).reshape(-1, 9)), mode
        s2.close()


@pytest.mark.parametrize("lds_cap,lifo_at,narrow_at", [("320", "", ""), ("320", "3072", ""), ("320", "64", "40"), ("", "", "0"), ("", "0", "")])
def test_cooperative_walk_spills_and_narrow_trips_do_not_change_the_image(lds_cap, lifo_at, narrow_at, big_env, monkeypatch):
    """The cooperative walk's node queue (rt_coop.h): with its LDS ring capped at the minimum (RSRT_COOP_LDS_CAP=320) a batch of up to 256 rays
    spills its newest items to the wave's arena block and takes them back — all the more when the wave never turns to newest-first
    (RSRT_COOP_LIFO_AT=3072; 0: it always pops newest first, the first version of the walk); with RSRT_COOP_NARROW_AT small a wave pops ONE
    item a trip whenever more than that many items are outstanding (0: always — a plain depth-first walk of the whole batch).  Which items
    travel together, and in which order, must be invisible: suzanne and the 15 k-triangle grid, reduced frames, bit for bit against the oracle,
    and the probe."""
    import sys
    sys.path.insert(0, util.ROOT + "/tools")
    import make_big_scene
    if lds_cap:
        monkeypatch.setenv("RSRT_COOP_LDS_CAP", lds_cap)
    if lifo_at:
        monkeypatch.setenv("RSRT_COOP_LIFO_AT", lifo_at)
    if narrow_at:
        monkeypatch.setenv("RSRT_COOP_NARROW_AT", narrow_at)
    for sc, w, h, spp in [(R.Scene.load_toml(util.scene_path("suzanne")), 96, 64, 3), (R.Scene.load_toml(make_big_scene.make(4)), 120, 68, 2)]:
        ref, ost = oracle.render(util.oracle_scene(sc), util.oracle_env(big_env), sc.camera_uniform().view(oracle.CAMERA), w, h, 0, spp, 10, fast=True)
        img, st = gpu_render(sc, big_env, w, h, 0, spp, 10)
        assert np.array_equal(util.bits(img), util.bits(ref))
        assert (st["ext_rays"], st["shadow_rays"]) == (ost["ext_rays"], ost["shadow_rays"])
        rng = np.random.default_rng(8)
        o = rng.uniform(-6, 6, (2048, 3)).astype(np.float32) + np.float32([0, 2, 2])
        d = rng.normal(size=(2048, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        hits = oracle.cast_rays(util.oracle_scene(sc), o, d, 0, 0)
        s2 = R.State.new(sc, util.small_env(), 16, 16)
        for mode in (6 << 1, (6 << 1) | 16):
            got = s2.cast_rays(o, d, mode, 0)
            assert np.array_equal(np.ascontiguousarray(got).view(np.uint32).reshape(-1, 9), hits.view(np.uint32).reshape(-1, 9)), mode
        s2.close()


def _merge_sibling_leaves(nodes, limit=8):
    """The same BVH with every pair of sibling leaves that together hold <= `limit` records merged into their parent (repeatedly): a valid
    foreign BVH in the reference's layout (pre-order, first child = i + 1) whose leaves hold up to `limit` records."""
    out = []

    def leaf_run(i):  # (first record, count) if the subtree at i can be ONE leaf of <= limit records, else None
        n = nodes[i]
        if n["primitives_len"] > 0:
            return int(n["primitives_or_second_child_index"]), int(n["primitives_len"])
        a, b = leaf_run(i + 1), leaf_run(int(n["primitives_or_second_child_index"]))
        if a and b and a[0] + a[1] == b[0] and a[1] + b[1] <= limit:
            return a[0], a[1] + b[1]
        return None

    def emit(i):
        me = len(out)
        out.append(nodes[i].copy())
        run = leaf_run(i)
        if run:
            out[me]["primitives_or_second_child_index"], out[me]["primitives_len"], out[me]["split_axis"] = run[0], run[1], 0
            return
        emit(i + 1)
        out[me]["primitives_or_second_child_index"] = len(out)
        emit(int(nodes[i]["primitives_or_second_child_index"]))

    import sys
    sys.setrecursionlimit(10000)
    emit(0)
    return np.array(out, nodes.dtype)


def test_cooperative_walk_with_leaves_of_up_to_eight_records(big_env, monkeypatch):
    """A leaf trip of the cooperative walk (rt_coop.h) spreads the records of the items it pops over the lanes by a prefix sum of their record
    counts — 4-bit numbers, 1..8.  The reference's builder stops at five records a leaf; a foreign BVH may hold eight: suzanne's tree with sibling
    leaves merged while they fit (leaves of 6, 7 and 8 records appear, the count's fourth bit is set), through the walk (RSRT_FLAT=0 keeps the flat
    loop out) and the probe, bit for bit against the oracle."""
    monkeypatch.setenv("RSRT_TRAVERSAL", "6")
    monkeypatch.setenv("RSRT_FLAT", "0")
    base = R.Scene.load_toml(util.scene_path("suzanne"))
    nodes = _merge_sibling_leaves(base.bvh_nodes)
    lens = nodes["primitives_len"][nodes["primitives_len"] > 0]
    assert lens.max() == 8 and {6, 7, 8} <= set(int(x) for x in lens) and lens.sum() == len(base.primitives)
    depth = 0
    stack = [(0, 1)]
    while stack:
        i, dd = stack.pop()
        depth = max(depth, dd)
        if nodes[i]["primitives_len"] == 0:
            stack += [(i + 1, dd + 1), (int(nodes[i]["primitives_or_second_child_index"]), dd + 1)]
    sc = R.Scene(base.materials, base.spheres, base.plane_descs, base.vertices, base.normals, base.triangles, base.camera_desc,
                 planes=base.planes, primitives=base.primitives, bvh_nodes=nodes, bvh_depth=depth)
    ref, ost = oracle.render(util.oracle_scene(sc), util.oracle_env(big_env), sc.camera_uniform().view(oracle.CAMERA), 96, 64, 0, 3, 10)
    img, st = gpu_render(sc, big_env, 96, 64, 0, 3, 10)
    assert np.array_equal(util.bits(img), util.bits(ref))
    assert (st["ext_rays"], st["shadow_rays"]) == (ost["ext_rays"], ost["shadow_rays"])
    rng = np.random.default_rng(9)
    o = rng.uniform(-3, 3, (4096, 3)).astype(np.float32) + np.float32([0, 1, 1])
    d = rng.normal(size=(4096, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    hits = oracle.cast_rays(util.oracle_scene(sc), o, d, 0, 0)
    s2 = R.State.new(sc, util.small_env(), 16, 16)
    for mode in (6 << 1, (6 << 1) | 16):
        got = s2.cast_rays(o, d, mode, 0)
        assert np.array_equal(np.ascontiguousarray(got).view(np.uint32).reshape(-1, 9), hits.view(np.uint32).reshape(-1, 9)), mode
    s2.close()


@pytest.mark.parametrize("name,w,h,spp,mb", [("house", 128, 72, 6, 8), ("default", 96, 64, 4, 10), ("suzanne", 80, 48, 3, 10)])
def test_two_pipelines_that_share_only_the_asset_files(name, w, h, spp, mb):
    """Every other image test feeds the oracle the product's OWN preprocessing output (tests/util.py: same BVH, same alias table, same
    plane matrices on both sides), which would hide a mistake common to the C++ preprocessing and whatever consumes it.  Here the two
    sides share nothing but the TOML / OBJ files and the environment's texels: the oracle side is the Python scene reader + the oracle's
    own build_bvh / plane_to_uniform / camera_uniform / alias table (oracle/scene_py.py, rt_oracle.cpp), the product side is
    librsrt_host's loader and builders + the alias table built ON THE DEVICE + the kernels.  Same bits."""
    from oracle import scene_py
    o = scene_py.load_toml(util.scene_path(name))
    env_rgba = R.Environment.synthetic(128, 64).rgba  # (the frozen synthetic-sky formula; its texels are the shared input)
    alias, _ = oracle.alias_table(env_rgba[..., :3])
    ref, ost = oracle.render(o["scene"], oracle.Env(env_rgba, alias), o["camera"].view(oracle.CAMERA), w, h, 0, spp, mb)
    sc = R.Scene.load_toml(util.scene_path(name))
    st = R.State.new(sc, R.Environment(env_rgba.copy()), w, h)  # (host-built alias table uploaded ...)
    L = R.state.lib()
    L.rsrt_environment_build_alias.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    assert L.rsrt_environment_build_alias(st._ctx, 0, None, 0, None) == 0  # (... and replaced by the one the device builds from the texels)
    st.max_bounces = mb
    st.render_range(0, spp)
    img, g = st.download(), st.stats()
    st.close()
    assert np.array_equal(util.bits(img), util.bits(ref))
    assert (g["paths"], g["ext_rays"], g["shadow_rays"]) == (ost["paths"], ost["ext_rays"], ost["shadow_rays"])
