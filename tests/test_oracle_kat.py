"""Pins the CPU oracle with the hand-derivable known-answer values of SURVEY.md §8(c).

The reference has no tests or golden vectors (parity unpinned by the reference); these KATs are
derived by hand from the reference source and are the external pins of the oracle.
"""
import ctypes as C

import numpy as np
import pytest

import oracle
import util
import rsoderh_raytracing_amd as R

# (pixel, sample) -> state after both salts, next four random_u32_uniform  (shader.wgsl:605-623)
RNG_KATS = [
    (0, 0, 0x4712A88E, [0x3BF6E0B1, 0x572F7439, 0x86FC4DDC, 0x34659162]),
    (1, 0, 0x41806E87, [0x97AAF6C6, 0x2A521372, 0xD0A714DD, 0x262CBC0B]),
    (0, 1, 0x1A8030D9, [0x3D1E3413, 0x95181899, 0xACD8F3A5, 0x4E6343DF]),
    (1037760, 255, 0xD52F5107, [0x0D5AF2CF, 0x779C4BAD, 0x447ECC66, 0x5628D0E7]),
    (2073599, 255, 0x22F9D2C0, [0xADB5FABC, 0x303520F0, 0xE3F4C383, 0xB5E4A678]),
]


@pytest.mark.parametrize("pixel,sample,state,draws", RNG_KATS)
def test_rng_seed_and_draws(pixel, sample, state, draws):
    s = oracle.rng_seed(pixel, sample)
    assert s == state
    got, _ = oracle.rng_draws(s, 4)
    assert got == draws


def test_rng_python_integer_model():
    """Same generator in Python ints (independent of the C code)."""
    def step(s):
        s = (s * 747796405 + 2891336453) & 0xFFFFFFFF
        r = (((s >> ((s >> 28) + 4)) ^ s) * 277803737) & 0xFFFFFFFF
        return s, (r >> 22) ^ r
    for pixel, sample in [(0, 0), (17, 3), (123456, 77), (2073599, 1023)]:
        s = 0
        s, _ = step(s ^ pixel)
        s, _ = step(s ^ sample)
        assert oracle.rng_seed(pixel, sample) == s
        draws = []
        for _ in range(6):
            s, r = step(s)
            draws.append(r)
        assert oracle.rng_draws(oracle.rng_seed(pixel, sample), 6)[0] == draws


def test_random_uniform_edges():
    L = oracle.lib()
    assert L.orc_u32_to_uniform(0x3BF6E0B1) == pytest.approx(0.234235808, abs=1e-9)
    assert np.float32(L.orc_u32_to_uniform(0xFFFFFF7F)) == np.float32(0.99999994)
    assert L.orc_u32_to_uniform(0xFFFFFF80) == 1.0  # [0,1] inclusive: f32(r) rounds to 2^32
    assert L.orc_u32_to_uniform(0xFFFFFFFF) == 1.0
    assert L.orc_u32_to_uniform(0) == 0.0


def test_constants_and_strides():
    assert np.float32(3.14159) == np.float32(3.14159012)
    assert np.float32(1.0 / 3.14159) == np.float32(0.31831014)
    assert np.float32(1.70141183460469231732e38) == np.float32(2.0 ** 127)
    assert (oracle.MATERIAL.itemsize, oracle.SPHERE.itemsize, oracle.PLANE.itemsize, oracle.TRIANGLE.itemsize,
            oracle.PRIM_INFO.itemsize, oracle.BVH_NODE.itemsize, oracle.VEC3.itemsize, oracle.ALIAS_ENTRY.itemsize,
            oracle.CAMERA.itemsize) == (48, 32, 96, 28, 8, 48, 16, 16, 80)


def test_alias_table_equal_luminance():
    rgb = np.ones((1, 4, 3), np.float32)
    al, left = oracle.alias_table(rgb)
    assert left == 4
    assert list(al["probability"]) == [1.0] * 4
    assert list(al["alias_index"]) == [0, 1, 2, 3]
    assert list(al["pmf"]) == [0.25] * 4


def test_alias_table_two_pixels():
    rgb = np.zeros((1, 2, 3), np.float32)
    rgb[0, 0] = 1.0  # luminance 1 (weights 0.2126+0.7152+0.0722 = 1 up to rounding)
    rgb[0, 1] = 3.0
    al, left = oracle.alias_table(rgb)
    assert al["probability"][0] == pytest.approx(0.5, abs=1e-6) and al["alias_index"][0] == 1
    assert al["pmf"][0] == pytest.approx(0.25, abs=1e-6)
    # the large pixel is never demoted -> leftover default, pmf 1/N = 0.5 (true 0.75): the reference's quirk
    assert al["probability"][1] == 1.0 and al["alias_index"][1] == 1 and al["pmf"][1] == 0.5
    assert left == 1


@pytest.mark.parametrize("name,nodes,depth", [("house", 39, 6), ("default", 15, 5), ("cube", 9, 4), ("suzanne", 549, 12)])
def test_bvh_shape(name, nodes, depth):
    sc = R.Scene.load_toml(util.scene_path(name))
    osc = util.oracle_scene(sc)
    prims, nd, d = oracle.build_bvh(osc.spheres, sc.plane_descs.view(oracle.PLANE_SRC), osc.vertices, osc.triangles)
    assert (len(nd), d) == (nodes, depth)
    n_prims = len(osc.spheres) + len(osc.planes) + len(osc.triangles)
    leaves = nd[nd["len"] > 0]
    assert leaves["len"].sum() == n_prims and leaves["len"].max() <= 5 and leaves["len"].min() >= 1
    assert sorted(map(tuple, prims.tolist())) == sorted(
        [(0, i) for i in range(len(osc.spheres))] + [(1, i) for i in range(len(osc.planes))] +
        [(2, i) for i in range(len(osc.triangles))])
    inner = np.nonzero(nd["len"] == 0)[0]
    assert np.all(nd["idx"][inner] > inner + 1) and np.all(nd["idx"][inner] < len(nd))


def test_suzanne_alone_545_nodes():
    sc = R.Scene.load_toml(util.scene_path("suzanne"))
    osc = util.oracle_scene(sc)
    _, nd, d = oracle.build_bvh(osc.spheres, np.zeros(0, oracle.PLANE_SRC), osc.vertices, osc.triangles)
    assert (len(nd), d) == (545, 12)


def _hit(fn, o, d, *args):
    out = np.zeros(1, oracle.HIT)
    fn(np.float32(o).ctypes.data_as(C.c_void_p), np.float32(d).ctypes.data_as(C.c_void_p), *args,
       out.ctypes.data_as(C.c_void_p))
    return out[0]


def test_analytic_sphere_hit():
    s = np.zeros(1, oracle.SPHERE)
    s["radius"] = 1.0
    h = _hit(oracle.lib().orc_cast_ray_sphere, [0, 0, 3], [0, 0, -1], s.ctypes.data_as(C.c_void_p))
    assert h["did_hit"] == 1 and h["distance"] == 2.0
    assert list(h["hit_point"]) == [0, 0, 1] and list(h["normal"]) == [0, 0, 1]


def test_analytic_triangle_hit():
    verts = np.zeros(3, oracle.VEC3)
    verts["v"] = [[-1, -1, 0], [1, -1, 0], [0, 1, 0]]
    norms = np.zeros(1, oracle.VEC3)
    norms["v"] = [[0, 0, 1]]
    tri = np.zeros(1, oracle.TRIANGLE)
    tri["v1"], tri["v2"] = 1, 2
    sc = oracle.Scene(materials=np.zeros(1, oracle.MATERIAL), spheres=np.zeros(0, oracle.SPHERE), planes=np.zeros(0, oracle.PLANE),
                      vertices=verts, normals=norms, triangles=tri, prims=np.zeros(0, oracle.PRIM_INFO),
                      nodes=np.zeros(0, oracle.BVH_NODE))
    h = _hit(oracle.lib().orc_cast_ray_triangle, [0, 0, 3], [0, 0, -1], C.byref(sc.c), tri.ctypes.data_as(C.c_void_p))
    assert h["did_hit"] == 1 and h["distance"] == 3.0  # u = 0.25, v = 0.5
    assert list(h["hit_point"]) == [0, 0, 0] and list(h["normal"]) == [0, 0, 1]


def test_analytic_plane_hit():
    src = np.zeros(1, oracle.PLANE_SRC)
    src["pos"], src["forward"], src["right"] = [-30, 0, -30], [0, 0, 60], [60, 0, 0]  # house.toml ground
    pl = oracle.plane_to_uniform(src)
    assert list(pl["normal"][0]) == [0, 1, 0]
    h = _hit(oracle.lib().orc_cast_ray_plane, [0, 1, 3], [0, -1, 0], pl.ctypes.data_as(C.c_void_p))
    assert h["did_hit"] == 1 and h["distance"] == 1.0 and list(h["hit_point"]) == [0, 0, 3] and list(h["normal"]) == [0, 1, 0]
    # plane-space coordinates of the hit: M * (hit - pos) = (0.5, 0, 0.55)
    m = pl["m"][0][:, :3].T  # columns -> matrix
    ps = m @ (np.array([0, 0, 3.0]) - np.array([-30, 0, -30.0]))
    assert ps == pytest.approx([0.5, 0, 0.55], abs=1e-6)
    # below the plane: normal flips on dot(origin, n) < 0
    h = _hit(oracle.lib().orc_cast_ray_plane, [0, -1, 3], [0, 1, 0], pl.ctypes.data_as(C.c_void_p))
    assert list(h["normal"]) == [0, -1, 0]


def test_camera_identity_and_fov():
    cam = oracle.camera_uniform([0, 1, 3], 0.0, 0.0, 1.7453293)
    assert np.array_equal(cam["rot"][0][:, :3], np.eye(3, dtype=np.float32))
    cam = oracle.camera_uniform([0, 0, 0], np.pi / 2, 0.0, 1.0)  # yaw 90 deg: -Z view axis turns to -X
    fwd = cam["rot"][0][:, :3].T @ np.array([0, 0, -1.0])
    assert fwd == pytest.approx([-1, 0, 0], abs=1e-6)


def test_white_furnace_bound():
    """E[f cos / pdf] over bsdf_sample <= 1 (+ noise) for every material of both scenes."""
    L = oracle.lib()
    for name in ("house", "default"):
        sc = R.Scene.load_toml(util.scene_path(name))
        mats = sc.materials.view(oracle.MATERIAL)
        for mi in range(len(mats)):
            m = mats[mi:mi + 1]
            rng = C.c_uint32(12345 + mi)
            n = np.float32([0, 0, 1])
            wo = np.float32([0.3, 0.2, 0.0])
            wo[2] = np.sqrt(1 - wo[0] ** 2 - wo[1] ** 2)
            ray_dir = (-wo).astype(np.float32)
            acc = np.zeros(3)
            N = 4000
            for _ in range(N):
                d, s = np.zeros(3, np.float32), np.zeros(3, np.float32)
                pdf = L.orc_bsdf_sample(m.ctypes.data_as(C.c_void_p), ray_dir.ctypes.data_as(C.c_void_p), n.ctypes.data_as(C.c_void_p),
                                        C.byref(rng), d.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p))
                if pdf > 0 and np.any(d != 0):
                    acc += s * max(0.0, float(d[2])) / pdf
            assert np.all(acc / N <= 1.08), (name, mi, acc / N)


def test_render_resume_is_exact():
    """samples [0,a) then [a,b) added into the same sum == [0,b) in one call (sample_begin contract)."""
    env = util.oracle_env(util.small_env())
    sc = R.Scene.load_toml(util.scene_path("default"))
    osc, cam = util.oracle_scene(sc), sc.camera_uniform().view(oracle.CAMERA)
    one, _ = oracle.render(osc, env, cam, 32, 24, 0, 6, 5)
    two, _ = oracle.render(osc, env, cam, 32, 24, 0, 2, 5)
    two, _ = oracle.render(osc, env, cam, 32, 24, 2, 4, 5, sum_rgba=two)
    assert np.array_equal(util.bits(one), util.bits(two))


def test_anyhit_shadow_is_exact_and_pruning_is_only_near_exact():
    """The any-hit exit of the shadow query cannot change a pixel (the shader reads only did_hit).
    t-pruning almost never does, but it is NOT exact: on house.toml 1920x1080 sample 149 of pixel
    (851,477) and sample 240 of pixel (1437,1068) it changes the path (found on the GPU box at
    256 spp, reproduced here) — which is why the product does not prune by default."""
    env = util.oracle_env(util.small_env())
    for name in ("house", "default", "suzanne"):
        sc = R.Scene.load_toml(util.scene_path(name))
        osc, cam = util.oracle_scene(sc), sc.camera_uniform().view(oracle.CAMERA)
        a, sa = oracle.render(osc, env, cam, 48, 32, 0, 4, 10)
        b, sb = oracle.render(osc, env, cam, 48, 32, 0, 4, 10, flags=oracle.FLAG_ANYHIT_SHADOW)
        c, sc_ = oracle.render(osc, env, cam, 48, 32, 0, 4, 10, flags=oracle.FLAG_PRUNE | oracle.FLAG_ANYHIT_SHADOW)
        assert np.array_equal(util.bits(a), util.bits(b))
        assert sb["nodes_visited"] < sa["nodes_visited"] and sc_["nodes_visited"] < sb["nodes_visited"]
        assert (sa["ext_rays"], sa["shadow_rays"]) == (sb["ext_rays"], sb["shadow_rays"])
        assert np.all(util.rmse_per_channel(a, c, 4) <= 1e-3)


def test_pruning_counterexample_house_1080p():
    import rsoderh_raytracing_amd as R2
    env = util.oracle_env(R2.Environment.synthetic(2048, 1024))
    sc = R.Scene.load_toml(util.scene_path("house"))
    osc, cam = util.oracle_scene(sc), sc.camera_uniform().view(oracle.CAMERA)
    exact, _ = oracle.render(osc, env, cam, 1920, 1080, 149, 1, 8, fast=True)
    anyhit, _ = oracle.render(osc, env, cam, 1920, 1080, 149, 1, 8, flags=oracle.FLAG_ANYHIT_SHADOW, fast=True)
    pruned, _ = oracle.render(osc, env, cam, 1920, 1080, 149, 1, 8, flags=oracle.FLAG_PRUNE, fast=True)
    assert np.array_equal(util.bits(exact), util.bits(anyhit))
    assert list(exact[477, 851, :3]) == [0.0, 0.0, 0.0]
    assert not np.array_equal(util.bits(exact[477, 851]), util.bits(pruned[477, 851]))
