"""build_bvh on the device (rsrt_build_bvh_device, csrc/hip/rt_bvh_device.h) against the host builder
(rsrt_build_bvh), which the CPU suite pins against the oracle's literal restatement of src/bvh.rs:215-337: the same
`primitives` and the same node array, node for node — bounds, child / primitive indices, lengths, split axes — and the
same depth.  The CPU part checks the closed form of the reference's unstable two-pointer partition that the device
builder rests on."""
import random
import sys

import numpy as np
import pytest

import util
import rsoderh_raytracing_amd as R
from rsoderh_raytracing_amd import host, types as T


def sequential_partition(cls):
    """The reference's loop (src/bvh.rs:304-315): `split` walks up, a RIGHT element is swapped with the last unplaced one."""
    p, s, e = list(range(len(cls))), 0, len(cls)
    while s < e:
        if cls[p[s]]:
            s += 1
        else:
            e -= 1
            p[s], p[e] = p[e], p[s]
    return p, s


def closed_form_partition(cls):
    """rt_bvh_level_kernel's arithmetic: one prefix count of the LEFT flags -> every item's destination."""
    n, L = len(cls), sum(cls)
    pre = [sum(cls[:i]) for i in range(n + 1)]
    m = L - pre[L]
    holes, backs = [None] * m, [None] * m
    for i in range(n):
        if i < L and not cls[i]:
            holes[i - pre[i]] = i
        if i >= L and cls[i]:
            backs[m - 1 - (pre[i] - pre[L])] = i
    b_m = backs[m - 1] if m else n
    out = [None] * n
    for i in range(n):
        if i < L:
            to = i if cls[i] else ((n if i - pre[i] == 0 else backs[i - pre[i] - 1]) - 1)
        else:
            to = holes[m - (pre[i] - pre[L]) - 1] if cls[i] else (b_m - 1 if i == L else i - 1)
        assert out[to] is None
        out[to] = i
    return out, L


def test_closed_form_of_the_two_pointer_partition():
    rng = random.Random(3)
    for _ in range(20000):
        n, p = rng.randint(1, 40), rng.random()
        cls = [rng.random() < p for _ in range(n)]
        assert closed_form_partition(cls) == sequential_partition(cls)
    for cls in ([True], [False], [True] * 7, [False] * 7, [False, True], [True, False]):
        assert closed_form_partition(cls) == sequential_partition(cls)


def same_tree(a, b):
    (p1, n1, d1), (p2, n2, d2) = a, b
    return util.fields_equal(p1, p2) and util.fields_equal(n1, n2) and d1 == d2


@pytest.mark.gpu
def test_device_builder_equals_the_host_builder_node_for_node():
    sys.path.insert(0, util.ROOT + "/tools")
    import make_big_scene
    st = R.State.new(R.Scene.load_toml(util.scene_path("default")), util.small_env(), 16, 16)
    report = []
    for name in ["house", "default", "cube", "spheres_only", "suzanne", make_big_scene.make(4)]:
        sc = R.Scene.load_toml(name if name.endswith(".toml") else util.scene_path(name))
        ref = host.build_bvh(sc.spheres, sc.plane_descs, sc.vertices, sc.triangles)
        assert util.fields_equal(ref[0], sc.primitives) and util.fields_equal(ref[1], sc.bvh_nodes)
        p, n, d, ms = st.build_bvh_device(sc.spheres, sc.plane_descs, sc.vertices, sc.triangles)
        assert same_tree((p, n, d), ref), name
        report.append("%s: %d primitives, %d nodes, depth %d, device %.2f ms" % (name.split("/")[-1], len(p), len(n), d, ms))
    rng = np.random.default_rng(7)  # the 25 random scenes of tests/test_host_preprocess.py::test_bvh_random_scenes_match_oracle
    for trial in range(25):
        ns, npl, nt = rng.integers(0, 12), rng.integers(0, 4), rng.integers(0, 60)
        if ns + npl + nt == 0:
            ns = 1
        sph = np.zeros(ns, T.SPHERE)
        sph["pos"] = rng.uniform(-5, 5, (ns, 3))
        sph["radius"] = rng.uniform(0.05, 1.5, ns)
        pls = np.zeros(npl, T.PLANE_DESC)
        pls["pos"] = rng.uniform(-5, 5, (npl, 3))
        pls["forward"] = rng.uniform(-3, 3, (npl, 3))
        pls["right"] = rng.uniform(-3, 3, (npl, 3))
        verts = np.zeros(max(3, nt), T.VEC3)
        verts["v"] = np.round(rng.uniform(-4, 4, (len(verts), 3)) * (2 if trial % 2 else 64)) / (2 if trial % 2 else 64)
        tri = np.zeros(nt, T.TRIANGLE)
        for k in ("vertex_0", "vertex_1", "vertex_2"):
            tri[k] = rng.integers(0, len(verts), nt)
        p, n, d, _ = st.build_bvh_device(sph, pls, verts, tri)
        assert same_tree((p, n, d), host.build_bvh(sph, pls, verts, tri)), trial
    # vertices on a symmetry plane written as 0.0 here and -0.0 there (OBJ exporters do that): min / max of (+0, -0) is either zero, by
    # operand order on the host (std::fmin / fmax, as the reference's f32::min / max) and always (-0, +0) on the device (monotone integer keys) —
    # the trees are the same node for node and the bounds the same VALUES; the sign bit of a zero bound may differ, and no traversal can tell
    # (the exception to "bit-identical", stated in rt_bvh_device.h)
    rng = np.random.default_rng(12)
    nt = 48
    verts = np.zeros(3 * nt, T.VEC3)
    v = np.round(rng.uniform(-2, 2, (3 * nt, 3)) * 2) / 2
    v[rng.random(3 * nt) < 0.5, 0] = 0.0
    v[:, 0] = np.where((v[:, 0] == 0) & (rng.random(3 * nt) < 0.5), -0.0, v[:, 0])
    verts["v"] = v
    assert np.signbit(verts["v"][:, 0][verts["v"][:, 0] == 0]).any() and not np.signbit(verts["v"][:, 0][verts["v"][:, 0] == 0]).all()
    tri = np.zeros(nt, T.TRIANGLE)
    tri["vertex_0"], tri["vertex_1"], tri["vertex_2"] = np.arange(nt) * 3, np.arange(nt) * 3 + 1, np.arange(nt) * 3 + 2
    p, n, d, _ = st.build_bvh_device(np.zeros(0, T.SPHERE), np.zeros(0, T.PLANE_DESC), verts, tri)
    assert same_tree((p, n, d), host.build_bvh(np.zeros(0, T.SPHERE), np.zeros(0, T.PLANE_DESC), verts, tri))  # (values: -0.0 == 0.0)
    # a big random triangle soup: long top-level ranges (many scan chunks per workgroup), deep tree
    rng = np.random.default_rng(2)
    nt = 40000
    verts = np.zeros(3 * nt, T.VEC3)
    c = rng.uniform(-20, 20, (nt, 1, 3))
    verts["v"] = (c + rng.normal(0, 0.3, (nt, 3, 3))).reshape(-1, 3)
    tri = np.zeros(nt, T.TRIANGLE)
    tri["vertex_0"], tri["vertex_1"], tri["vertex_2"] = np.arange(nt) * 3, np.arange(nt) * 3 + 1, np.arange(nt) * 3 + 2
    import time
    t = time.perf_counter()
    ref = host.build_bvh(np.zeros(0, T.SPHERE), np.zeros(0, T.PLANE_DESC), verts, tri)
    host_ms = (time.perf_counter() - t) * 1e3
    p, n, d, ms = st.build_bvh_device(np.zeros(0, T.SPHERE), np.zeros(0, T.PLANE_DESC), verts, tri)
    assert same_tree((p, n, d), ref)
    report.append("40,000 random triangles: %d nodes, depth %d, device %.2f ms, host %.1f ms" % (len(n), d, ms, host_ms))
    with pytest.raises(R.RsrtError, match="empty"):
        st.build_bvh_device(np.zeros(0, T.SPHERE), np.zeros(0, T.PLANE_DESC), np.zeros(0, T.VEC3), np.zeros(0, T.TRIANGLE))
    st.close()
    print("\n".join(report))
