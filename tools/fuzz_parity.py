"""Differential fuzz: random small scenes (spheres / planes / triangles on a coarse grid, with coincident,
degenerate and axis-aligned geometry, cameras that look exactly along an axis) rendered by the product and by
the oracle; every image must match bit for bit and the ray counts must agree.
Every 20th trial is a scene of thousands of small clustered triangles (a tree deeper than the wide walk's register stack), every other 20th a soup of
large overlapping triangles (long stacks).
python tools/fuzz_parity.py [trials] [first seed]      (run on the GPU box; ~0.25 s per trial)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import oracle, util
import rsoderh_raytracing_amd as R
from rsoderh_raytracing_amd import host, types as T

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
env = R.Environment.synthetic(128, 64)
oenv = util.oracle_env(env)
bad = 0
t0 = time.time()
for trial in range(trials):
    rng = np.random.default_rng(seed0 + trial)
    ns, npl = int(rng.integers(0, 6)), int(rng.integers(0, 4))
    nt = int(rng.integers(1, 64 - ns - npl)) if trial % 4 else int(rng.integers(60, 120))  # every 4th: too big for the flat loop
    if trial % 20 == 10:
        nt = int(rng.integers(350, 500))  # image > 24 KB: nodes + escape links in LDS, the rest in global memory
    deep = trial % 20 in (5, 15)  # trees deeper than the wide walk's register stack (kernel variant TRAV 5; see below)
    if deep:
        nt = int(rng.integers(3000, 7000)) if trial % 20 == 5 else int(rng.integers(1200, 2500))
    mats = np.zeros(4, T.MATERIAL)
    mats["color"] = rng.uniform(0.05, 1, (4, 3))
    mats["roughness"] = rng.choice([0.0, 0.05, 0.3, 1.0], 4)
    mats["metallic"] = rng.choice([0.0, 0.5, 1.0], 4)
    mats["emission"] = rng.choice([0.0, 0.0, 0.4], (4, 3))
    sph = np.zeros(ns, T.SPHERE)
    sph["pos"], sph["radius"], sph["material_id"] = np.round(rng.uniform(-3, 3, (ns, 3)) * 2) / 2, rng.choice([0.25, 0.5, 1.0], ns), rng.integers(0, 4, ns)
    pls = np.zeros(npl, T.PLANE_DESC)
    axis = np.eye(3)
    for k in range(npl):  # axis-aligned parallelograms: rays in them, and box faces on them, are common
        a, b = rng.choice(3, 2, replace=False)
        pls["pos"][k] = np.round(rng.uniform(-4, 0, 3))
        pls["forward"][k], pls["right"][k] = axis[a] * rng.choice([2.0, 4.0, 8.0]), axis[b] * rng.choice([2.0, 4.0, 8.0])
    pls["material_id"] = rng.integers(0, 4, npl)
    verts = np.zeros(3 * nt, T.VEC3)
    verts["v"] = np.round(rng.uniform(-3, 3, (3 * nt, 3)) * 2) / 2
    if deep and trial % 20 == 5:  # thousands of small triangles in a few dense clusters of very different sizes: a deep, lopsided tree
        centres = rng.uniform(-3, 3, (6, 3))
        which = rng.choice(6, nt, p=[0.5, 0.25, 0.12, 0.07, 0.04, 0.02])
        base = centres[which] + rng.normal(size=(nt, 3)) * rng.choice([0.05, 0.3, 1.0], 6)[which][:, None]
        verts["v"] = (np.repeat(base, 3, axis=0) + rng.uniform(-0.15, 0.15, (3 * nt, 3))).astype(np.float32)
    elif deep:  # a soup of large triangles: most boxes overlap, a ray meets most of the tree and its stack runs deep
        verts["v"] = (rng.uniform(-3, 3, (3 * nt, 3)) + np.repeat(rng.uniform(-0.5, 0.5, (nt, 3)), 3, axis=0)).astype(np.float32)
    for k in range(0, nt - 1, 5):  # coincident copies (ties) and zero-area triangles
        verts["v"][3 * k + 3:3 * k + 6] = verts["v"][3 * k:3 * k + 3]
    if nt > 3:
        verts["v"][6:9] = verts["v"][6]
    norms = np.zeros(3 * nt, T.VEC3)
    nn = rng.normal(size=(3 * nt, 3))
    norms["v"] = nn / np.linalg.norm(nn, axis=1, keepdims=True)
    tri = np.zeros(nt, T.TRIANGLE)
    tri["vertex_0"], tri["vertex_1"], tri["vertex_2"] = np.arange(nt) * 3, np.arange(nt) * 3 + 1, np.arange(nt) * 3 + 2
    tri["normal_0"], tri["normal_1"], tri["normal_2"] = tri["vertex_0"], tri["vertex_1"], tri["vertex_2"]
    tri["material_id"] = rng.integers(0, 4, nt)
    if trial % 3 == 0:  # looking exactly down -z from a grid point: direction components of exactly 0 at the image centre lines
        cam = host.make_camera_desc(np.round(rng.uniform(-1, 1, 3)) + [0, 0, 6], yaw=0.0, pitch=0.0, fov_y=1.0)
    else:
        cam = host.make_camera_desc(rng.uniform(-1, 1, 3) + [0, 1, 6], yaw=rng.uniform(-0.4, 0.4), pitch=rng.uniform(-0.3, 0.1), fov_y=1.2)
    sc = R.Scene(mats, sph, pls, verts, norms, tri, cam)
    w, h, spp, mb = 64, 40, 3, 5
    ref, ost = oracle.render(util.oracle_scene(sc), oenv, sc.camera_uniform().view(oracle.CAMERA), w, h, 0, spp, mb)
    # the product's choice (flat loop / wide walk) for every scene; every 5th also through the wide walk with small scenes forced
    # onto it and every ray parked after one round, the fixed-order walk and the two tree walks
    for cap in ('4', '4-noflat-b1', '3-noflat', '1', '0') if trial % 5 == 0 else ('4',):
        os.environ['RSRT_TRAVERSAL'] = cap[0]
        os.environ['RSRT_FLAT'] = '0' if 'noflat' in cap else '1'
        os.environ.pop('RSRT_TRACE_BUDGET', None)
        if cap.endswith('b1'): os.environ['RSRT_TRACE_BUDGET'] = '1'
        st = R.State.new(sc, env, w, h); st.max_bounces = mb
        st.render_range(0, spp); img = st.download(); g = st.stats(); st.close()
        ok = np.array_equal(util.bits(img), util.bits(ref)) and (g['ext_rays'], g['shadow_rays']) == (ost['ext_rays'], ost['shadow_rays'])
        if not ok:
            bad += 1
            print('MISMATCH seed %d traversal cap %s: %d pixels differ, rays %s vs %s' % (seed0 + trial, cap, int((util.bits(img) != util.bits(ref)).any(axis=2).sum()),
                  (g['ext_rays'], g['shadow_rays']), (ost['ext_rays'], ost['shadow_rays'])), flush=True)
    if (trial + 1) % 25 == 0:
        print('%d trials, %d mismatches, %.0f s' % (trial + 1, bad, time.time() - t0), flush=True)
print('fuzz_parity: %d trials (seeds %d..%d), %d mismatches' % (trials, seed0, seed0 + trials - 1, bad))
sys.exit(1 if bad else 0)
