"""Differential fuzz of the device BVH builder against the host builder: random mixes of spheres, planes and triangles, 1 .. ~60,000 primitives, on
coarse vertex grids (equal centroids, degenerate splits) and in clusters (long ranges at the top, lopsided trees); primitives, nodes and depth compared
field for field.    python tools/fuzz_bvh.py [trials] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import util
import rsoderh_raytracing_amd as R
from rsoderh_raytracing_amd import host, types as T
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
st = R.State.new(R.Scene.load_toml(util.scene_path('default')), util.small_env(), 16, 16)
bad = 0
for t in range(trials):
    rng = np.random.default_rng(seed0 + t)
    big = t % 10 == 9
    nt = int(rng.integers(20000, 60000)) if big else int(rng.integers(0, 3000) if t % 3 else rng.integers(0, 40))
    ns, npl = int(rng.integers(0, 30)), int(rng.integers(0, 6))
    if ns + npl + nt == 0: ns = 1
    sph = np.zeros(ns, T.SPHERE); sph['pos'] = rng.uniform(-5, 5, (ns, 3)); sph['radius'] = rng.uniform(0.05, 1.5, ns)
    pls = np.zeros(npl, T.PLANE_DESC); pls['pos'] = rng.uniform(-5, 5, (npl, 3)); pls['forward'] = rng.uniform(-3, 3, (npl, 3)); pls['right'] = rng.uniform(-3, 3, (npl, 3))
    nv = max(3, 3 * nt if t % 2 else nt // 2 + 3)
    verts = np.zeros(nv, T.VEC3)
    if t % 4 == 0:  # coarse grid: many equal coordinates
        verts['v'] = np.round(rng.uniform(-4, 4, (nv, 3)) * 2) / 2
    elif t % 4 == 1:  # clusters of very different sizes
        c = rng.uniform(-20, 20, (8, 3)); which = rng.choice(8, nv, p=[0.6, 0.2, 0.1, 0.05, 0.02, 0.01, 0.01, 0.01])
        verts['v'] = c[which] + rng.normal(size=(nv, 3)) * rng.choice([0.01, 0.5, 3.0], 8)[which][:, None]
    else:
        verts['v'] = rng.uniform(-8, 8, (nv, 3))
    tri = np.zeros(nt, T.TRIANGLE)
    if t % 2:
        tri['vertex_0'], tri['vertex_1'], tri['vertex_2'] = np.arange(nt) * 3, np.arange(nt) * 3 + 1, np.arange(nt) * 3 + 2
    else:
        for k in ('vertex_0', 'vertex_1', 'vertex_2'): tri[k] = rng.integers(0, nv, nt)
    p, n, d, ms = st.build_bvh_device(sph, pls, verts, tri)
    rp, rn, rd = host.build_bvh(sph, pls, verts, tri)
    if not (util.fields_equal(p, rp) and util.fields_equal(n, rn) and d == rd):
        bad += 1
        print('MISMATCH seed %d: %d spheres %d planes %d triangles' % (seed0 + t, ns, npl, nt), flush=True)
    if (t + 1) % 50 == 0: print('%d trials, %d mismatches' % (t + 1, bad), flush=True)
print('fuzz_bvh: %d trials (seeds %d..%d), %d mismatches' % (trials, seed0, seed0 + trials - 1, bad))
st.close()
sys.exit(1 if bad else 0)
