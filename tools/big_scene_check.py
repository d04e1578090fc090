"""Scale check of the general-BVH path: suzanne instanced n x n (8 x 8 = 61,952 triangles, 16 x 16 = 247,808) through the normal loader — host
and device BVH builders node for node, a reduced frame bit for bit against the oracle, which traversal and how much of it in LDS, and the rate.
    python tools/big_scene_check.py [n ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np
import make_big_scene, oracle, util
import rsoderh_raytracing_amd as R
ns = [int(a) for a in sys.argv[1:]] or [8, 16]
env = R.Environment.synthetic(2048, 1024)
small_env = R.Environment.synthetic(256, 128)
for n in ns:
    t = time.time()
    sc = R.Scene.load_toml(make_big_scene.make(n))
    t_load = time.time() - t
    nodes = np.asarray(sc.bvh_nodes)
    print('grid %d x %d: %d triangles, %d BVH nodes (loader + host builder %.2f s)' % (n, n, len(sc.triangles), len(nodes), t_load), flush=True)
    # parity, reduced frame
    st = R.State.new(sc, small_env, 240, 135); st.max_bounces = 10
    t = time.time()
    p, nd, depth, ms = st.build_bvh_device(sc.spheres, sc.plane_descs, sc.vertices, sc.triangles)
    print('  device builder: %.1f ms on the device (%.2f s with transfers), depth %d; same tree as the host builder: %s' % (
        ms, time.time() - t, depth, bool(len(nd) == len(nodes) and util.fields_equal(nd, nodes) and util.fields_equal(p, np.asarray(sc.primitives)))), flush=True)
    st.render_range(0, 2); img, g = st.download(), st.stats(); st.close()
    t = time.time()
    ref, ost = oracle.render(util.oracle_scene(sc), util.oracle_env(small_env), sc.camera_uniform().view(oracle.CAMERA), 240, 135, 0, 2, 10, fast=True)
    ok = bool(np.array_equal(util.bits(img), util.bits(ref))) and (g['ext_rays'], g['shadow_rays']) == (ost['ext_rays'], ost['shadow_rays'])
    print('  240x135 x 2 spp vs oracle (%.1f s): bit-exact %s, rays %d, traversal steps / ray %.1f' % (time.time() - t, ok, g['ext_rays'] + g['shadow_rays'], g['traversal_steps'] / max(1, g['ext_rays'] + g['shadow_rays'])), flush=True)
    # rate
    for trav in ('6', '5', '3'):
        os.environ['RSRT_TRAVERSAL'] = trav
        st = R.State.new(sc, env, 1280, 720); st.max_bounces = 10
        st.render_range(0, 8); st.synchronize(); st.stats()
        for rnd in range(2):
            st.clear(); st.render_range(0, 8); st.synchronize(); g = st.stats()
        rays = g['ext_rays'] + g['shadow_rays']
        print('  1280x720 x 8 spp, RSRT_TRAVERSAL=%s: trace kernel %.2f ms, %.0f Mrays/s, %.1f traversal steps / ray' % (trav, g['trace_kernel_ms'], rays / g['trace_kernel_ms'] / 1e3, g['traversal_steps'] / rays), flush=True)
        st.close()
    del os.environ['RSRT_TRAVERSAL']
