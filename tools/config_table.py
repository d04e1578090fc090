"""BASELINE.json configs 2-4 (+ the builder-authored stress scenes) on one GPU: ms/frame, Mrays/s, and a
bit-exactness spot check against the oracle on the first samples of a 1/8-scale frame."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np, oracle, util, make_big_scene
import rsoderh_raytracing_amd as R
GRID = make_big_scene.make(4)
env = R.Environment.synthetic(2048, 1024)
oenv = util.oracle_env(env)
rows = []
for label, scene, w, h, spp, mb in [
        ('config 2: default.toml 1280x720 64spp (as shipped, BVH)', 'default', 1280, 720, 64, 10),
        ('config 2b: spheres_only.toml 1280x720 64spp', 'spheres_only', 1280, 720, 64, 10),
        ('config 3: cube.toml 1280x720 128spp', 'cube', 1280, 720, 128, 10),
        ('config 3b: suzanne.toml 1280x720 128spp (968 tris, 549 nodes; wide walk, all nodes in LDS)', 'suzanne', 1280, 720, 128, 10),
        ('config 3c: suzanne grid 4x4 1280x720 32spp (15,488 tris, 8,731 nodes; wide walk, top 192 nodes in LDS)', GRID, 1280, 720, 32, 10),
        ('config 4: house.toml 1920x1080 256spp 8 bounces', 'house', 1920, 1080, 256, 8),
        ('config 4b: house.toml 1920x1080 256spp 10 bounces (the reference constant)', 'house', 1920, 1080, 256, 10),
        ('interactive: house.toml 1920x1080, 1 spp per call (State::render)', 'house', 1920, 1080, 1, 10)]:
    sc = R.Scene.load_toml(scene if scene.endswith('.toml') else util.scene_path(scene))
    st = R.State.new(sc, env, w, h); st.max_bounces = mb
    for _ in range(1 if spp > 1 else 20):  # warm-up (clocks, allocations)
        st.render_range(0, spp)
    st.synchronize(); st.stats()
    n = 3 if spp > 1 else 50
    t = time.perf_counter()
    for i in range(n):
        st.render_range(0, spp)
    st.synchronize(); dt = (time.perf_counter() - t) / n
    g = st.stats(); rays = (g['ext_rays'] + g['shadow_rays']) / n
    # parity spot check: 1/8-scale frame, 4 spp
    sw, sh = max(16, w // 8), max(16, h // 8)
    st.resize(sw, sh); st.render_range(0, 4); img = st.download()
    ref, _ = oracle.render(util.oracle_scene(sc), oenv, sc.camera_uniform().view(oracle.CAMERA), sw, sh, 0, 4, mb, fast=True)
    ok = bool(np.array_equal(util.bits(img), util.bits(ref)))
    st.close()
    rows.append(dict(config=label, ms_per_frame=dt * 1e3, mrays_per_s=rays / dt / 1e6, rays_per_frame=rays, bit_exact_spot_check=ok))
    print('%-88s %9.2f ms/frame %9.0f Mrays/s  %5.2f rays/path  bit-exact %s' % (label, dt * 1e3, rays / dt / 1e6, rays / (w * h * spp), ok), flush=True)
json.dump(rows, open(os.path.join(ROOT, 'gpurun_out', 'config_table.json'), 'w'), indent=1)
