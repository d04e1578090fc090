"""Scratch: first GPU bring-up — parity vs oracle on small frames, then a timing of the bench config."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import oracle, util
import rsoderh_raytracing_amd as R

env = R.Environment.synthetic(256, 128)
oenv = util.oracle_env(env)
for name, W, H, spp, mb in [('default', 64, 48, 4, 3), ('house', 160, 90, 8, 8), ('cube', 96, 64, 4, 10), ('suzanne', 96, 64, 4, 10), ('spheres_only', 64, 48, 4, 10)]:
    sc = R.Scene.load_toml(util.scene_path(name))
    osc = util.oracle_scene(sc)
    cam = sc.camera_uniform()
    ref, ost = oracle.render(osc, oenv, cam.view(oracle.CAMERA), W, H, 0, spp, mb)
    st = R.State.new(sc, env, W, H)
    st.max_bounces = mb
    for flags in (0, 1):
        st.flags = flags
        st.clear(); st._last_hash = st._scene_hash()
        st.render_range(0, spp)
        img = st.download()
        s = st.stats()
        nbad = int((util.bits(img) != util.bits(ref)).any(axis=2).sum())
        print(f'{name:13s} {W}x{H} spp{spp} mb{mb} flags{flags}: mismatching pixels {nbad}/{W*H}  rmse {util.rmse_per_channel(img, ref, spp)}  rays gpu {s["ext_rays"]}+{s["shadow_rays"]} oracle {ost["ext_rays"]}+{ost["shadow_rays"]}  ms {s["kernel_ms"]:.2f}', flush=True)
    # ray probe
    rng = np.random.default_rng(1)
    o = np.tile(np.float32(sc.camera_desc['pos'][0]), (1000, 1)); d = rng.normal(size=(1000, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    hg = st.cast_rays(o, d, 0, 1); ho = oracle.cast_rays(osc, o, d, 0, 0)
    print('   cast_rays identical:', util.fields_equal(hg, ho), 'hits', int(hg['did_hit'].sum()), flush=True)
    st.close()

# benchmark-shaped timing
env = R.Environment.synthetic(2048, 1024)
sc = R.Scene.load_toml(util.scene_path('house'))
st = R.State.new(sc, env, 1920, 1080)
st.max_bounces = 8
print(st.describe())
for spp in (4, 16, 64):
    st.clear(); st._last_hash = st._scene_hash()
    t = time.time(); st.render_range(0, spp); st.synchronize(); dt = time.time() - t
    s = st.stats()
    rays = s['ext_rays'] + s['shadow_rays']
    print(f'house 1080p spp{spp}: wall {dt*1e3:.1f} ms kernel {s["kernel_ms"]:.1f} ms  rays {rays/1e6:.1f}M  {rays/s["kernel_ms"]/1e3:.1f} Mrays/s  launches {s["launches"]}', flush=True)
