import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, oracle, util
import rsoderh_raytracing_amd as R
env = R.Environment.synthetic(2048, 1024)
sc = R.Scene.load_toml(util.scene_path('house'))
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ref, ost = oracle.render(util.oracle_scene(sc), util.oracle_env(env), sc.camera_uniform().view(oracle.CAMERA), 1920, 1080, 0, spp, 8, fast=True)
print('oracle', ost['paths'], ost['ext_rays'], ost['shadow_rays'])
for v in '1', '2':
    os.environ['RSRT_KERNEL'] = v
    st = R.State.new(sc, env, 1920, 1080); st.max_bounces = 8
    st.render_range(0, spp); img = st.download(); g = st.stats()
    print('kernel', v, g['paths'], g['ext_rays'], g['shadow_rays'], 'bit-exact', np.array_equal(util.bits(img), util.bits(ref)))
    st.render_range(spp, spp); st.synchronize(); g = st.stats(); print('   second call', g['paths'], g['ext_rays'], g['shadow_rays'])
    st.close()
