import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, util, oracle
import rsoderh_raytracing_amd as R
env = R.Environment.synthetic(2048, 1024)
sc = R.Scene.load_toml(util.scene_path('house'))
osc, oenv, cam = util.oracle_scene(sc), util.oracle_env(env), sc.camera_uniform().view(oracle.CAMERA)
for v in '0', '2':
    os.environ['RSRT_KERNEL'] = v
    st = R.State.new(sc, env, 1920, 1080)
    for (x, y, k) in [(851, 477, 149), (1437, 1068, 240)]:
        for mb in (1, 2, 3, 4, 5, 6, 7, 8):
            row = []
            for flags in (0, 1):
                st.flags = flags; st.max_bounces = mb
                st.clear(); st.render_range(k, 1); a = st.download()[y, x]; g = st.stats()
                row.append(tuple(a[:3]))
            print('kernel', v, (x, y, k), 'bounces', mb, 'flags0', row[0], 'flags1', row[1], 'same' if row[0] == row[1] else 'DIFF')
    st.close()
