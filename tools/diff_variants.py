import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, util
import rsoderh_raytracing_amd as R
env = R.Environment.synthetic(2048, 1024)
sc = R.Scene.load_toml(util.scene_path('house'))
imgs = {}
for v in '0', '2':
    os.environ['RSRT_KERNEL'] = v
    st = R.State.new(sc, env, 1920, 1080); st.max_bounces = 8
    st.render_range(0, 256); imgs[v] = st.download(); st.close()
bad = np.argwhere((util.bits(imgs['0']) != util.bits(imgs['2'])).any(axis=2))
print('differing pixels', len(bad))
for y, x in bad[:20]:
    print(x, y, imgs['0'][y, x], imgs['2'][y, x])
# narrow down the sample index for the first few
os.environ['RSRT_KERNEL'] = '0'; s0 = R.State.new(sc, env, 1920, 1080); s0.max_bounces = 8
os.environ['RSRT_KERNEL'] = '1'; s2 = R.State.new(sc, env, 1920, 1080); s2.max_bounces = 8
for y, x in bad[:5]:
    for k in range(256):
        s0.clear(); s0.render_range(k, 1); a = s0.download()[y, x]
        s2.clear(); s2.render_range(k, 1); b = s2.download()[y, x]
        if not np.array_equal(a.view(np.uint32), b.view(np.uint32)):
            print('pixel', x, y, 'sample', k, 'v0', a, 'v2', b)
