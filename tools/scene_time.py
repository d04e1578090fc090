"""Scratch: time one scene/config under the current environment knobs.  python tools/scene_time.py cube 1280 720 128 10
(scene: a name under tests/golden/scenes, a .toml path, or gridN = suzanne instanced N x N by tools/make_big_scene.py)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import util
import rsoderh_raytracing_amd as R
scene, w, h, spp, mb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
env = R.Environment.synthetic(2048, 1024)
if scene.startswith('grid'):
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import make_big_scene
    path = make_big_scene.make(int(scene[4:] or 4))
else:
    path = scene if scene.endswith('.toml') else util.scene_path(scene)
sc = R.Scene.load_toml(path)
st = R.State.new(sc, env, w, h); st.max_bounces = mb
for i in range(3):
    st.clear(); st.render_range(0, spp); st.synchronize()
    g = st.stats(); rays = g['ext_rays'] + g['shadow_rays']
    print('%s %s: trace %.2f ms  %.0f Mrays/s  %.2f traversal steps/ray' % (scene, ' '.join('%s=%s' % (k, v) for k, v in os.environ.items() if k.startswith('RSRT_')), g['trace_kernel_ms'], rays / g['trace_kernel_ms'] / 1e3, g['traversal_steps'] / max(1, rays)), flush=True)
