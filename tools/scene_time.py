"""Scratch: time one scene/config under the current environment knobs.  python tools/scene_time.py cube 1280 720 128 10"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import util
import rsoderh_raytracing_amd as R
scene, w, h, spp, mb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
env = R.Environment.synthetic(2048, 1024)
sc = R.Scene.load_toml(util.scene_path(scene))
st = R.State.new(sc, env, w, h); st.max_bounces = mb
for i in range(3):
    st.clear(); st.render_range(0, spp); st.synchronize()
    g = st.stats(); rays = g['ext_rays'] + g['shadow_rays']
    print('%s %s: trace %.2f ms  %.0f Mrays/s' % (scene, ' '.join('%s=%s' % (k, v) for k, v in os.environ.items() if k.startswith('RSRT_')), g['trace_kernel_ms'], rays / g['trace_kernel_ms'] / 1e3), flush=True)
