"""Scratch: time the house 1080p frame for values of one environment knob.
python tools/knob_sweep.py RSRT_TRACE_BUDGET 8,12,16,1000 [kernel variant] [spp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
knob, values = sys.argv[1], sys.argv[2].split(',')
if len(sys.argv) > 3: os.environ['RSRT_KERNEL'] = sys.argv[3]
spp = int(sys.argv[4]) if len(sys.argv) > 4 else 64
import util
import rsoderh_raytracing_amd as R
env = R.Environment.synthetic(2048, 1024)
sc = R.Scene.load_toml(util.scene_path('house'))
states = {}
for v in values:
    os.environ[knob] = v
    st = R.State.new(sc, env, 1920, 1080); st.max_bounces = 8
    states[v] = st
for rnd in range(3):
    for v in values:
        st = states[v]
        st.clear(); st.render_range(0, spp); st.synchronize()
        g = st.stats(); rays = g['ext_rays'] + g['shadow_rays']
        print(f'round {rnd} {knob}={v}: trace {g["trace_kernel_ms"]:.1f} ms  {rays/g["trace_kernel_ms"]/1e3:.0f} Mrays/s', flush=True)
