"""Scratch: run-time knob matrix on the general-BVH scenes (suzanne, suzanne grid) for the library named by RSRT_LIB
(or the product).  python tools/bvh_knobs.py "RSRT_TRACE_BUDGET=6;RSRT_TRACE_BUDGET=12;RSRT_KERNEL=1,RSRT_TRACE_BUDGET=24" """
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import util, make_big_scene
import rsoderh_raytracing_amd as R
from rsoderh_raytracing_amd import state as S
configs = [dict(kv.split('=') for kv in c.split(',') if kv) for c in (sys.argv[1] if len(sys.argv) > 1 else '').split(';')]
env = R.Environment.synthetic(2048, 1024)
scenes = [('grid', R.Scene.load_toml(make_big_scene.make(4)), 1280, 720, 16, 10), ('suzanne', R.Scene.load_toml(util.scene_path('suzanne')), 1280, 720, 64, 10)]
print('library', S.build_id(), flush=True)
for name, sc, w, h, spp, mb in scenes:
    states = []
    for c in configs:
        for k in list(os.environ):
            if k.startswith('RSRT_') and k not in ('RSRT_LIB',):
                del os.environ[k]
        os.environ.update(c)
        st = R.State.new(sc, env, w, h); st.max_bounces = mb
        st.render_range(0, spp); st.synchronize(); st.stats()  # warm-up (allocations, occupancy query under this env)
        states.append(st)
    for rnd in range(2):
        for c, st in zip(configs, states):
            st.clear(); st.render_range(0, spp); st.synchronize()
            g = st.stats(); rays = g['ext_rays'] + g['shadow_rays']
            print('%-8s round %d %-44s trace %7.2f ms  %6.0f Mrays/s' % (name, rnd, ','.join('%s=%s' % kv for kv in c.items()) or '(defaults)', g['trace_kernel_ms'], rays / g['trace_kernel_ms'] / 1e3), flush=True)
    for st in states:
        st.close()
