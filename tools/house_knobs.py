"""Scratch: run-time knob matrix on the BASELINE scene (house 1920x1080, 8 bounces, 64 spp per measurement) for the
library named by RSRT_LIB (or the product); every configuration is checked against the first one bit for bit.
    python tools/house_knobs.py "RSRT_COOP_LANES=0;RSRT_COOP_LANES=24;RSRT_COOP_LANES=32" [spp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import util
import rsoderh_raytracing_amd as R
from rsoderh_raytracing_amd import state as S
configs = [dict(kv.split('=') for kv in c.split(',') if kv) for c in (sys.argv[1] if len(sys.argv) > 1 else '').split(';')]
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
env = R.Environment.synthetic(2048, 1024)
sc = R.Scene.load_toml(util.scene_path('house'))
print('library', S.build_id(), flush=True)
states = []
for c in configs:
    for k in list(os.environ):
        if k.startswith('RSRT_') and k not in ('RSRT_LIB',):
            del os.environ[k]
    os.environ.update(c)
    st = R.State.new(sc, env, 1920, 1080); st.max_bounces = 8
    st.render_range(0, spp); st.synchronize(); st.stats()
    states.append(st)
ref = None
for rnd in range(3):
    for c, st in zip(configs, states):
        st.clear(); st.render_range(0, spp); st.synchronize()
        g = st.stats(); rays = g['ext_rays'] + g['shadow_rays']
        same = ''
        if rnd == 0:
            img = st.download()
            if ref is None: ref = img
            same = '  image == first: %s' % bool(np.array_equal(ref.view(np.uint32), img.view(np.uint32)))
        print('round %d %-44s trace %7.2f ms  %6.0f Mrays/s%s' % (rnd, ','.join('%s=%s' % kv for kv in c.items()) or '(defaults)', g['trace_kernel_ms'], rays / g['trace_kernel_ms'] / 1e3, same), flush=True)
