"""Diagnostic: where SHADE's wave time goes (instrumented build compiled with -DRT_SHADE_PROFILE: s_memtime stamps between the
sections of the stage, lane 0 of every wave).  Build here: RSRT_HIPCC_FLAGS=-DRT_SHADE_PROFILE python -c "from
rsoderh_raytracing_amd import _build; print(_build.build_hip(instrument=True))", then on the GPU box
    RSRT_LIB=<that library> python tools/shade_profile.py [spp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import util
import rsoderh_raytracing_amd as R
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
env = R.Environment.synthetic(2048, 1024)
sc = R.Scene.load_toml(util.scene_path('house'))
st = R.State.new(sc, env, 1920, 1080); st.max_bounces = 8
st.render_range(0, spp); st.synchronize()
g = st.stats(); c = st.debug_counters().astype(np.float64)
inv = max(c[3], 1)
parts = [('hot reads, alias gather and cold loads issued', c[10]), ('hit geometry + material (LDS)', c[11]), ('environment sample finished (alias entry, texels)', c[12]),
         ('cold loads consumed, frame, NEE term', c[13]), ('BSDF sample', c[14]), ('throughput, termination, stores', c[19])]
tot = sum(v for _, v in parts)
print('SHADE: %.0f invocations, %.0f cycles each (s_memtime ticks of lane 0)' % (c[3], tot / inv))
for n, v in parts:
    print('  %-52s %7.0f cycles  %5.1f%%' % (n, v / inv, 100 * v / tot))
