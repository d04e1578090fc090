"""Scratch: kernel time of small jobs (house 1080p, 1 / 4 / 16 / 64 spp per call, 50 calls each) for the library named by RSRT_LIB."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import util
import rsoderh_raytracing_amd as R
from rsoderh_raytracing_amd import state as S
env = R.Environment.synthetic(2048, 1024)
sc = R.Scene.load_toml(util.scene_path('house'))
st = R.State.new(sc, env, 1920, 1080); st.max_bounces = 8
out = []
for spp in (1, 4, 16, 64):
    n = 50 if spp < 64 else 10
    st.render_range(0, spp); st.synchronize(); st.stats()
    for i in range(n): st.render_range(0, spp)
    st.synchronize(); g = st.stats()
    out.append('%d spp %.3f ms' % (spp, g['trace_kernel_ms'] / n))
print(S.build_id(), ' | '.join(out), flush=True)
