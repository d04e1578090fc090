"""The reference's interactive mode (State::render, src/state.rs:760-833: ONE sample per frame) on the BASELINE scene: wall time per
rsrt_render call over 16 back-to-back calls, with the two-lane overlap of consecutive calls (default) and without (RSRT_OVERLAP=0),
interleaved rounds on one box; the accumulators of both are compared bit for bit.
    python tools/interactive_ab.py [spp_per_call] [calls]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import util
import rsoderh_raytracing_amd as R
from rsoderh_raytracing_amd import state as S
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 1
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 16
env = R.Environment.synthetic(2048, 1024)
sc = R.Scene.load_toml(util.scene_path('house'))
print('library', S.build_id(), flush=True)
states = {}
for ov in ('1', '0'):
    os.environ['RSRT_OVERLAP'] = ov
    st = R.State.new(sc, env, 1920, 1080); st.max_bounces = 10
    for i in range(20): st.render_range(i * spp, spp)  # warm-up: clocks, both lanes' buffers
    st.synchronize(); st.stats()
    states[ov] = st
imgs = {}
for rnd in range(4):
    for ov, st in states.items():
        st.clear(); st.synchronize()
        t = time.perf_counter()
        for i in range(calls): st.render_range(i * spp, spp)
        st.synchronize(); dt = (time.perf_counter() - t) / calls
        g = st.stats(); rays = (g['ext_rays'] + g['shadow_rays']) / calls
        print('round %d overlap %s: %d spp/call x %d calls: %.3f ms/call wall  %.0f Mrays/s  (sum of kernel times %.3f ms/call)' % (rnd, ov, spp, calls, dt * 1e3, rays / dt / 1e6, g['kernel_ms'] / calls), flush=True)
        if rnd == 0: imgs[ov] = st.download()
print('accumulators identical:', bool(np.array_equal(imgs['1'].view(np.uint32), imgs['0'].view(np.uint32))))
bulk = states['1']; bulk.clear(); bulk.synchronize()
t = time.perf_counter(); bulk.render_range(0, spp * calls); bulk.synchronize(); dt = time.perf_counter() - t
print('one call of %d spp: %.3f ms per %d spp; == the %d calls: %s' % (spp * calls, dt * 1e3 / calls, spp, calls, bool(np.array_equal(bulk.download().view(np.uint32), imgs['1'].view(np.uint32)))))
