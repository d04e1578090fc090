"""Scratch: pipelined single-sample calls (house 1920x1080, 10 bounces) under lane count / grid divisor knobs, interleaved rounds on one box.
    python tools/pipe_sweep.py "RSRT_PIPE_LANES=4;RSRT_PIPE_LANES=8,RSRT_PIPE_DIV=2" [calls]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import util
import rsoderh_raytracing_amd as R
configs = [dict(kv.split('=') for kv in c.split(',') if kv) for c in (sys.argv[1] if len(sys.argv) > 1 else '').split(';')]
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 16
env = R.Environment.synthetic(2048, 1024)
sc = R.Scene.load_toml(util.scene_path('house'))
states = []
for c in configs:
    for k in list(os.environ):
        if k.startswith('RSRT_') and k != 'RSRT_LIB': del os.environ[k]
    os.environ.update(c)
    st = R.State.new(sc, env, 1920, 1080); st.max_bounces = 10
    for i in range(24): st.render_range(i, 1)  # warm-up: clocks, every lane's buffers
    st.synchronize(); st.stats()
    states.append(st)
ref = None
for rnd in range(3):
    for c, st in zip(configs, states):
        st.clear(); st.synchronize()
        t = time.perf_counter()
        for i in range(calls): st.render_range(i, 1)
        st.synchronize(); dt = (time.perf_counter() - t) / calls
        g = st.stats()
        img = st.download().view(np.uint32)
        if ref is None: ref = img
        print('round %d %-40s %d calls: %.3f ms/call wall (sum of kernel times %.3f ms/call)  bits == first: %s' % (
            rnd, ','.join('%s=%s' % kv for kv in c.items()) or '(defaults)', calls, dt * 1e3, g['kernel_ms'] / calls, bool(np.array_equal(img, ref))), flush=True)
