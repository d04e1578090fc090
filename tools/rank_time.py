"""Scratch: kernel time of rank 0's share of the BASELINE frame at world 8 (and of the whole frame) under the current knobs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import util
import rsoderh_raytracing_amd as R
env = R.Environment.synthetic(2048, 1024)
sc = R.Scene.load_toml(util.scene_path('house'))
st = R.State.new(sc, env, 1920, 1080); st.max_bounces = 8
out = []
for world, rank in ((8, 0), (8, 3), (1, 0)):
    st.set_partition(rank, world)
    st.clear(); st.render_range(0, 256); st.synchronize(); st.stats()
    ts = []
    for _ in range(3):
        st.clear(); st.render_range(0, 256); st.synchronize(); ts.append(st.stats()['trace_kernel_ms'])
    out.append('world %d rank %d: %.2f ms' % (world, rank, min(ts)))
print(' '.join('%s=%s' % kv for kv in sorted(os.environ.items()) if kv[0].startswith('RSRT_')), '|', ' | '.join(out), flush=True)
