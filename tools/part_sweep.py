"""Scratch: per-rank kernel time when the frame is partitioned over `world` ranks (one GPU plays each rank in turn)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import util
import rsoderh_raytracing_amd as R
env = R.Environment.synthetic(2048, 1024)
sc = R.Scene.load_toml(util.scene_path('house'))
st = R.State.new(sc, env, 1920, 1080); st.max_bounces = 8
for world in (1, 2, 4, 8):
    ts = []
    for rank in range(world if world <= 2 else 2):
        st.set_partition(rank, world)
        st.clear(); st.render_range(0, 256); st.synchronize(); st.stats()
        st.clear(); st.render_range(0, 256); st.synchronize(); g = st.stats()
        ts.append(g['kernel_ms'])
    print(f'world {world}: rank kernel ms {["%.1f" % t for t in ts]}  ideal {ts[0] if world == 1 else 0:.1f}', flush=True)
