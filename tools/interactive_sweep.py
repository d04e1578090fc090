import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import util
import rsoderh_raytracing_amd as R
env = R.Environment.synthetic(2048, 1024)
sc = R.Scene.load_toml(util.scene_path('house'))
for kv, b in [('4', '1'), ('2', '4'), ('3', '4'), ('2', '3'), ('2', '2'), ('4', '1'), ('2', '4')]:
    os.environ['RSRT_BLOCKS_PER_CU'] = b; os.environ['RSRT_KERNEL'] = kv
    st = R.State.new(sc, env, 1920, 1080); st.max_bounces = 10
    for spp in (1, 2, 4, 16):
        st.render_range(0, spp); st.synchronize(); st.stats()
        t = time.perf_counter()
        for i in range(50): st.render_range(0, spp)
        st.synchronize(); dt = (time.perf_counter() - t) / 50
        g = st.stats()
        print(f'kernel {kv} blocks/CU {b} spp/call {spp}: wall {dt*1e3:.2f} ms  kernel {g["kernel_ms"]/50:.2f} ms (trace {g["trace_kernel_ms"]/50:.2f}, resolve {g["resolve_kernel_ms"]/50:.3f})', flush=True)
    st.close()
