"""Scratch: time vs resident workgroups per CU for kernel variants.  python tools/occ_sweep.py "1:1,2,3,4 2:1,2,3,4" [spp]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import util
import rsoderh_raytracing_amd as R
spec = sys.argv[1] if len(sys.argv) > 1 else "0:3 1:3,4 2:2,3,4 3:2,3"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
env = R.Environment.synthetic(2048, 1024)
sc = R.Scene.load_toml(util.scene_path('house'))
for item in spec.split():
    kv, bl = item.split(':')
    for b in bl.split(','):
        os.environ['RSRT_KERNEL'] = kv; os.environ['RSRT_BLOCKS_PER_CU'] = b
        st = R.State.new(sc, env, 1920, 1080); st.max_bounces = 8
        st.render_range(0, spp); st.synchronize(); st.stats()
        st.clear(); st.render_range(0, spp); st.synchronize(); g = st.stats()
        rays = g['ext_rays'] + g['shadow_rays']
        print(f'kernel {kv} blocks/CU {b}: {g["trace_kernel_ms"]:.1f} ms  {rays/g["trace_kernel_ms"]/1e3:.0f} Mrays/s', flush=True)
        st.close()
