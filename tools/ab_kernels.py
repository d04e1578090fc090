"""Scratch: parity + timing of the kernel variants (RSRT_KERNEL=0..2) in one process, interleaved rounds."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import oracle, util
import rsoderh_raytracing_amd as R

variants = [int(v) for v in (sys.argv[1].split(',') if len(sys.argv) > 1 else '0,1,2'.split(','))]
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
env_small = R.Environment.synthetic(256, 128)
for name, W, H, s, mb in [('house', 160, 90, 8, 8), ('default', 96, 64, 4, 10), ('suzanne', 96, 64, 4, 10)]:
    sc = R.Scene.load_toml(util.scene_path(name))
    ref, ost = oracle.render(util.oracle_scene(sc), util.oracle_env(env_small), sc.camera_uniform().view(oracle.CAMERA), W, H, 0, s, mb)
    for v in variants:
        os.environ['RSRT_KERNEL'] = str(v)
        st = R.State.new(sc, env_small, W, H); st.max_bounces = mb
        for flags in (0, 1):
            st.flags = flags; st.clear(); st.render_range(0, s)
            img = st.download(); g = st.stats()
            ok = np.array_equal(util.bits(img), util.bits(ref)) and (g['ext_rays'], g['shadow_rays']) == (ost['ext_rays'], ost['shadow_rays'])
            print(f'{name:8s} kernel {v} flags {flags}: bit-exact {ok}  {g["trace_kernel_ms"]:.2f} ms', flush=True)
        st.close()

env = R.Environment.synthetic(2048, 1024)
sc = R.Scene.load_toml(util.scene_path('house'))
states = {}
for v in variants:
    os.environ['RSRT_KERNEL'] = str(v)
    st = R.State.new(sc, env, 1920, 1080); st.max_bounces = 8
    states[v] = st
for rnd in range(3):
    for v in variants:
        st = states[v]
        st.clear(); st.render_range(0, spp); st.synchronize()
        g = st.stats(); rays = g['ext_rays'] + g['shadow_rays']
        print(f'round {rnd} kernel {v}: house 1080p {spp}spp trace {g["trace_kernel_ms"]:.1f} ms  {rays/g["trace_kernel_ms"]/1e3:.0f} Mrays/s', flush=True)
