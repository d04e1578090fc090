"""Diagnostic: SIMD efficiency per code region of the wave-pool kernel, from the loop-trip counters of
the instrumented build (RSRT_INSTRUMENT=1).  python tools/simd_efficiency.py [kernel_variant] [spp] [scene name | path.toml] [w] [h] [bounces]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ['RSRT_INSTRUMENT'] = '1'
os.environ['RSRT_KERNEL'] = sys.argv[1] if len(sys.argv) > 1 else '4'
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
import numpy as np
import util
import rsoderh_raytracing_amd as R
env = R.Environment.synthetic(2048, 1024)
scene = sys.argv[3] if len(sys.argv) > 3 else 'house'
sc = R.Scene.load_toml(scene if scene.endswith('.toml') else util.scene_path(scene))
w, h, mb = (int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (1920, 1080, 8)
st = R.State.new(sc, env, w, h); st.max_bounces = mb
st.region_counters()  # (reset)
st.render_range(0, spp); st.synchronize()
g = st.stats(); c = st.debug_counters().astype(np.float64)
regions = [float(x) for x in st.region_counters()]
names = ['GEN', 'TRACE', 'MISS', 'SHADE', 'FINISH']
print(os.path.basename(scene), ' '.join('%s=%s' % kv for kv in sorted(os.environ.items()) if kv[0].startswith('RSRT_')), 'trace kernel %.1f ms' % g['trace_kernel_ms'], 'rays', g['ext_rays'] + g['shadow_rays'],
      'traversal steps/ray %.1f' % (g['traversal_steps'] / (g['ext_rays'] + g['shadow_rays'])))
for i, n in enumerate(names):
    if c[i]:
        print('  stage %-5s invocations %12.0f  avg lanes %.1f / 64' % (n, c[i], c[5 + i] / c[i]))
rays = g['ext_rays'] + g['shadow_rays']
print('  trace: outer trips/wave-invocation %.2f' % (c[14] / max(c[1], 1)))
print('  descend loop: wave trips %.3e, lane trips %.3e -> efficiency %.1f%% of 64 lanes ; nodes/ray %.2f' % (c[10], c[11], 100 * c[11] / max(64 * c[10], 1), c[11] / rays))
print('  leaf loop   : wave trips %.3e, lane trips %.3e -> efficiency %.1f%% ; prim tests/ray %.2f' % (c[12], c[13], 100 * c[13] / max(64 * c[12], 1), c[13] / rays))
print('  per TRACE invocation: descend wave trips %.1f, leaf wave trips %.1f' % (c[10] / max(c[1], 1), c[12] / max(c[1], 1)))
tot = sum(c[16:23])
if tot:  # also as a file: bench.py attaches it to the roofline object's `overhead` (what the non-algorithmic share consists of)
    import json
    from rsoderh_raytracing_amd import state as S
    shares = {n: c[16 + i] / tot for i, n in enumerate(names)}
    shares.update(census=c[22] / tot, other=c[21] / tot)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'stage_shares_%s.json' % os.path.basename(scene).replace('.toml', '')), 'w') as f:
        json.dump({'workload': '%s %dx%d %d spp %d bounces' % (os.path.basename(scene), w, h, spp, mb), 'build_id': S.build_id(),
                   'wave_time_shares': shares, 'stage_lanes_of_64': {n: (c[5 + i] / c[i] if c[i] else None) for i, n in enumerate(names)},
                   'counters': [float(x) for x in c], 'region_lanes': regions, 'rays': float(rays), 'paths': float(g['paths']), 'n_spheres': int(len(sc.spheres)), 'n_planes': int(len(sc.planes)),
                   'source': 'tools/simd_efficiency.py: instrumented build (librsrt_instr.so), s_memtime stamps of lane 0 between the stages'}, f, indent=1)
if tot:
    print('  wave-time shares (s_memtime, lane 0): ' + '  '.join('%s %.1f%%' % (n, 100 * c[16 + i] / tot) for i, n in enumerate(names)) +
          '  census %.1f%%  other %.1f%%' % (100 * c[22] / tot, 100 * c[21] / tot))
    print('  cycles per invocation: ' + '  '.join('%s %.0f' % (n, c[16 + i] / max(c[i], 1)) for i, n in enumerate(names)) + '  census %.0f' % (c[22] / max(sum(c[0:5]), 1)))
if c[14] > 0.5 * c[1] and c[25]:  # the cooperative walk (rt_coop.h): a TRACE call takes every waiting ray
    inv = max(c[1], 1)
    print('  cooperative walk per TRACE call: %.1f rays (slots), %.1f node trips at %.1f%% of the lanes, %.1f leaf trips popping %.1f items each, %.1f record passes (64 records each) at %.1f%% of the lanes' % (
        c[6] / inv, c[10] / inv, 100 * c[11] / max(64 * c[10], 1), c[14] / inv, c[28] / max(c[14], 1), c[12] / inv, 100 * c[13] / max(64 * c[12], 1)))
    print('    items popped: node %.1f%% of the lane slots (%.1f%% of them dropped: shadow rays already occluded), leaf %.1f%% (%.1f%% dropped)' % (
        100 * c[30] / max(64 * c[10], 1), 100 * (1 - c[11] / max(c[30], 1)), 100 * c[29] / max(64 * c[14], 1), 100 * (1 - c[28] / max(c[29], 1))))
    print('    rays handed to the exact fixed-order walk (equal closest t, or a non-finite 1/d): %d of %d (%.4f %%)' % (regions[4], rays, 100.0 * regions[4] / max(rays, 1)))
    print('    wave cycles: %.0f a node trip, %.0f a leaf trip (%.0f a record pass); node trips %.1f%% / leaf trips %.1f%% of the TRACE stage' % (
        c[25] / max(c[10], 1), c[26] / max(c[14], 1), c[26] / max(c[12], 1), 100 * c[25] / max(c[17], 1), 100 * c[26] / max(c[17], 1)))
elif c[15] or c[28]:
    inv = max(c[1], 1)
    print('  typed leaf loops per TRACE invocation: triangle trips %.2f, plane trips %.2f, sphere trips %.2f' % (c[12] / inv, c[15] / inv, c[28] / inv))
if (c[29] or c[30]) and not (c[14] > 0.5 * c[1] and c[25]):
    tri_l = c[13] - c[29] - c[30]
    print('  flat primitive loops, lanes busy: triangles %.1f%% of %.3e wave trips, planes %.1f%% of %.3e, spheres %.1f%% of %.3e' % (
        100 * tri_l / max(64 * c[12], 1), c[12], 100 * c[29] / max(64 * c[15], 1), c[15], 100 * c[30] / max(64 * c[28], 1), c[28]))
if c[23]:
    print('  SHADE: NEE evaluation branch taken in %.1f%% of the invocations at %.1f%% of the stage\'s lanes; hits on triangles %.1f%%, spheres %.1f%%' % (
        100 * c[23] / max(c[3], 1), 100 * c[24] / max(c[8], 1), 100 * c[27] / max(c[8], 1), 100 * c[31] / max(c[8], 1)))
