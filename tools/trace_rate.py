"""Feasibility probe for DESIGN §9 ("more waves in flight for TRACE"): how fast is the flat traversal when it runs as a kernel
of its own — rt_cast_rays_kernel<LDS image, flat>, no pool, no scheduler, whatever occupancy its own register count
allows — on incoherent secondary rays of the BASELINE scene?  Rays: camera rays -> hit points -> cosine-distributed
directions about the normal (two generations), plus rays from the same points towards the synthetic sun.
Run under rocprofv3 --kernel-trace --stats (tools/trace_rate.sh); the kernel's average duration / the ray count is the figure.
    python tools/trace_rate.py [million rays] [repeats]"""
import os, sys
os.environ.setdefault('RSRT_PROBE_REPEAT', os.environ.get('TRACE_REPEAT', '32'))  # read once, by rsrt_context_create
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import util
import rsoderh_raytracing_amd as R
n = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 4_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = np.random.default_rng(5)
env = R.Environment.synthetic(64, 32)
sc = R.Scene.load_toml(util.scene_path('house'))
st = R.State.new(sc, env, 64, 64)
MODE = (3 << 1) | 16 | 1  # flat traversal, scene in LDS as the production kernel stages it, cast_ray_bvh only
cam = sc.camera_uniform()
pos = np.array(cam['pos'], np.float32).reshape(-1)[:3]

def around(nrm, k):
    u1, u2 = rng.random(k, np.float32), rng.random(k, np.float32)
    r, phi = np.sqrt(u1), 2 * np.pi * u2
    lx, ly, lz = r * np.cos(phi), r * np.sin(phi), np.sqrt(np.maximum(0, 1 - u1))
    a = np.where(np.abs(nrm[:, :1]) > 0.9, np.array([[0, 1, 0]], np.float32), np.array([[1, 0, 0]], np.float32))
    t = np.cross(a, nrm); t /= np.linalg.norm(t, axis=1, keepdims=True)
    b = np.cross(nrm, t)
    return (t * lx[:, None] + b * ly[:, None] + nrm * lz[:, None]).astype(np.float32)

# generation 0: rays from the camera position over a 100-degree cone about -z
k = n // 2
d0 = rng.normal(size=(k, 3)).astype(np.float32) * np.array([1.2, 0.7, 0], np.float32) + np.array([0, 0, -1], np.float32)
d0 /= np.linalg.norm(d0, axis=1, keepdims=True)
o0 = np.tile(pos, (k, 1)).astype(np.float32)
h0 = st.cast_rays(o0, d0, MODE)
hit = h0['did_hit'] != 0
p1, n1 = h0['hit_point'][hit], h0['normal'][hit]
d1 = around(n1, len(p1))
h1 = st.cast_rays(p1, d1, MODE)
hit1 = h1['did_hit'] != 0
p2, n2 = h1['hit_point'][hit1], h1['normal'][hit1]
d2 = around(n2, len(p2))
sun = np.array([0.3, 0.8, 0.5], np.float32); sun /= np.linalg.norm(sun)
ds = np.tile(sun, (len(p1), 1)) + rng.normal(size=(len(p1), 3)).astype(np.float32) * 0.02
ds = (ds / np.linalg.norm(ds, axis=1, keepdims=True)).astype(np.float32)
O = np.concatenate([p1, p2, p1])[:n]; D = np.concatenate([d1, d2, ds])[:n]
perm = rng.permutation(len(O)); O, D = np.ascontiguousarray(O[perm]), np.ascontiguousarray(D[perm])
print('rays %d: %d first-bounce, %d second-bounce, %d towards the sun; camera rays that hit: %.0f%%' % (len(O), len(p1), len(p2), len(p1), 100 * hit.mean()), flush=True)
REPEAT = int(os.environ.get('TRACE_REPEAT', '32'))
# (RSRT_PROBE_REPEAT is read when the context is created: set at the top of this script)
for _ in range(reps):
    h = st.cast_rays(O, D, MODE)
print('hit fraction of the secondary set %.2f' % (h['did_hit'] != 0).mean())
print('RAYS_PER_LAUNCH', len(O), 'REPEAT', REPEAT)
