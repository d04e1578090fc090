"""Differential fuzz of the device alias-table builder against the host builder: random sizes (1 .. ~40,000 texels) and random texel statistics
(sun-like, flat, log-uniform, sparse, with negative texels), every field of every entry and the leftover count compared bit for bit.
    python tools/fuzz_alias.py [trials] [seed]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import util
import rsoderh_raytracing_amd as R
from rsoderh_raytracing_amd import state, types as T
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
L = state.lib()
L.rsrt_environment_build_alias.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p]
st = R.State(0)
st.upload_scene(R.Scene.load_toml(util.scene_path('default')))
bad = 0
for t in range(trials):
    rng = np.random.default_rng(seed0 + t)
    w, h = int(rng.integers(1, 300)), int(rng.integers(1, 140))
    kind = t % 6
    if kind == 0: v = rng.uniform(0.9, 1.1, (h, w, 3))
    elif kind == 1: v = 10.0 ** rng.uniform(-3, 3, (h, w, 3))
    elif kind == 2: v = np.where(rng.uniform(size=(h, w, 1)) < 0.01, 10.0 ** rng.uniform(2, 5, (h, w, 3)), rng.uniform(0, 0.2, (h, w, 3)))  # a few suns
    elif kind == 3: v = np.where(rng.uniform(size=(h, w, 1)) < 0.5, 0.0, rng.uniform(0, 2, (h, w, 3)))  # half black
    elif kind == 4: v = np.round(rng.uniform(0, 4, (h, w, 3)))  # few distinct values: many exact ties, p == 1 exactly somewhere
    else: v = np.where(rng.uniform(size=(h, w, 1)) < 0.03, -1.0, 1.0) * 10.0 ** rng.uniform(-1, 1, (h, w, 3))  # some negative texels
    rgba = np.ascontiguousarray(np.concatenate([v, np.zeros((h, w, 1))], axis=2).astype(np.float32))
    ref, left_ref = R.AliasTable.build_by_luminance(rgba[:, :, :3])
    assert L.rsrt_upload_environment(st._ctx, 0, w, h, rgba.ctypes.data_as(C.c_void_p), None) == 0
    out = np.zeros(w * h, T.ALIAS_ENTRY); left = C.c_uint32(0)
    assert L.rsrt_environment_build_alias(st._ctx, 0, out.ctypes.data_as(C.c_void_p), out.size, C.byref(left)) == 0
    ok = left.value == left_ref and all(np.array_equal(out[n].view(np.uint32), ref[n].view(np.uint32)) for n in ('probability', 'alias_index', 'pmf'))
    if not ok:
        bad += 1
        print('MISMATCH seed %d kind %d %dx%d: leftover %d vs %d' % (seed0 + t, kind, w, h, left.value, left_ref), flush=True)
print('fuzz_alias: %d trials (seeds %d..%d), %d mismatches' % (trials, seed0, seed0 + trials - 1, bad))
st.close()
sys.exit(1 if bad else 0)
