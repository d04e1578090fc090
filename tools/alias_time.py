"""Scratch: the device alias-table build of the 2048 x 1024 synthetic environment, three times (for rocprofv3 --kernel-trace --stats):
    rocprofv3 --kernel-trace --stats -d gpurun_out/alias_prof -- python3 tools/alias_time.py"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import util
import rsoderh_raytracing_amd as R
from rsoderh_raytracing_amd import host, state, types as T
w, h = 2048, 1024
rgba = host.synth_environment(w, h)
L = state.lib()
L.rsrt_environment_build_alias.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p]
st = R.State(0)
st.upload_scene(R.Scene.load_toml(util.scene_path('default')))
assert L.rsrt_upload_environment(st._ctx, 0, w, h, rgba.ctypes.data_as(C.c_void_p), None) == 0
out = np.zeros(w * h, T.ALIAS_ENTRY); left = C.c_uint32(0)
for k in range(3):
    t = time.perf_counter()
    assert L.rsrt_environment_build_alias(st._ctx, 0, out.ctypes.data_as(C.c_void_p), out.size, C.byref(left)) == 0
    print('build %d: %.1f ms incl. copy back' % (k, (time.perf_counter() - t) * 1e3), flush=True)
st.close()
