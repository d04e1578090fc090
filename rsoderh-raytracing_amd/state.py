"""`State` — the host render state of the reference (src/state.rs:29-834) over librsrt.so.

    State.new(scene, environments)   State::new   (state.rs:60-649): uploads the eight scene
                                     buffers and the environment maps + alias tables
    state.resize(w, h)               State::resize (state.rs:651-666): accumulator follows the size
    state.update(camera=..)          State::update (state.rs:722-758): per-frame uniforms
    state.render()                   State::render (state.rs:760-833): one progressive frame —
                                     scene-hash reset, sample_count += 1, one dispatch
    state.render_samples(n)          the batched form the C-ABI adds: n samples in one call

No CPU fallback: constructing a State without librsrt.so / a gfx950 GPU raises RsrtError.
"""
import ctypes as C

import numpy as np

from . import _build, types as T

FLAG_REFERENCE_TRAVERSAL = 1  # shadow query runs to the end too (literal shader.wgsl:1249)
FLAG_PRUNE = 2                # opt-in t-pruning; NOT exactly result-preserving (include/rsrt.h)

_SYMBOLS = ["rsrt_context_create", "rsrt_context_destroy", "rsrt_last_error", "rsrt_upload_scene",
            "rsrt_upload_environment", "rsrt_set_partition", "rsrt_accumulator_resize", "rsrt_accumulator_bind",
            "rsrt_accumulator_clear", "rsrt_accumulator_download", "rsrt_resolve_mean_f16", "rsrt_debug_view_f16", "rsrt_render",
            "rsrt_synchronize", "rsrt_get_stats", "rsrt_cast_rays", "rsrt_describe", "rsrt_get_debug_counters", "rsrt_get_region_counters", "rsrt_display_srgb8",
            "rsrt_selftest_numerics", "rsrt_build_id", "rsrt_wide_tree_build", "rsrt_build_bvh_device",
            "rsrt_partition_owner", "rsrt_partition_mask", "rsrt_partition_tiles", "rsrt_comm_available", "rsrt_comm_unique_id", "rsrt_comm_init", "rsrt_comm_reduce", "rsrt_comm_set_mode", "rsrt_comm_destroy",
            "rsrt_multi_create", "rsrt_multi_destroy", "rsrt_multi_last_error", "rsrt_multi_size", "rsrt_multi_context",
            "rsrt_multi_upload_scene", "rsrt_multi_upload_environment", "rsrt_multi_resize", "rsrt_multi_clear", "rsrt_multi_render",
            "rsrt_multi_synchronize", "rsrt_multi_download", "rsrt_multi_display_srgb8", "rsrt_multi_get_stats", "rsrt_multi_uses_rccl"]


class RsrtError(RuntimeError):
    pass


def build_id():
    """rsrt_build_id(): which kernel sources the loaded librsrt.so was compiled from."""
    return lib().rsrt_build_id().decode()


class Stats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("ext_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("kernel_ms", C.c_double),
                ("total_paths", C.c_uint64), ("total_ext_rays", C.c_uint64), ("total_shadow_rays", C.c_uint64),
                ("total_kernel_ms", C.c_double), ("launches", C.c_uint32), ("_pad", C.c_uint32),
                ("trace_kernel_ms", C.c_double), ("resolve_kernel_ms", C.c_double), ("reduce_ms", C.c_double), ("traversal_steps", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if n != "_pad"}


_lib = None


def lib():
    """Loads librsrt.so (building it in-tree when sources are newer). Raises if it cannot be loaded."""
    global _lib
    if _lib is None:
        try:
            import os
            path = os.environ.get("RSRT_LIB") or _build.build_hip(instrument=os.environ.get("RSRT_INSTRUMENT") == "1")  # RSRT_LIB: experiment builds (tools/flag_sweep.sh)
            L = C.CDLL(path)
        except Exception as e:  # noqa: BLE001
            raise RsrtError("librsrt.so (the HIP integrator) is not available: %s" % e) from e
        for s in _SYMBOLS:
            getattr(L, s)
        L.rsrt_last_error.restype = C.c_char_p
        L.rsrt_last_error.argtypes = [C.c_void_p]
        L.rsrt_describe.restype = C.c_char_p
        L.rsrt_describe.argtypes = [C.c_void_p]
        L.rsrt_build_id.restype = C.c_char_p
        L.rsrt_build_id.argtypes = []
        L.rsrt_context_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.rsrt_context_destroy.argtypes = [C.c_void_p]
        L.rsrt_upload_scene.argtypes = [C.c_void_p] + [C.c_void_p, C.c_uint32] * 8
        L.rsrt_upload_environment.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.rsrt_set_partition.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.rsrt_accumulator_resize.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.rsrt_accumulator_bind.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.rsrt_accumulator_clear.argtypes = [C.c_void_p]
        L.rsrt_accumulator_download.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.rsrt_resolve_mean_f16.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]
        L.rsrt_debug_view_f16.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t]
        L.rsrt_display_srgb8.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]
        L.rsrt_render.argtypes = [C.c_void_p, C.c_void_p] + [C.c_uint32] * 7 + [C.c_void_p]
        L.rsrt_synchronize.argtypes = [C.c_void_p]
        L.rsrt_get_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.rsrt_get_debug_counters.argtypes = [C.c_void_p, C.c_void_p]
        L.rsrt_get_region_counters.argtypes = [C.c_void_p, C.c_void_p]
        L.rsrt_selftest_numerics.argtypes = [C.c_void_p, C.c_void_p]
        L.rsrt_cast_rays.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.rsrt_partition_owner.restype = C.c_uint32
        L.rsrt_partition_owner.argtypes = [C.c_uint32] * 7
        L.rsrt_partition_mask.argtypes = [C.c_uint32] * 6 + [C.c_void_p, C.c_void_p]
        L.rsrt_partition_tiles.argtypes = [C.c_uint32] * 6 + [C.c_void_p, C.c_void_p]
        L.rsrt_comm_available.restype = C.c_int
        L.rsrt_comm_available.argtypes = []
        L.rsrt_comm_unique_id.argtypes = [C.c_void_p]
        L.rsrt_comm_init.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.rsrt_comm_reduce.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.rsrt_comm_destroy.argtypes = [C.c_void_p]
        L.rsrt_comm_set_mode.argtypes = [C.c_void_p, C.c_uint32]
        L.rsrt_multi_create.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
        L.rsrt_multi_destroy.argtypes = [C.c_void_p]
        L.rsrt_multi_last_error.restype = C.c_char_p
        L.rsrt_multi_last_error.argtypes = [C.c_void_p]
        L.rsrt_multi_size.restype = C.c_uint32
        L.rsrt_multi_size.argtypes = [C.c_void_p]
        L.rsrt_multi_context.restype = C.c_void_p
        L.rsrt_multi_context.argtypes = [C.c_void_p, C.c_uint32]
        L.rsrt_multi_upload_scene.argtypes = [C.c_void_p] + [C.c_void_p, C.c_uint32] * 8
        L.rsrt_multi_upload_environment.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.rsrt_multi_resize.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.rsrt_multi_clear.argtypes = [C.c_void_p]
        L.rsrt_multi_render.argtypes = [C.c_void_p, C.c_void_p] + [C.c_uint32] * 7
        L.rsrt_multi_synchronize.argtypes = [C.c_void_p]
        L.rsrt_multi_download.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.rsrt_multi_display_srgb8.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]
        L.rsrt_multi_get_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.rsrt_multi_uses_rccl.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


class State:
    def __init__(self, device=0):
        self._L = lib()
        self._ctx = C.c_void_p()
        rc = self._L.rsrt_context_create(device, C.byref(self._ctx))
        if rc != 0:
            raise RsrtError("rsrt_context_create: %s" % self._L.rsrt_last_error(None).decode())
        self.width = self.height = 0
        self.sample_count = 0          # hdr.sample_count (state.rs:775-789)
        self.max_bounces = 10          # MAX_BOUNCES (shader.wgsl:232)
        self.environment_index = 0     # state.rs:638
        self.camera = None
        self._last_hash = None
        self.flags = 0

    # -- construction ---------------------------------------------------------------------------
    @classmethod
    def new(cls, scene, environments, width, height, device=0, camera=None):
        st = cls(device)
        st.upload_scene(scene)
        for i, env in enumerate(environments if isinstance(environments, (list, tuple)) else [environments]):
            st.upload_environment(i, env)
        st.resize(width, height)
        st.camera = np.array(camera if camera is not None else scene.camera_uniform()).view(T.CAMERA).reshape(1).copy()
        return st

    def _check(self, rc, what):
        if rc != 0:
            raise RsrtError("%s failed (%d): %s" % (what, rc, self._L.rsrt_last_error(self._ctx).decode()))

    def close(self):
        if self._ctx:
            self._L.rsrt_context_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def describe(self):
        return self._L.rsrt_describe(self._ctx).decode()

    def upload_scene(self, scene):
        a = [scene.materials, scene.spheres, scene.planes, scene.vertices, scene.normals, scene.triangles, scene.primitives,
             scene.bvh_nodes]
        dts = [T.MATERIAL, T.SPHERE, T.PLANE, T.VEC3, T.VEC3, T.TRIANGLE, T.PRIMITIVE_INFO, T.BVH_NODE]
        args = []
        for arr, dt in zip(a, dts):
            arr = np.ascontiguousarray(arr)
            assert arr.dtype.itemsize == dt.itemsize, (arr.dtype, dt)
            args += [_p(arr), len(arr)]
        self._keep = a
        self._check(self._L.rsrt_upload_scene(self._ctx, *args), "rsrt_upload_scene")

    def upload_environment(self, slot, env):
        rgba = np.ascontiguousarray(env.rgba, dtype=np.float32)
        alias = np.ascontiguousarray(env.alias)
        assert rgba.shape == (env.height, env.width, 4) and alias.dtype.itemsize == 16 and len(alias) == env.width * env.height
        self._check(self._L.rsrt_upload_environment(self._ctx, slot, env.width, env.height, _p(rgba), _p(alias)),
                    "rsrt_upload_environment")

    def set_partition(self, rank, world_size, tile_w=16, tile_h=16):
        self._check(self._L.rsrt_set_partition(self._ctx, rank, world_size, tile_w, tile_h), "rsrt_set_partition")

    # -- multi-GPU, one process per GPU: the RCCL reduce lives in the library (include/rsrt.h) -----
    @staticmethod
    def comm_unique_id():
        """Rank 0: the 128 bytes every rank passes to comm_init (hand them over by any channel)."""
        L = lib()
        buf = C.create_string_buffer(128)
        rc = L.rsrt_comm_unique_id(buf)
        if rc != 0:
            raise RsrtError("rsrt_comm_unique_id failed (%d): %s" % (rc, L.rsrt_last_error(None).decode()))
        return buf.raw

    @staticmethod
    def comm_available():
        """True when librccl can be loaded (a dlopen, no collective): every rank asks BEFORE the collective comm_init."""
        return bool(lib().rsrt_comm_available())

    def comm_init(self, rank, world_size, unique_id):
        assert len(unique_id) == 128
        self._check(self._L.rsrt_comm_init(self._ctx, rank, world_size, C.create_string_buffer(unique_id, 128)), "rsrt_comm_init")

    def comm_reduce(self, root=0, recv_ptr=None, stream=None):
        """The exchange step: every rank's tiles onto `root` over RCCL (in place unless recv_ptr names a device buffer)."""
        self._check(self._L.rsrt_comm_reduce(self._ctx, root, C.c_void_p(recv_ptr) if recv_ptr else None,
                                             C.c_void_p(stream) if stream else None), "rsrt_comm_reduce")

    def comm_set_mode(self, dense_reduce):
        """False: the exchange is the gather of compact tile buffers (default); True: the dense ncclReduce of the full accumulators."""
        self._check(self._L.rsrt_comm_set_mode(self._ctx, 1 if dense_reduce else 0), "rsrt_comm_set_mode")

    def comm_destroy(self):
        self._check(self._L.rsrt_comm_destroy(self._ctx), "rsrt_comm_destroy")

    # -- State::resize / update / render --------------------------------------------------------
    def resize(self, width, height):
        self._check(self._L.rsrt_accumulator_resize(self._ctx, width, height), "rsrt_accumulator_resize")
        self.width, self.height = width, height
        self._last_hash = None

    def bind_accumulator(self, device_ptr, width, height):
        """Use caller-owned device memory (W*H*4 f32), e.g. a torch tensor's data_ptr()."""
        self._check(self._L.rsrt_accumulator_bind(self._ctx, C.c_void_p(device_ptr), width, height), "rsrt_accumulator_bind")
        self.width, self.height = width, height
        self._last_hash = None

    def update(self, camera=None, environment_index=None):
        if camera is not None:
            self.camera = np.array(camera).view(T.CAMERA).reshape(1).copy()
        if environment_index is not None:
            self.environment_index = environment_index

    def _scene_hash(self):
        return hash((self.camera.tobytes(), self.environment_index))

    def clear(self):
        self._check(self._L.rsrt_accumulator_clear(self._ctx), "rsrt_accumulator_clear")
        self.sample_count = 0

    def render_samples(self, n, stream=None):
        """Adds samples [sample_count, sample_count+n); resets first when camera/environment changed."""
        h = self._scene_hash()
        if h != self._last_hash:  # state.rs:778-786
            self._last_hash = h
            self.clear()
        self._check(self._L.rsrt_render(self._ctx, _p(self.camera), self.width, self.height, self.sample_count, n,
                                        self.max_bounces, self.environment_index, self.flags,
                                        C.c_void_p(stream) if stream else None), "rsrt_render")
        self.sample_count += n

    def render(self):
        """One reference frame: exactly one more sample per pixel."""
        self.render_samples(1)

    def render_range(self, sample_begin, sample_count, stream=None):
        """Raw rsrt_render: no hash check, no counter update."""
        self._check(self._L.rsrt_render(self._ctx, _p(self.camera), self.width, self.height, sample_begin, sample_count,
                                        self.max_bounces, self.environment_index, self.flags,
                                        C.c_void_p(stream) if stream else None), "rsrt_render")

    def synchronize(self):
        self._check(self._L.rsrt_synchronize(self._ctx), "rsrt_synchronize")

    # -- results ---------------------------------------------------------------------------------
    def download(self):
        """cumulative_light_texture: [H, W, 4] float32 sums, alpha 1 where rendered."""
        out = np.empty((self.height, self.width, 4), np.float32)
        self._check(self._L.rsrt_accumulator_download(self._ctx, _p(out), out.size), "rsrt_accumulator_download")
        return out

    def download_mean_f16(self, sample_total=None):
        """out_texture: [H, W, 4] float16 mean radiance."""
        out = np.empty((self.height, self.width, 4), np.float16)
        n = sample_total if sample_total is not None else self.sample_count
        self._check(self._L.rsrt_resolve_mean_f16(self._ctx, n, _p(out), out.size), "rsrt_resolve_mean_f16")
        return out

    def debug_view(self, dev_index, out_texture=None, sample_count=None, environment_index=None):
        """The reference's developer views (shader.wgsl:1314-1338) as out_texture, [H, W, 4] float16: dev_index 3 = the HDRI, 2 = draws
        of the alias table added onto `out_texture` (the previous frame's; zeros when None).  See rsrt_debug_view_f16."""
        out = np.zeros((self.height, self.width, 4), np.float16) if out_texture is None else np.ascontiguousarray(out_texture, np.float16).copy()
        assert out.shape == (self.height, self.width, 4)
        n = self.sample_count if sample_count is None else sample_count
        e = self.environment_index if environment_index is None else environment_index
        self._check(self._L.rsrt_debug_view_f16(self._ctx, dev_index, e, n, _p(out), out.size), "rsrt_debug_view_f16")
        return out

    def display_srgb8(self, sample_total=None):
        """What the reference shows on screen: [H, W, 4] uint8 (ACES tonemap of the f16 mean, sRGB encoded)."""
        out = np.empty((self.height, self.width, 4), np.uint8)
        n = sample_total if sample_total is not None else self.sample_count
        self._check(self._L.rsrt_display_srgb8(self._ctx, n, _p(out), out.size), "rsrt_display_srgb8")
        return out

    def stats(self):
        s = Stats()
        self._check(self._L.rsrt_get_stats(self._ctx, C.byref(s)), "rsrt_get_stats")
        return s.as_dict()

    def debug_counters(self):
        out = np.zeros(32, np.uint64)
        self._check(self._L.rsrt_get_debug_counters(self._ctx, _p(out)), "rsrt_get_debug_counters")
        return out

    def region_counters(self):
        """Lanes that passed each region mark since the last call (instrumented build; zeros in the product build)."""
        out = np.zeros(32, np.uint64)
        self._check(self._L.rsrt_get_region_counters(self._ctx, _p(out)), "rsrt_get_region_counters")
        return out

    def selftest_numerics(self):
        """Exhaustive device check of the short reciprocal (all 2^32 inputs): dict of the four words of rsrt_selftest_numerics."""
        out = np.zeros(4, np.uint64)
        self._check(self._L.rsrt_selftest_numerics(self._ctx, _p(out)), "rsrt_selftest_numerics")
        return {"mismatches": int(out[0]), "short_path_inputs": int(out[1]), "bare_rcp_wrong": int(out[2]), "first_bad": int(out[3])}

    def build_bvh_device(self, spheres, plane_descs, vertices, triangles):
        """build_bvh on the device (rsrt_build_bvh_device): (primitives, nodes, depth, device milliseconds) — the host
        builder's arrays, bit for bit."""
        sph, pls = np.ascontiguousarray(spheres).view(T.SPHERE).reshape(-1), np.ascontiguousarray(plane_descs).view(T.PLANE_DESC).reshape(-1)
        ver, tri = np.ascontiguousarray(vertices).view(T.VEC3).reshape(-1), np.ascontiguousarray(triangles).view(T.TRIANGLE).reshape(-1)
        n = len(sph) + len(pls) + len(tri)
        prims, nodes = np.zeros(max(n, 1), T.PRIMITIVE_INFO), np.zeros(max(2 * n, 1), T.BVH_NODE)
        n_nodes, depth, ms = C.c_uint32(0), C.c_uint32(0), C.c_double(0.0)
        self._check(self._L.rsrt_build_bvh_device(self._ctx, _p(sph), len(sph), _p(pls), len(pls), _p(ver), len(ver), _p(tri), len(tri), _p(prims), _p(nodes),
                                                  C.byref(n_nodes), C.byref(depth), C.byref(ms)), "rsrt_build_bvh_device")
        return prims[:n], nodes[:n_nodes.value].copy(), depth.value, ms.value

    def cast_rays(self, origins, directions, mode=0, flags=0):
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
        assert o.shape == d.shape
        out = np.zeros(len(o), T.HIT)
        self._check(self._L.rsrt_cast_rays(self._ctx, len(o), _p(o), _p(d), mode, flags, _p(out)), "rsrt_cast_rays")
        return out


class MultiState:
    """`State` over a LIST of devices of one node, driven by one thread (rsrt_multi_*, include/rsrt.h): device i renders
    the tiles of rank i (partition.py), the frame is gathered onto devices[0] by RCCL inside the library when it is asked for."""

    def __init__(self, scene, environments, width, height, devices=(0,), camera=None):
        self._L = lib()
        self._m = C.c_void_p()
        devs = (C.c_int * len(devices))(*devices)
        rc = self._L.rsrt_multi_create(devs, len(devices), C.byref(self._m))
        if rc != 0:
            raise RsrtError("rsrt_multi_create failed (%d): %s" % (rc, self._L.rsrt_multi_last_error(None).decode()))
        a = [scene.materials, scene.spheres, scene.planes, scene.vertices, scene.normals, scene.triangles, scene.primitives, scene.bvh_nodes]
        args = []
        for arr in a:
            arr = np.ascontiguousarray(arr)
            args += [_p(arr), len(arr)]
        self._check(self._L.rsrt_multi_upload_scene(self._m, *args), "rsrt_multi_upload_scene")
        for i, env in enumerate(environments if isinstance(environments, (list, tuple)) else [environments]):
            rgba, alias = np.ascontiguousarray(env.rgba, dtype=np.float32), np.ascontiguousarray(env.alias)
            self._check(self._L.rsrt_multi_upload_environment(self._m, i, env.width, env.height, _p(rgba), _p(alias)), "rsrt_multi_upload_environment")
        self._check(self._L.rsrt_multi_resize(self._m, width, height), "rsrt_multi_resize")
        self.width, self.height = width, height
        self.camera = np.array(camera if camera is not None else scene.camera_uniform()).view(T.CAMERA).reshape(1).copy()
        self.max_bounces, self.environment_index, self.flags, self.sample_count = 10, 0, 0, 0

    def _check(self, rc, what):
        if rc != 0:
            raise RsrtError("%s failed (%d): %s" % (what, rc, self._L.rsrt_multi_last_error(self._m).decode()))

    def close(self):
        if self._m:
            self._L.rsrt_multi_destroy(self._m)
            self._m = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def size(self):
        return self._L.rsrt_multi_size(self._m)

    def uses_rccl(self):
        return bool(self._L.rsrt_multi_uses_rccl(self._m))

    def clear(self):
        self._check(self._L.rsrt_multi_clear(self._m), "rsrt_multi_clear")
        self.sample_count = 0

    def render_samples(self, n):
        self._check(self._L.rsrt_multi_render(self._m, _p(self.camera), self.width, self.height, self.sample_count, n, self.max_bounces,
                                              self.environment_index, self.flags), "rsrt_multi_render")
        self.sample_count += n

    def download(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        self._check(self._L.rsrt_multi_download(self._m, _p(out), out.size), "rsrt_multi_download")
        return out

    def display_srgb8(self):
        out = np.empty((self.height, self.width, 4), np.uint8)
        self._check(self._L.rsrt_multi_display_srgb8(self._m, self.sample_count, _p(out), out.size), "rsrt_multi_display_srgb8")
        return out

    def stats(self):
        s = Stats()
        self._check(self._L.rsrt_multi_get_stats(self._m, C.byref(s)), "rsrt_multi_get_stats")
        return s.as_dict()
