"""ctypes binding of the CPU oracle (oracle/rt_oracle.cpp).

TEST INFRASTRUCTURE, NOT PRODUCT: only tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of bench.py may import this package.  Parity is UNPINNED by the reference
(it has no tests or golden vectors); the hand-derivable KATs pin it instead.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# encase std430/std140 layouts uploaded by the reference (src/state.rs:394-458)
MATERIAL = np.dtype({"names": ["color", "roughness", "metallic", "emission"],
                     "formats": [("f4", 3), "f4", "f4", ("f4", 3)], "offsets": [0, 12, 16, 32], "itemsize": 48})
SPHERE = np.dtype({"names": ["pos", "radius", "material_id"], "formats": [("f4", 3), "f4", "u4"],
                   "offsets": [0, 12, 16], "itemsize": 32})
PLANE = np.dtype({"names": ["pos", "normal", "m", "material_id"], "formats": [("f4", 3), ("f4", 3), ("f4", (3, 4)), "u4"],
                  "offsets": [0, 16, 32, 80], "itemsize": 96})
VEC3 = np.dtype({"names": ["v"], "formats": [("f4", 3)], "offsets": [0], "itemsize": 16})
TRIANGLE = np.dtype([("v0", "u4"), ("v1", "u4"), ("v2", "u4"), ("n0", "u4"), ("n1", "u4"), ("n2", "u4"),
                     ("material_id", "u4")])
PRIM_INFO = np.dtype([("type", "u4"), ("index", "u4")])
BVH_NODE = np.dtype({"names": ["bmin", "bmax", "idx", "len", "axis"], "formats": [("f4", 3), ("f4", 3), "u4", "u4", "u4"],
                     "offsets": [0, 16, 32, 36, 40], "itemsize": 48})
ALIAS_ENTRY = np.dtype([("probability", "f4"), ("alias_index", "u4"), ("pmf", "f4"), ("_pad", "u4")])
CAMERA = np.dtype({"names": ["pos", "rot", "fov_y"], "formats": [("f4", 3), ("f4", (3, 4)), "f4"],
                   "offsets": [0, 16, 64], "itemsize": 80})
PLANE_SRC = np.dtype([("pos", "f4", 3), ("forward", "f4", 3), ("right", "f4", 3), ("material_id", "u4")])
HIT = np.dtype([("did_hit", "u4"), ("distance", "f4"), ("hit_point", "f4", 3), ("normal", "f4", 3), ("material_id", "u4")])

STAT_FIELDS = ["paths", "ext_rays", "shadow_rays", "nodes_visited", "prim_refs", "sphere_tests", "plane_tests",
               "tri_tests", "closest_tri", "nee_events", "escapes", "shaded_hits", "fallback_sphere_tests",
               "fallback_plane_tests", "f32_ops", "int_ops"]  # the last two: only fast="ops" (liboracle_ops.so) fills them
FLAG_PRUNE = 1
FLAG_ANYHIT_SHADOW = 2


class _Scene(C.Structure):
    _fields_ = [(n, t) for pair in [("materials", "n_materials"), ("spheres", "n_spheres"), ("planes", "n_planes"),
                                    ("vertices", "n_vertices"), ("normals", "n_normals"), ("triangles", "n_triangles"),
                                    ("prims", "n_prims"), ("nodes", "n_nodes")]
                for n, t in ((pair[0], C.c_void_p), (pair[1], C.c_uint32))]


class _Env(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rgba", C.c_void_p), ("alias", C.c_void_p)]


class _Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in STAT_FIELDS]


def _cpu_has_fma():
    try:
        with open("/proc/cpuinfo") as f:
            return " fma " in f.read()
    except OSError:
        return False


def build(force=False):
    """Compile liboracle.so / liboracle_fast.so (gcc, a few seconds)."""
    args = ["make", "-C", _HERE, "FMA=%d" % (1 if _cpu_has_fma() else 0)]
    if force:
        args.append("-B")
    subprocess.run(args, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    with open(os.path.join(_HERE, ".built_fma"), "w") as f:
        f.write("%d" % (1 if _cpu_has_fma() else 0))


_libs = {}


def lib(fast=False):
    """fast: False = strict build (defines the golden values), True = -O3 twin (bit-identical), "libm" = the
    sensitivity build (libm transcendentals, free contraction: NOT a parity reference), "ops" = the counting build
    (strict arithmetic + orc_stats.f32_ops / int_ops: the algorithmic work of the integrator)."""
    name = {"libm": "liboracle_libm.so", "ops": "liboracle_ops.so"}.get(fast, "liboracle_fast.so" if fast else "liboracle.so")
    if name in _libs:
        return _libs[name]
    path = os.path.join(os.environ.get("ORACLE_LIB_DIR") or _HERE, name)  # ORACLE_LIB_DIR: sanitizer builds (tools/sanitize_host.sh)
    marker = os.path.join(_HERE, ".built_fma")
    built_fma = open(marker).read().strip() if os.path.exists(marker) else "1"
    if not os.path.exists(path) or (built_fma == "1" and not _cpu_has_fma()):
        build(force=os.path.exists(path))
    L = C.CDLL(path)
    L.orc_render.restype = C.c_int
    L.orc_build_bvh.restype = C.c_int
    L.orc_rng_seed.restype = C.c_uint32
    L.orc_rng_next_u32.restype = C.c_uint32
    L.orc_u32_to_uniform.restype = C.c_float
    L.orc_u32_to_uniform.argtypes = [C.c_uint32]
    L.orc_bsdf_pdf_local.restype = C.c_float
    L.orc_bsdf_sample.restype = C.c_float
    L.orc_environment_direction_pdf.restype = C.c_float
    L.orc_sample_environment.restype = C.c_float
    L.orc_detmath.restype = C.c_float
    L.orc_detmath.argtypes = [C.c_int, C.c_float, C.c_float]
    L.orc_camera_uniform.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p]
    _libs[name] = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


def _f3(v):
    return np.ascontiguousarray(v, dtype=np.float32)


class Scene:
    """Holds the eight group-2 arrays (kept alive) and the C view of them."""

    ARRAYS = [("materials", MATERIAL), ("spheres", SPHERE), ("planes", PLANE), ("vertices", VEC3), ("normals", VEC3),
              ("triangles", TRIANGLE), ("prims", PRIM_INFO), ("nodes", BVH_NODE)]

    def __init__(self, **arrays):
        self.c = _Scene()
        for name, dt in self.ARRAYS:
            a = np.ascontiguousarray(arrays[name]).view(dt).reshape(-1) if len(arrays[name]) else np.zeros(0, dt)
            setattr(self, name, a)
            setattr(self.c, name, _p(a))
            setattr(self.c, "n_" + name, len(a))


class Env:
    def __init__(self, rgba, alias):
        self.rgba = np.ascontiguousarray(rgba, dtype=np.float32)
        assert self.rgba.ndim == 3 and self.rgba.shape[2] == 4
        self.height, self.width = self.rgba.shape[:2]
        self.alias = np.ascontiguousarray(alias).view(ALIAS_ENTRY).reshape(-1)
        assert len(self.alias) == self.width * self.height
        self.c = _Env(self.width, self.height, _p(self.rgba), _p(self.alias))


def build_bvh(spheres, planes_src, vertices, triangles):
    n = len(spheres) + len(planes_src) + len(triangles)
    prims = np.zeros(n, PRIM_INFO)
    nodes = np.zeros(2 * n, BVH_NODE)
    depth = C.c_uint32(0)
    cnt = lib().orc_build_bvh(_p(spheres), C.c_uint32(len(spheres)), _p(planes_src), C.c_uint32(len(planes_src)),
                              _p(vertices), _p(triangles), C.c_uint32(len(triangles)), _p(prims), _p(nodes), C.byref(depth))
    if cnt < 0:
        raise ValueError("orc_build_bvh: empty scene")
    return prims, nodes[:cnt].copy(), depth.value


def alias_table(rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    h, w = rgb.shape[:2]
    out = np.zeros(w * h, ALIAS_ENTRY)
    left = C.c_uint32(0)
    lib().orc_alias_table(C.c_uint32(w), C.c_uint32(h), _p(rgb), _p(out), C.byref(left))
    return out, left.value


def plane_to_uniform(planes_src):
    out = np.zeros(len(planes_src), PLANE)
    for i in range(len(planes_src)):
        lib().orc_plane_to_uniform(C.c_void_p(planes_src.ctypes.data + i * PLANE_SRC.itemsize),
                                   C.c_void_p(out.ctypes.data + i * PLANE.itemsize))
    return out


def camera_uniform(pos, yaw, pitch, fov_y):
    out = np.zeros(1, CAMERA)
    lib().orc_camera_uniform(_p(_f3(pos)), yaw, pitch, fov_y, _p(out))
    return out


def render(scene, env, camera, width, height, sample_begin, sample_count, max_bounces, flags=0, n_threads=0,
           sum_rgba=None, fast=False):
    """Returns (sum_rgba[H,W,4], stats dict); adds into ``sum_rgba`` when given."""
    if sum_rgba is None:
        sum_rgba = np.zeros((height, width, 4), np.float32)
    assert sum_rgba.dtype == np.float32 and sum_rgba.shape == (height, width, 4) and sum_rgba.flags.c_contiguous
    st = _Stats()
    cam = np.ascontiguousarray(camera).view(CAMERA)
    rc = lib(fast).orc_render(C.byref(scene.c), C.byref(env.c), _p(cam), C.c_uint32(width), C.c_uint32(height),
                              C.c_uint32(sample_begin), C.c_uint32(sample_count), C.c_uint32(max_bounces),
                              C.c_uint32(flags), C.c_int(n_threads), _p(sum_rgba), C.byref(st))
    if rc != 0:
        raise RuntimeError("orc_render failed: %d" % rc)
    return sum_rgba, {n: getattr(st, n) for n in STAT_FIELDS}


def cast_rays(scene, origins, dirs, mode=0, flags=0):
    o, d = _f3(origins).reshape(-1, 3), _f3(dirs).reshape(-1, 3)
    out = np.zeros(len(o), HIT)
    lib().orc_cast_rays(C.byref(scene.c), C.c_uint32(len(o)), _p(o), _p(d), C.c_uint32(mode), C.c_uint32(flags), _p(out))
    return out


def rng_seed(pixel, sample):
    return lib().orc_rng_seed(C.c_uint32(pixel), C.c_uint32(sample))


def rng_draws(state, n):
    s = C.c_uint32(state)
    return [lib().orc_rng_next_u32(C.byref(s)) for _ in range(n)], s.value


def detmath(fn, a, b=0.0):
    return lib().orc_detmath({"sin": 0, "cos": 1, "atan2": 2, "asin": 3}[fn], a, b)


def debug_view(dev_index, env, width, height, sample_count, out_texture=None):
    """The shader's developer views (shader.wgsl:1314-1338), restated with numpy: what `main` leaves in out_texture ([height, width, 4]
    float16) when dev_index is 3 (":1333 Display HDRI": the environment's texel under the pixel, saturated, alpha 0; textureLoad outside
    the map gives zeros) or 2 (":1315 Draw pixels based on distribution": per pixel, seeded as :1309-1312, twenty
    random_index_in_environment draws (:689-706), each adding vec3(0.1) / f32(20) to that texel's position of out_texture through a
    binary16 load and store, alpha 0; a store outside the texture is dropped).  The shader's invocations race on the texture; this is
    the serial execution, pixel after pixel — with equal addends the order does not matter, only that every draw lands."""
    import numpy as np
    rgba = np.asarray(env.rgba, np.float32)
    eh, ew = rgba.shape[:2]
    out = np.zeros((height, width, 4), np.float16) if out_texture is None else np.array(out_texture, np.float16, copy=True)
    if dev_index == 3:
        out[:] = 0
        h, w = min(eh, height), min(ew, width)
        out[:h, :w, :3] = np.clip(rgba[:h, :w, :3], 0.0, 1.0).astype(np.float16)
        return out
    assert dev_index == 2
    alias = np.asarray(env.alias).reshape(-1)
    length = ew * eh
    step = np.float32(0.1) / np.float32(20)
    for py in range(height):
        for px in range(width):
            draws, _ = rng_draws(rng_seed(py * width + px, sample_count), 40)
            for k in range(20):
                u1 = np.float32(lib().orc_u32_to_uniform(C.c_uint32(draws[2 * k])))
                u2 = np.float32(lib().orc_u32_to_uniform(C.c_uint32(draws[2 * k + 1])))
                f = np.float32(u1 * np.float32(length))
                index = min(int(f) if f > 0 else 0, length - 1)
                entry = alias[index]
                pick = index if u2 < np.float32(entry["probability"]) else int(entry["alias_index"])
                x, y = pick % ew, pick // ew
                if x < width and y < height:
                    out[y, x, :3] = (out[y, x, :3].astype(np.float32) + step).astype(np.float16)
                    out[y, x, 3] = 0
    return out
