"""Host-side scene layer: ctypes binding of librsrt_host.so (include/rsrt_host.h).

Mirrors the reference's L2 functions by name: ``Scene.load_toml`` (src/scene.rs:235),
``build_bvh`` (src/bvh.rs:13), ``AliasTable.build_by_luminance`` (src/environments.rs:97),
``Plane.to_uniform`` (src/scene.rs:191), ``CameraUniform.new`` (src/camera.rs:112).
"""
import ctypes as C
import os

import numpy as np

from . import _build, types as T

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.environ.get("RSRT_HOST_LIB") or _build.build_host()  # RSRT_HOST_LIB: a sanitizer build (tools/sanitize_host.sh)
        L = C.CDLL(path)
        L.rsrt_scene_load_toml.restype = C.c_int
        L.rsrt_scene_load_toml.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
        L.rsrt_scene_free.argtypes = [C.c_void_p]
        L.rsrt_scene_get_counts.argtypes = [C.c_void_p, C.c_void_p]
        L.rsrt_scene_get_camera.argtypes = [C.c_void_p, C.c_void_p]
        for n in ("materials", "spheres", "plane_descs", "planes", "vertices", "normals", "triangles", "primitives",
                  "bvh_nodes"):
            f = getattr(L, "rsrt_scene_" + n)
            f.restype = C.c_void_p
            f.argtypes = [C.c_void_p]
        L.rsrt_build_bvh.restype = C.c_int
        L.rsrt_alias_table_build.restype = C.c_int
        L.rsrt_synth_environment.restype = C.c_int
        L.rsrt_load_hdr.restype = C.c_int
        L.rsrt_load_hdr.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
        L.rsrt_free.argtypes = [C.c_void_p]
        L.rsrt_camera_deserialize.restype = C.c_int
        L.rsrt_camera_deserialize.argtypes = [C.c_char_p, C.c_void_p, C.c_char_p, C.c_size_t]
        L.rsrt_camera_serialize.argtypes = [C.c_void_p, C.c_char_p]
        L.rsrt_write_png.restype = C.c_int
        L.rsrt_write_png.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.rsrt_write_pfm.restype = C.c_int
        L.rsrt_write_pfm.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
        L.rsrt_display_srgb8_host.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


class SceneError(Exception):
    """anyhow::Error of Scene::load_toml (same message text)."""


class Scene:
    """`Scene` (src/scene.rs:224-231) after `State::new` has derived its uploads from it: the eight
    storage-buffer arrays, the plane descriptors and the camera."""

    _COUNTS = ["n_materials", "n_spheres", "n_planes", "n_vertices", "n_normals", "n_triangles", "n_primitives",
               "n_bvh_nodes", "bvh_depth"]

    def __init__(self, materials, spheres, plane_descs, vertices, normals, triangles, camera_desc,
                 planes=None, primitives=None, bvh_nodes=None, bvh_depth=None):
        self.materials = np.ascontiguousarray(materials).view(T.MATERIAL).reshape(-1)
        self.spheres = np.ascontiguousarray(spheres).view(T.SPHERE).reshape(-1)
        self.plane_descs = np.ascontiguousarray(plane_descs).view(T.PLANE_DESC).reshape(-1)
        self.vertices = np.ascontiguousarray(vertices).view(T.VEC3).reshape(-1)
        self.normals = np.ascontiguousarray(normals).view(T.VEC3).reshape(-1)
        self.triangles = np.ascontiguousarray(triangles).view(T.TRIANGLE).reshape(-1)
        self.camera_desc = np.ascontiguousarray(camera_desc).view(T.CAMERA_DESC).reshape(1)
        self.planes = planes if planes is not None else plane_to_uniform(self.plane_descs)
        if primitives is None:
            primitives, bvh_nodes, bvh_depth = build_bvh(self.spheres, self.plane_descs, self.vertices, self.triangles)
        self.primitives, self.bvh_nodes, self.bvh_depth = primitives, bvh_nodes, bvh_depth

    @classmethod
    def load_toml(cls, path):
        L = lib()
        h = C.c_void_p()
        err = C.create_string_buffer(4096)
        rc = L.rsrt_scene_load_toml(os.fsencode(path), C.byref(h), err, len(err))
        if rc != 0:
            raise SceneError(err.value.decode("utf-8", "replace"))
        try:
            counts = np.zeros(len(cls._COUNTS), np.uint32)
            L.rsrt_scene_get_counts(h, _p(counts))
            c = dict(zip(cls._COUNTS, (int(x) for x in counts)))

            def grab(name, dt, n):
                ptr = getattr(L, "rsrt_scene_" + name)(h)
                if n == 0:
                    return np.zeros(0, dt)
                return np.frombuffer(bytearray(C.string_at(ptr, n * dt.itemsize)), dt)  # keeps padding bytes

            cam = np.zeros(1, T.CAMERA_DESC)
            L.rsrt_scene_get_camera(h, _p(cam))
            return cls(grab("materials", T.MATERIAL, c["n_materials"]), grab("spheres", T.SPHERE, c["n_spheres"]),
                       grab("plane_descs", T.PLANE_DESC, c["n_planes"]), grab("vertices", T.VEC3, c["n_vertices"]),
                       grab("normals", T.VEC3, c["n_normals"]), grab("triangles", T.TRIANGLE, c["n_triangles"]), cam,
                       planes=grab("planes", T.PLANE, c["n_planes"]),
                       primitives=grab("primitives", T.PRIMITIVE_INFO, c["n_primitives"]),
                       bvh_nodes=grab("bvh_nodes", T.BVH_NODE, c["n_bvh_nodes"]), bvh_depth=c["bvh_depth"])
        finally:
            L.rsrt_scene_free(h)

    def camera_uniform(self):
        return camera_uniform(self.camera_desc)


def build_bvh(spheres, plane_descs, vertices, triangles):
    """build_bvh(&Scene) -> (Vec<PrimitiveInfoUniform>, Vec<BvhNodeUniform>) (+ tree depth)."""
    n = len(spheres) + len(plane_descs) + len(triangles)
    prims = np.zeros(n, T.PRIMITIVE_INFO)
    nodes = np.zeros(max(2 * n, 1), T.BVH_NODE)
    n_nodes, depth = C.c_uint32(0), C.c_uint32(0)
    rc = lib().rsrt_build_bvh(_p(spheres), C.c_uint32(len(spheres)), _p(plane_descs), C.c_uint32(len(plane_descs)),
                              _p(vertices), C.c_uint32(len(vertices)), _p(triangles), C.c_uint32(len(triangles)),
                              _p(prims), _p(nodes), C.byref(n_nodes), C.byref(depth))
    if rc != 0:
        raise ValueError("rsrt_build_bvh failed (%d): %s" % (rc, "empty scene" if rc == 1 else "vertex index out of range"))
    return prims, nodes[:n_nodes.value].copy(), depth.value


def plane_to_uniform(plane_descs):
    plane_descs = np.ascontiguousarray(plane_descs).view(T.PLANE_DESC).reshape(-1)
    out = np.zeros(len(plane_descs), T.PLANE)
    for i in range(len(plane_descs)):
        lib().rsrt_plane_to_uniform(C.c_void_p(plane_descs.ctypes.data + i * T.PLANE_DESC.itemsize),
                                    C.c_void_p(out.ctypes.data + i * T.PLANE.itemsize))
    return out


def camera_uniform(camera_desc):
    camera_desc = np.ascontiguousarray(camera_desc).view(T.CAMERA_DESC).reshape(1)
    out = np.zeros(1, T.CAMERA)
    lib().rsrt_camera_uniform(_p(camera_desc), _p(out))
    return out


def make_camera_desc(pos, yaw=0.0, pitch=0.0, fov_y=1.0):
    d = np.zeros(1, T.CAMERA_DESC)
    d["pos"][0] = pos
    d["yaw"], d["pitch"], d["fov_y"] = yaw, pitch, fov_y
    return d


class AliasTable:
    @staticmethod
    def build_by_luminance(rgb):
        """rgb: [H, W, 3] float32 -> (entries[W*H] ALIAS_ENTRY, leftover count)."""
        rgb = np.ascontiguousarray(rgb, dtype=np.float32)
        assert rgb.ndim == 3 and rgb.shape[2] == 3
        h, w = rgb.shape[:2]
        out = np.zeros(w * h, T.ALIAS_ENTRY)
        left = C.c_uint32(0)
        rc = lib().rsrt_alias_table_build(C.c_uint32(w), C.c_uint32(h), _p(rgb), _p(out), C.byref(left))
        if rc != 0:
            raise ValueError("rsrt_alias_table_build failed")
        return out, left.value


def synth_environment(width, height):
    """Deterministic stand-in for the missing HDRIs: [H, W, 4] float32, alpha 0."""
    out = np.zeros((height, width, 4), np.float32)
    if lib().rsrt_synth_environment(C.c_uint32(width), C.c_uint32(height), _p(out)) != 0:
        raise ValueError("rsrt_synth_environment failed")
    return out


class Environment:
    """One HDRI + its alias table, as `EnvironmentMaps::new` prepares it (src/environments.rs:19-64)."""

    def __init__(self, rgba, alias=None):
        self.rgba = np.ascontiguousarray(rgba, dtype=np.float32)
        self.height, self.width = self.rgba.shape[:2]
        if alias is None:
            alias, self.leftover = AliasTable.build_by_luminance(self.rgba[:, :, :3])
        self.alias = alias

    @classmethod
    def synthetic(cls, width=2048, height=1024):
        return cls(synth_environment(width, height))


    @classmethod
    def load_hdr(cls, path):
        """A Radiance .hdr as the reference's `image::load_from_memory(..).into_rgb32f()` decodes it."""
        return cls(load_hdr(path))


def load_hdr(path):
    """-> [H, W, 4] float32 RGBA (alpha 0, as src/texture.rs:112-115 writes it)."""
    w, h, ptr = C.c_uint32(0), C.c_uint32(0), C.c_void_p()
    err = C.create_string_buffer(512)
    rc = lib().rsrt_load_hdr(os.fsencode(path), C.byref(w), C.byref(h), C.byref(ptr), err, len(err))
    if rc != 0:
        raise ValueError("rsrt_load_hdr: " + err.value.decode())
    try:
        rgb = np.frombuffer(C.string_at(ptr, w.value * h.value * 12), np.float32).reshape(h.value, w.value, 3)
    finally:
        lib().rsrt_free(ptr)
    rgba = np.zeros((h.value, w.value, 4), np.float32)
    rgba[..., :3] = rgb
    return rgba


def camera_serialize(camera_desc):
    """Camera::serialize (src/camera.rs:30-49): the string the `p` key prints and --state takes."""
    camera_desc = np.ascontiguousarray(camera_desc).view(T.CAMERA_DESC).reshape(1)
    out = C.create_string_buffer(33)
    lib().rsrt_camera_serialize(_p(camera_desc), out)
    return out.value.decode()


def camera_deserialize(encoded):
    """Camera::deserialize (src/camera.rs:51-89); raises ValueError with the reference's message."""
    out = np.zeros(1, T.CAMERA_DESC)
    err = C.create_string_buffer(256)
    if lib().rsrt_camera_deserialize(encoded.encode(), _p(out), err, len(err)) != 0:
        raise ValueError(err.value.decode())
    return out


def display_srgb8(sum_rgba, sample_total):
    """hdr.wgsl display pass on the CPU: [H, W, 4] f32 sums -> [H, W, 4] uint8."""
    sum_rgba = np.ascontiguousarray(sum_rgba, np.float32)
    out = np.empty(sum_rgba.shape[:2] + (4,), np.uint8)
    lib().rsrt_display_srgb8_host(_p(sum_rgba), sum_rgba.shape[0] * sum_rgba.shape[1], sample_total, _p(out))
    return out


def write_png(path, rgba8):
    rgba8 = np.ascontiguousarray(rgba8, np.uint8)
    assert rgba8.ndim == 3 and rgba8.shape[2] == 4
    if lib().rsrt_write_png(os.fsencode(path), rgba8.shape[1], rgba8.shape[0], _p(rgba8)) != 0:
        raise OSError("rsrt_write_png failed: " + str(path))


def write_pfm(path, rgb):
    rgb = np.ascontiguousarray(rgb, np.float32)
    assert rgb.ndim == 3 and rgb.shape[2] >= 3
    if lib().rsrt_write_pfm(os.fsencode(path), rgb.shape[1], rgb.shape[0], _p(rgb), rgb.shape[2]) != 0:
        raise OSError("rsrt_write_pfm failed: " + str(path))
