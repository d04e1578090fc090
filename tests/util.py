"""Shared helpers of the test-suite: product <-> oracle array views, small environments, metrics."""
import os

import numpy as np

import oracle
import rsoderh_raytracing_amd as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASSETS = os.path.join(ROOT, "tests", "golden", "assets")


def scene_path(name):
    return os.path.join(ASSETS, "scenes", name + ".toml")


def oracle_scene(scene):
    """oracle.Scene over the product Scene's arrays (identical encase layouts)."""
    return oracle.Scene(materials=scene.materials.view(oracle.MATERIAL), spheres=scene.spheres.view(oracle.SPHERE),
                        planes=scene.planes.view(oracle.PLANE), vertices=scene.vertices.view(oracle.VEC3),
                        normals=scene.normals.view(oracle.VEC3), triangles=scene.triangles.view(oracle.TRIANGLE),
                        prims=scene.primitives.view(oracle.PRIM_INFO), nodes=scene.bvh_nodes.view(oracle.BVH_NODE))


def oracle_env(env):
    return oracle.Env(env.rgba, env.alias.view(oracle.ALIAS_ENTRY))


def fields_equal(a, b):
    return len(a) == len(b) and all(np.array_equal(a[n], b[m]) for n, m in zip(a.dtype.names, b.dtype.names))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def rmse_per_channel(a, b, spp):
    d = (a[..., :3].astype(np.float64) - b[..., :3].astype(np.float64)) / spp
    return np.sqrt((d * d).mean(axis=(0, 1)))


_env_cache = {}


def small_env(w=64, h=32):
    if (w, h) not in _env_cache:
        _env_cache[(w, h)] = R.Environment.synthetic(w, h)
    return _env_cache[(w, h)]
