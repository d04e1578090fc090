"""The oracle against its committed golden vectors (frozen outputs; see tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

import oracle
import util
import rsoderh_raytracing_amd as R

SCENES = ["house", "default", "cube", "suzanne", "spheres_only"]


def golden(name):
    return np.load(os.path.join(util.ROOT, "tests", "golden", "scene_%s.npz" % name))


def golden_env():
    g = np.load(os.path.join(util.ROOT, "tests", "golden", "env_64x32.npz"))
    return R.Environment(g["rgba"], g["alias"].view(R.types.ALIAS_ENTRY).reshape(-1))


@pytest.mark.parametrize("name", SCENES)
def test_oracle_images_match_golden(name):
    g = golden(name)
    sc = R.Scene.load_toml(util.scene_path(name))
    osc, cam, env = util.oracle_scene(sc), sc.camera_uniform().view(oracle.CAMERA), util.oracle_env(golden_env())
    for key in [k for k in g.files if k.startswith("sum_")]:
        spp, mb = int(key.split("_")[1][:-3]), int(key.split("_")[2][:-1])
        img, st = oracle.render(osc, env, cam, 64, 64, 0, spp, mb)
        assert np.array_equal(util.bits(img), util.bits(g[key])), key
        assert [st["paths"], st["ext_rays"], st["shadow_rays"]] == list(g["rays_%dspp_%db" % (spp, mb)])
        assert np.all(img[..., 3] == 1.0) and np.isfinite(img).all()


@pytest.mark.parametrize("name", SCENES)
def test_oracle_ray_batch_matches_golden(name):
    g = golden(name)
    osc = util.oracle_scene(R.Scene.load_toml(util.scene_path(name)))
    for mode, key in [(0, "hits"), (1, "hits_bvh")]:
        h = oracle.cast_rays(osc, g["ray_o"], g["ray_d"], mode, 0)
        assert np.array_equal(h.view(np.uint32).reshape(-1, 9), g[key])


def test_fast_build_is_bit_identical_to_strict_build():
    """bench.py times liboracle_fast.so (-O3); it must produce the strict build's bits."""
    sc = R.Scene.load_toml(util.scene_path("house"))
    osc, cam, env = util.oracle_scene(sc), sc.camera_uniform().view(oracle.CAMERA), util.oracle_env(golden_env())
    a, _ = oracle.render(osc, env, cam, 64, 64, 0, 4, 8)
    b, _ = oracle.render(osc, env, cam, 64, 64, 0, 4, 8, fast=True)
    assert np.array_equal(util.bits(a), util.bits(b))
