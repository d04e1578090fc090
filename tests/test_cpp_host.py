"""A host written purely against the C/C++ boundary (include/rsrt_state.hpp, tests/cpp/state_demo.cpp)."""
import os
import subprocess

import numpy as np
import pytest
from PIL import Image

import util
import rsoderh_raytracing_amd as R
from rsoderh_raytracing_amd import _build


def build_demo(tmp_path):
    exe = str(tmp_path / "state_demo")
    pkg = os.path.join(util.ROOT, "rsoderh-raytracing_amd")
    _build.build_host()
    _build.build_hip()
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-I", os.path.join(util.ROOT, "include"),
           os.path.join(util.ROOT, "tests", "cpp", "state_demo.cpp"), "-o", exe, "-L", pkg, "-lrsrt", "-lrsrt_host",
           "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return exe


def test_cpp_host_compiles_against_the_headers(tmp_path):
    build_demo(tmp_path)


def test_cpp_host_reports_errors_without_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    exe = build_demo(tmp_path)
    r = subprocess.run([exe, util.scene_path("house"), "16", "16", "1", "0", "3", "16", "8", str(tmp_path / "o.f32"), str(tmp_path / "o.png")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr
    r = subprocess.run([exe, str(tmp_path / "missing.toml"), "16", "16", "1", "0", "3", "16", "8", "a", "b"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "Couldn't open scene" in r.stderr


@pytest.mark.gpu
def test_cpp_host_renders_the_oracle_image(tmp_path):
    import oracle
    exe = build_demo(tmp_path)
    w, h, frames, batch, mb = 96, 54, 3, 5, 8
    r = subprocess.run([exe, util.scene_path("house"), str(w), str(h), str(frames), str(batch), str(mb), "256", "128",
                        str(tmp_path / "o.f32"), str(tmp_path / "o.png")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    sums = np.fromfile(tmp_path / "o.f32", np.float32).reshape(h, w, 4)
    sc = R.Scene.load_toml(util.scene_path("house"))
    env = R.Environment.synthetic(256, 128)
    ref, st = oracle.render(util.oracle_scene(sc), util.oracle_env(env), sc.camera_uniform().view(oracle.CAMERA), w, h, 0, frames + batch, mb)
    assert np.array_equal(util.bits(sums), util.bits(ref))
    assert "samples %d paths %d rays %d" % (frames + batch, st["paths"], st["ext_rays"] + st["shadow_rays"]) in r.stdout
    png = np.array(Image.open(tmp_path / "o.png"))
    assert np.array_equal(png, R.host.display_srgb8(sums, frames + batch))
