"""rsoderh-raytracing_amd — MI355X-native path-tracing integrator behind the reference's
State/Scene surface.

    host.py    Scene.load_toml / build_bvh / AliasTable / uniforms  (librsrt_host.so, C++)
    state.py   State: upload + render, the mirror of src/state.rs    (librsrt.so, HIP gfx950)
    csrc/      the native sources; include/ at the repo root declares the C-ABI

There is no CPU fallback: `State` raises if librsrt.so or a GPU is missing.
"""
from . import state, types  # noqa: F401
from .host import AliasTable, Environment, Scene, SceneError, build_bvh, camera_uniform, plane_to_uniform  # noqa: F401

__all__ = ["types", "Scene", "SceneError", "Environment", "AliasTable", "build_bvh", "camera_uniform",
           "plane_to_uniform", "State"]
from .state import RsrtError, State  # noqa: E402,F401  (librsrt.so itself is loaded on first use)
