"""rsoderh-raytracing_amd — MI355X-native path-tracing integrator behind the reference's
State/Scene surface.

    host.py    Scene.load_toml / build_bvh / AliasTable / uniforms  (librsrt_host.so, C++)
    state.py   State: upload + render, the mirror of src/state.rs    (librsrt.so, HIP gfx950)
    csrc/      the native sources; include/ at the repo root declares the C-ABI

There is no CPU fallback: `State` raises if librsrt.so or a GPU is missing.
"""
import os as _os

# pipelined single-sample calls (state.State.render, the reference's interactive mode) want four kernels of a context resident at
# once, each on a stream of its own; HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default) and those that share
# one run one after the other.  Counts only before the process's first HIP call, so this HOST-side package asks for 8 as early as
# it can, unless the variable is already set or RSRT_KEEP_HW_QUEUES=1 says to leave the process environment alone.  (librsrt.so
# itself never touches the environment: INTEGRATION.md §4.)
if _os.environ.get("RSRT_KEEP_HW_QUEUES", "0") in ("", "0"):
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import state, types  # noqa: F401,E402
from .host import AliasTable, Environment, Scene, SceneError, build_bvh, camera_uniform, plane_to_uniform  # noqa: F401

__all__ = ["types", "Scene", "SceneError", "Environment", "AliasTable", "build_bvh", "camera_uniform",
           "plane_to_uniform", "State"]
from .state import RsrtError, State  # noqa: E402,F401  (librsrt.so itself is loaded on first use)
