"""In-tree native builds: librsrt_host.so (g++, CPU preprocessing) and librsrt.so (hipcc, gfx950)."""
import contextlib
import fcntl
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(os.path.dirname(PKG), "include")

HOST_SOURCES = ["host/bvh_build.cpp", "host/preprocess.cpp", "host/scene_load.cpp", "host/image_io.cpp"]
HIP_SOURCES = ["hip/rsrt_api.hip"]
HOST_LIB = os.path.join(PKG, "librsrt_host.so")
HIP_LIB = os.path.join(PKG, "librsrt.so")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _deps(subdir):
    out = [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE)]
    d = os.path.join(CSRC, subdir)
    out += [os.path.join(d, f) for f in os.listdir(d)]
    return out


@contextlib.contextmanager
def _locked():
    """Serialises builds between processes (N bench ranks import the package at the same moment)."""
    with open(os.path.join(PKG, ".build.lock"), "w") as f:
        fcntl.flock(f, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(f, fcntl.LOCK_UN)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def build_host(force=False):
    srcs = [os.path.join(CSRC, s) for s in HOST_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    with _locked():
        if force or _newer(HOST_LIB, _deps("host")):
            tmp = HOST_LIB + ".tmp%d" % os.getpid()
            _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall", "-Wextra", "-o", tmp] + srcs)
            os.replace(tmp, HOST_LIB)
    return HOST_LIB


def hipcc_path():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def build_hip(force=False, extra_flags=(), instrument=False):
    """librsrt.so; instrument=True builds the diagnostic twin librsrt_instr.so (-DRT_INSTRUMENT)."""
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    target = HIP_LIB.replace(".so", "_instr.so") if instrument else HIP_LIB
    flags = list(extra_flags) + (["-DRT_INSTRUMENT"] if instrument else [])
    if os.environ.get("RSRT_HIPCC_FLAGS"):  # experiment knob: extra compiler flags
        flags += os.environ["RSRT_HIPCC_FLAGS"].split()
        force = True
    if os.environ.get("RSRT_WPS"):  # experiment knob: waves per SIMD the pool kernel is compiled for
        flags.append("-DRT_POOL_WAVES_PER_SIMD=" + os.environ["RSRT_WPS"])
        force = True
    if os.environ.get("RSRT_LEAFQ"):  # experiment knob: leaves a lane holds before the wave tests primitives
        flags.append("-DRT_LEAFQ=" + os.environ["RSRT_LEAFQ"])
        force = True
    with _locked():
        if force or _newer(target, _deps("hip")):
            tmp = target + ".tmp%d" % os.getpid()
            _run([hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                  "-fno-fast-math", "-fno-slp-vectorize", "-fno-vectorize", "-Wall", "-Wextra", "-Wno-unused-parameter", "-I", INCLUDE, "-o", tmp] + flags + srcs)
            os.replace(tmp, target)
    return target


def build_all(force=False):
    return build_host(force), build_hip(force)
