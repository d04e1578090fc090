"""In-tree native builds: librsrt_host.so (g++, CPU preprocessing) and librsrt.so (hipcc, gfx950)."""
import contextlib
import fcntl
import hashlib
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


def _knob_flags():
    """Experiment knobs (tools/knob_sweep.py, tools/flag_sweep.sh): extra compiler flags / macro overrides."""
    flags = []
    if os.environ.get("RSRT_HIPCC_FLAGS"):  # extra compiler flags
        flags += os.environ["RSRT_HIPCC_FLAGS"].split()
    if os.environ.get("RSRT_WPS"):  # waves per SIMD the pool kernel is compiled for
        flags.append("-DRT_POOL_WAVES_PER_SIMD=" + os.environ["RSRT_WPS"])
    if os.environ.get("RSRT_LEAFQ"):  # leaves a lane holds before the wave tests primitives
        flags.append("-DRT_LEAFQ=" + os.environ["RSRT_LEAFQ"])
    return flags


def source_id():
    """sha256 (16 hex digits) over every file that goes into librsrt.so: compiled in as RSRT_BUILD_ID
    (rsrt_build_id()) and written into the rocprofv3 summaries under profiles/, so that bench.py can tell
    whether a committed profile was taken on the kernel it is timing."""
    h = hashlib.sha256()
    for f in sorted(_deps("hip")):
        if f.endswith((".h", ".hip", ".hpp")):
            h.update(os.path.basename(f).encode())
            with open(f, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def build_hip(force=False, extra_flags=(), instrument=False):
    """librsrt.so; instrument=True builds the diagnostic twin librsrt_instr.so (-DRT_INSTRUMENT).
    With an experiment knob set the build goes to librsrt_exp_<hash of the flags>.so instead: the
    product library is never overwritten by an experiment."""
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    target = HIP_LIB.replace(".so", "_instr.so") if instrument else HIP_LIB
    flags = list(extra_flags) + (["-DRT_INSTRUMENT"] if instrument else [])
    knobs = _knob_flags()
    if knobs:
        flags += knobs
        target = target.replace(".so", "_exp_%s.so" % hashlib.sha256(" ".join(knobs).encode()).hexdigest()[:10])
    sid = source_id() + ("+" + "+".join(k[2:] if k.startswith("-D") else k.lstrip("-") for k in knobs) if knobs else "") + ("+instr" if instrument else "")
    flags.append('-DRSRT_BUILD_ID="%s"' % sid)
    with _locked():
        if force or _newer(target, _deps("hip")):
            tmp = target + ".tmp%d" % os.getpid()
            _run([hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                  "-fno-fast-math", "-fno-slp-vectorize", "-fno-vectorize", "-Wall", "-Wextra", "-Wno-unused-parameter", "-I", INCLUDE, "-o", tmp] + flags + srcs)
            os.replace(tmp, target)
    return target


def build_all(force=False):
    return build_host(force), build_hip(force)
