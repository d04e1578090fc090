"""The C-ABI libraries load and export every symbol the headers declare (no GPU needed)."""
import ctypes as C
import os
import re

import pytest

import util
from rsoderh_raytracing_amd import _build


def declared(header):
    text = open(os.path.join(util.ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rsrt_[a-z0-9_]+)\s*\(", text)))


def test_host_library_exports_every_declared_symbol():
    lib = C.CDLL(_build.build_host())
    names = declared("rsrt_host.h")
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), n


def test_device_library_builds_for_gfx950_and_exports_every_declared_symbol():
    lib = C.CDLL(_build.build_hip())
    names = declared("rsrt.h")
    assert len(names) >= 16
    for n in names:
        assert hasattr(lib, n), n


def test_device_library_contains_gfx950_code_object():
    data = open(_build.build_hip(), "rb").read()
    assert b"gfx950" in data


def test_no_cpu_fallback_without_gpu():
    """Without a GPU, context creation fails loudly (the product has no CPU path)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from rsoderh_raytracing_amd.state import RsrtError, State
    with pytest.raises(RsrtError, match="no HIP device|not available"):
        State()


def test_product_never_references_the_oracle():
    pkg = os.path.join(util.ROOT, "rsoderh-raytracing_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                text = open(os.path.join(root, f), errors="replace").read()
                assert "oracle" not in text.lower() or f == "README.md", os.path.join(root, f)
