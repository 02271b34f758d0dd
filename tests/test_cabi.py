"""CPU-side checks of the C-ABI library: it builds, loads, and exports every symbol the header
declares.  No compute call is made without a GPU."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "femo_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(femo_[a-z_A-Z0-9]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    from femo_alpha_amd import _build, _lib
    _build.build()
    lib = _lib.load()
    syms = _header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"libfemo_hip.so does not export {s}"
    assert sorted(_lib.SIGNATURES) == syms, "ctypes SIGNATURES and include/femo_hip.h disagree"
    assert lib.femo_version() >= 100


def test_symbolic_library_builds_and_exports_every_declared_symbol():
    from femo_alpha_amd import _build
    from femo_alpha_amd.solver import _native
    _build.build_symbolic()
    lib = _native.load()
    text = open(os.path.join(ROOT, "include", "femo_symbolic.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    syms = sorted(set(re.findall(r"\b(femo_plan_[a-z_A-Z0-9]+)\s*\(", text)))
    assert len(syms) == 7
    for s in syms:
        assert hasattr(lib, s), f"libfemo_symbolic.so does not export {s}"
    assert sorted(_native.SIGNATURES) == syms, "ctypes SIGNATURES and include/femo_symbolic.h disagree"


def test_product_path_fails_loudly_without_a_gpu():
    from femo_alpha_amd import _lib
    from femo_alpha_amd.backend import ShellContext
    from femo_alpha_amd.mesh import plate_mesh
    if _lib.load().femo_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.FemoHipError, match="no CPU fallback"):
        ShellContext(plate_mesh(2.0, 10.0, 2, 4))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "femo_alpha_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
