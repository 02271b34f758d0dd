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
    assert len(syms) == 8
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


@pytest.mark.parametrize("nvc,nquad,nred,cg1,cr", [(4, 3, 0, 0, 0), (4, 4, 0, 0, 0), (4, 5, 0, 0, 0), (4, 6, 0, 0, 0), (4, 4, 2, 0, 0), (4, 5, 0, 1, 0),
                                                   (3, 4, 0, 0, 0), (3, 6, 0, 0, 0), (3, 9, 0, 0, 0), (3, 12, 0, 0, 0), (3, 9, 0, 1, 0),
                                                   (3, 6, 0, 0, 1), (3, 9, 0, 0, 1)])
def test_quadrature_tables_are_bit_identical_to_the_oracle(nvc, nquad, nred, cg1, cr):
    """The quadrature table is part of the discrete problem (a last-place change of one weight moves the 1 M-DOF answers by 3e-7,
    DESIGN.md section 2): the library's tables -- built on the host, no device needed -- and the oracle's are the same doubles, for the
    Gauss rules of the quadrilaterals and for the symmetric rules of degree 4 / 6 / 9 / 12 of the triangles (scripts/derive_triangle_rules.py)."""
    import ctypes as C

    import numpy as np

    from femo_alpha_amd import _lib
    from femo_alpha_amd.mesh import ShellMesh
    from oracle.rm_shell_oracle import ShellOracle
    lib = _lib.load()
    Q = 36
    w, wS = np.zeros(Q), np.zeros(Q)
    N2, dN2, N1, dN1, NR, dNR = np.zeros((Q, 9)), np.zeros((Q, 9, 2)), np.zeros((Q, 4)), np.zeros((Q, 4, 2)), np.zeros((Q, 4)), np.zeros((Q, 4, 2))
    nq = C.c_int32()
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    assert lib.femo_quadrature_tables(nvc, nquad, nred, cg1, cr, C.byref(nq), p(w), p(wS), p(N2), p(dN2), p(N1), p(dN1), p(NR), p(dNR)) == 0
    nq = nq.value
    if nvc == 4:
        nodes = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0.0]]); cells = np.array([[0, 1, 2, 3]])
    else:
        nodes = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0.0]]); cells = np.array([[0, 1, 2]])
    m = ShellMesh(nodes, cells, element="CG1CG1" if cg1 else ("CG2CR1" if cr else "CG2CG1"))
    o = ShellOracle(m, nquad=nquad, nred=nred)
    assert o.nq == nq
    npc = o.N2.shape[1]
    assert np.array_equal(o.wts, w[:nq]) and np.array_equal(o.wts_strain, wS[:nq])
    assert np.array_equal(o.N2, N2[:nq, :npc]) and np.array_equal(o.dN2, dN2[:nq, :npc])
    assert np.array_equal(o.N1, N1[:nq, :nvc]) and np.array_equal(o.dN1, dN1[:nq, :nvc])
    assert np.array_equal(o.NR, NR[:nq, :nvc]) and np.array_equal(o.dNR, dNR[:nq, :nvc])
    if nvc == 3:
        assert abs(w[:nq].sum() - 0.5) < 1e-16          # rounds 1-5: 15-digit literals, 0.5 + 1e-15
    assert lib.femo_quadrature_tables(3, 5, 0, 0, 0, None, *([None] * 8)) == 2       # no rule of that degree


def test_triangle_rules_integrate_their_degree_exactly():
    """Every monomial up to the rule's degree, against the exact integral a! b! / (a + b + 2)! over the unit triangle."""
    from math import factorial

    import numpy as np

    from oracle.rm_shell_oracle import TRI_DEGREES, tri_rule
    assert TRI_DEGREES == (4, 6, 9, 12)
    for deg, npts in zip(TRI_DEGREES, (6, 12, 19, 33)):
        pts, wts = tri_rule(deg)
        assert pts.shape == (npts, 2) and np.all(wts > 0) and np.all(pts > 0) and np.all(pts.sum(axis=1) < 1)
        for a in range(deg + 1):
            for b in range(deg + 1 - a):
                exact = factorial(a) * factorial(b) / factorial(a + b + 2)
                assert abs(np.sum(wts * pts[:, 0] ** a * pts[:, 1] ** b) - exact) < 3e-17 + 1e-15 * exact, (deg, a, b)
        a = deg + 1                                       # and not one degree more
        assert abs(np.sum(wts * pts[:, 0] ** a) - factorial(a) / factorial(a + 2)) > 1e-12
    with pytest.raises(ValueError, match="degree"):
        tri_rule(5)
