"""The device-vector building blocks of the C ABI (include/femo_hip.h: the calls a host-side time loop or a multi-GPU driver composes
-- operator and solves on vector ids, gradient accumulators, level-range factorisation and sweeps, Schur blocks of a front, raw
device pointers, timers) against the whole-problem entry points and the oracle.  tests/conftest.py's FEMO_CALL_AUDIT showed these
entry points to be the ones the other GPU tests never reach in this process."""
import ctypes as C

import numpy as np
import pytest

from femo_alpha_amd.mesh import plate_mesh, wing_skin_mesh

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-300)


def _context(m, strong=None, beta=None, fields=None):
    from femo_alpha_amd.backend import ShellContext
    c = ShellContext(m)
    rng = np.random.default_rng(3)
    f = dict(thickness=0.05 * (1 + 0.3 * rng.uniform(-1, 1, m.nn)), E=[3e7], nu=[0.3], density=10 * (1 + 0.1 * rng.uniform(-1, 1, m.nn)),
             F_solid=rng.uniform(-1, 1, (m.nn, 3)))
    f.update(fields or {})
    for k, v in f.items():
        c.set_field(k, v)
    if strong is not None:
        c.set_strong_dofs(strong)
    if beta is not None:
        c.set_penalty_facets(m.penalty_facets(lambda x: np.less(x[1], 1e-12)), beta)
    return c, f, rng


def _put(c, name, x):
    import torch
    c.vec_tensor(name).copy_(torch.from_numpy(np.ascontiguousarray(x)))
    c.sync()


def _get(c, name):
    c.sync()
    return c.vec_tensor(name).cpu().numpy()


def test_version_device_count_and_plain_create():
    from femo_alpha_amd import _lib
    lib = _lib.load()
    assert lib.femo_version() >= 100 and lib.femo_device_count() >= 1
    # femo_create (the entry point without ghost cells; the package itself always goes through femo_create_ghost)
    m = plate_mesh(1.0, 2.0, 3, 4)
    h = C.c_void_p()
    nodes = np.ascontiguousarray(m.nodes, dtype=np.float64)
    cells = np.ascontiguousarray(m.cells, dtype=np.int32)
    cp2 = np.ascontiguousarray(m.cell_p2, dtype=np.int32)
    rc = lib.femo_create(C.byref(h), 0, m.nn, m.nel, m.nvc, m.nP2, _lib.dptr(nodes.ravel()), _lib.iptr(cells.ravel()), _lib.iptr(cp2.ravel()),
                         0, 0, 4)
    assert rc == 0, lib.femo_last_error(None)
    assert lib.femo_ndof(h) == m.ndof
    lib.femo_destroy(h)


@pytest.mark.parametrize("variant,uhat", [("CG2CG1", False), ("CG2CG1", True), ("CG1CG1", False)])
def test_operator_with_four_and_five_lanes_per_element(variant, uhat):
    """Option apply_lanes: the matrix-free operator (k_apply4) with the quadrature points of an element dealt to four lanes (a DPP quad;
    the 25 points of the 5 x 5 rule are 7 + 6 + 6 + 6; the default) or to five (5 each, twelve elements on 60 lanes of a wave, partial
    results through ds_bpermute; measured slower, profiles/r6_apply_lanes.txt).  Both layouts against the oracle's matrix at 1e-11, each other at 1e-13, with
    and without mesh motion, the stiffness alone and the step operator of the transient path (aK K + aM M: the inertia loop of the kernel),
    on an element count that leaves the last wave and the last block partly empty; other rules (here 4 x 4) accept five lanes as well."""
    from femo_alpha_amd.backend import ShellContext
    from femo_alpha_amd.mesh import ShellMesh
    from femo_alpha_amd._lib import FemoHipError
    from oracle.rm_shell_oracle import ShellOracle
    base = wing_skin_mesh(7, 11, shuffle=True)              # 77 cells: 48 + 29 (five lanes), 64 + 13 (four)
    m = base if variant == "CG2CG1" else ShellMesh(base.nodes, base.cells, variant)
    rng = np.random.default_rng(11)
    f = dict(thickness=0.05 * (1 + 0.3 * rng.uniform(-1, 1, m.nn)), E=3e7 * (1 + 0.2 * rng.uniform(-1, 1, m.nn)),
             nu=0.3 + 0.05 * rng.uniform(-1, 1, m.nn), density=10 * (1 + 0.1 * rng.uniform(-1, 1, m.nn)), F_solid=rng.uniform(-1, 1, (m.nn, 3)))
    if uhat:
        f["uhat"] = 0.02 * rng.uniform(-1, 1, (m.nn, 3))
    x = rng.uniform(-1, 1, m.ndof)
    for nquad in (5, 4):
        o = ShellOracle(m, nquad=nquad)
        o.set_fields(h=f["thickness"], E=f["E"], nu=f["nu"], rho=f["density"], f=f["F_solid"], uhat=f.get("uhat"))
        Kx, Mx = o.assemble_K() @ x, o.assemble_M() @ x
        c = ShellContext(m, nquad=nquad)
        for k, v in f.items():
            c.set_field(k, v)
        got = {}
        for lanes in (0, 4, 5):
            c.set_option("apply_lanes", lanes)
            y = c.apply_K(x)
            _put(c, "p", x)
            c.op_apply_vec2("p", "Ap", 0.5, 200.0, with_penalty=False)
            got[lanes] = (y, _get(c, "Ap").copy())
            assert rel(y, Kx) < 1e-11 and rel(got[lanes][1], 0.5 * Kx + 200.0 * Mx) < 1e-11, (nquad, lanes)
        assert rel(got[5][0], got[4][0]) < 1e-13 and rel(got[5][1], got[4][1]) < 1e-13
        assert np.array_equal(got[0][0], got[4][0]) and np.array_equal(got[0][1], got[4][1])      # 0: the default, a quad
        with pytest.raises(FemoHipError):
            c.set_option("apply_lanes", 6)
        c.close()


def test_operator_and_solves_on_vector_ids():
    """femo_op_apply_vec / femo_solve_vec / femo_vec_mask_zero / femo_device_ptr / femo_sync against apply_K and solve_linear."""
    m = wing_skin_mesh(8, 20, shuffle=True).renumbered()[0]
    sd = m.locate_dofs_geometrical(lambda x: np.less(x[1], 1e-9))
    c, f, rng = _context(m, strong=sd)
    x = rng.uniform(-1, 1, m.ndof)
    # K_local x without the Dirichlet treatment: the same numbers as a context that has no conditions at all
    c0, _, _ = _context(m)
    _put(c, "p", x)
    c.op_apply_vec("p", "Ap")
    assert rel(_get(c, "Ap"), c0.apply_K(x)) < 1e-12
    c0.close()
    # femo_op_apply_vec2: the same un-masked operator with explicit coefficients, aK K + aM M (the step operator of the transient path)
    Kx = _get(c, "Ap").copy()
    c.op_apply_vec2("p", "Ap", 1.0, 0.0)
    assert rel(_get(c, "Ap"), Kx) < 1e-13
    from oracle.rm_shell_oracle import ShellOracle
    o = ShellOracle(m)
    o.set_fields(h=f["thickness"], E=3e7, nu=0.3, rho=f["density"], f=f["F_solid"])
    c.op_apply_vec2("p", "Ap", 0.5, 200.0)
    assert rel(_get(c, "Ap"), 0.5 * Kx + 200.0 * (o.assemble_M() @ x)) < 1e-11
    free = np.ones(m.ndof, bool); free[sd] = False
    # vec_mask_zero: the constrained entries, nothing else
    _put(c, "z", x)
    c.vec_mask_zero("z")
    z = _get(c, "z")
    assert np.all(z[sd] == 0.0) and np.array_equal(z[free], x[free])
    # solve on vector ids = solve_linear
    c.use_direct_solver(leaf_size=6)
    b = rng.uniform(-1, 1, m.ndof); b[sd] = 0.0
    ref, it_ref, _ = c.solve_linear(b)
    _put(c, "b", b)
    it, rr = c.solve_vec("b", "adjoint")
    assert it == it_ref and rr < 1e-11 and rel(_get(c, "adjoint"), ref) < 1e-12
    from femo_alpha_amd._lib import FemoHipError
    with pytest.raises(FemoHipError):                 # vectors 2..5 (r, z, p, Ap) are the solver's work space
        c.solve_vec("b", "z")
    # raw pointers: the state vector is vector id 0; fields have their own buffers; unknown names give null
    lib = c.lib
    lib.femo_device_ptr.restype = C.c_void_p
    lib.femo_vec_ptr.restype = C.c_void_p
    assert lib.femo_device_ptr(c._h, b"state") == lib.femo_vec_ptr(c._h, 0)
    assert lib.femo_device_ptr(c._h, b"thickness") and not lib.femo_device_ptr(c._h, b"no_such_buffer")
    t = c.last_timing()
    assert t["total_ms"] > 0 and t["operator_launches"] >= 1
    assert 0 < c.bench_kernel("apply", 5) < 50.0
    c.close()


def test_gradient_accumulator_and_field_gradient_on_vector_ids():
    """femo_grad_reset / _add / _get: y^T (dK/dh) x and y^T (dM/dh) x; femo_field_gradient_vec: the adjoint formula on vector ids."""
    m = wing_skin_mesh(6, 14, shuffle=True)
    c, f, rng = _context(m, beta=1e6)
    w = rng.uniform(-1, 1, m.ndof) * 1e-3
    lam = rng.uniform(-1, 1, m.ndof)
    c.set_state(w)
    _put(c, "adjoint", lam)
    c.grad_reset()
    c.grad_add("K", "state", "adjoint", 1.0)
    gK = c.grad_get()
    assert rel(gK, c.dRdarg_T("thickness", lam)) < 1e-12            # the load does not depend on the thickness
    c.grad_add("K", "state", "adjoint", -0.25)                       # accumulates
    assert rel(c.grad_get(), 0.75 * gK) < 1e-12
    # inertia: central differences of lam^T M(h) w through the operator with aK = 0, aM = 1
    c.grad_reset()
    c.grad_add("M", "state", "adjoint", 1.0)
    gM = c.grad_get()
    h0 = np.asarray(f["thickness"], dtype=float)

    def lMw(h):
        c.set_field("thickness", h)
        _put(c, "p", w)
        c.op_apply_vec2("p", "Ap", 0.0, 1.0, with_penalty=False)
        return float(lam @ _get(c, "Ap"))
    for i in rng.choice(m.nn, 4, replace=False):
        st = 1e-5 * h0[i]
        hp = h0.copy(); hp[i] += st; hm = h0.copy(); hm[i] -= st
        fd = (lMw(hp) - lMw(hm)) / (2 * st)
        assert abs(gM[i] - fd) <= 1e-6 * np.abs(gM).max() + 1e-8 * abs(fd), (i, gM[i], fd)
    c.set_field("thickness", h0)
    # d functional / d arg - (dR / d arg)^T lambda with lambda taken from a vector id
    c.set_state(w)
    _put(c, "adjoint", lam)
    for arg in ("thickness", "E", "nu"):
        want = c.dfunctional("compliance", arg) - c.dRdarg_T(arg, lam)
        assert rel(c.field_gradient_vec("compliance", arg, "adjoint"), want) < 1e-12
    c.close()


@pytest.mark.parametrize("assemble_fc", [2, 0])           # both front assemblies: one workgroup per leaf front | one wave per element + atomics
def test_level_ranges_sweeps_and_schur_blocks(assemble_fc):
    """femo_factorize_range + femo_frontal_sweep compose to the preconditioner; the Schur block femo_front_schur_get hands out is the
    Schur complement of the subtree's own stiffness onto its boundary (dense algebra on the oracle's element matrices);
    femo_front_block_set refuses a front that has pivots; the instrumented factorisation and sweep profiles report every level."""
    import scipy.sparse as sp
    import torch
    from femo_alpha_amd._lib import FemoHipError
    from oracle.rm_shell_oracle import ShellOracle
    m = wing_skin_mesh(8, 20, shuffle=True).renumbered()[0]
    sd = m.locate_dofs_geometrical(lambda x: np.less(x[1], 1e-9))
    c, f, rng = _context(m, strong=sd)
    c.set_option("assemble_fc", assemble_fc)
    plan = c.enable_frontal(6)
    c.set_solver(preconditioner=2, rtol=1e-12, maxit=30, check_every=1)
    L = plan.nlevels
    c.factorize_range(0, L - 1, True)                    # everything below the root
    o = ShellOracle(m, strong_dofs=sd)
    o.set_fields(h=f["thickness"], E=3e7, nu=0.3, rho=f["density"], f=f["F_solid"])
    K = o.assemble_K().tocsr()                           # constrained rows and columns emptied, unit diagonal
    masked = np.zeros(m.ndof, bool); masked[o.strong_dofs] = True
    root = int(plan.level_nodes[L - 1][0])
    cd = m.cell_dofs()
    Ke = o.element_matrices()
    # cells of each child's subtree: climb from every cell's leaf front
    top = np.asarray(plan.elem_front).copy()
    kids = (int(plan.left[root]), int(plan.right[root]))
    for _ in range(L):
        up = plan.parent[top]
        top = np.where((up >= 0) & ~np.isin(top, kids), up, top)
    for t in kids:
        cells = np.nonzero(top == t)[0]
        dofs = plan.front_dofs[plan.dof_off[t]:plan.dof_off[t + 1]]
        npv = int(plan.npiv[t]); nb = dofs.size - npv
        bnd = dofs[npv:]
        # the subtree's own matrix: the element matrices of its cells; constrained rows and columns emptied, a unit diagonal where a
        # constrained DOF is eliminated inside the subtree (on the boundary it gets its one when an ancestor eliminates it)
        rows = np.repeat(cd[cells], cd.shape[1], axis=1).ravel(); cols = np.tile(cd[cells], (1, cd.shape[1])).ravel()
        Ks = sp.coo_matrix((Ke[cells].ravel(), (rows, cols)), shape=K.shape).tocsr()
        keep = sp.diags((~masked).astype(float))
        interior = np.setdiff1d(np.unique(cd[cells]), bnd)
        unit = np.zeros(m.ndof); unit[interior[masked[interior]]] = 1.0
        Ks = (keep @ Ks @ keep + sp.diags(unit)).tocsr()
        A = Ks[interior][:, interior].toarray(); B = Ks[bnd][:, interior].toarray(); D = Ks[bnd][:, bnd].toarray()
        S_ref = D - B @ np.linalg.solve(A, B.T)
        S = torch.empty(nb * nb, dtype=torch.float64, device="cuda")
        c.front_schur_get(t, S)
        S = S.cpu().numpy().reshape(nb, nb).T             # column-major on the device; the lower triangle is what is maintained
        lo = np.tril_indices(nb)
        assert np.abs(S[lo] - S_ref[lo]).max() < 1e-9 * np.abs(S_ref).max(), t
        with pytest.raises(FemoHipError):
            c.front_block_set(t, torch.zeros(1, dtype=torch.float64, device="cuda"))      # a front with pivots is never overwritten
    c.factorize_range(L - 1, L, False)                   # the root: its block is gathered from the two children
    # forward sweep over all levels, then backward: M^-1 v; K M^-1 v = v up to the rounding of the factor
    v = rng.uniform(-1, 1, m.ndof)
    _put(c, "z", v)
    c.frontal_sweep("z", 0, L, False)
    c.frontal_sweep("z", 0, L, True)
    x = _get(c, "z")
    assert rel(K @ x, v) < 1e-7
    # ... and in two halves of the tree, as the multi-GPU driver sweeps (local levels, then the replicated top)
    _put(c, "z", v)
    c.frontal_sweep("z", 0, L // 2, False); c.frontal_sweep("z", L // 2, L, False)
    c.frontal_sweep("z", L // 2, L, True); c.frontal_sweep("z", 0, L // 2, True)
    assert rel(_get(c, "z"), x) < 1e-13
    prof = c.factorize_profile()
    assert prof["trailing"]["launches"] > 0 and prof["trailing_flops"] > 0
    again = c.factorize_profile(run=False)
    assert again["trailing"]["launches"] >= prof["trailing"]["launches"]
    sw = c.sweep_profile()
    assert sw.shape == (L, 2) and np.all(sw >= 0)
    c.close()
