"""GPU parity: every libfemo_hip entry point against the CPU oracle on the same seeded inputs.

Tolerances (float64 everywhere): operator-level quantities (K x, residual, load, diagonal,
element matrices, functionals, partial gradients) agree to 1e-11 relative -- same formulas,
different summation order; solved quantities (displacement, compliance, total gradient) to 1e-7
relative with the PCG tolerance set to 1e-12, the north-star bar being 1e-8 on d compliance /
d thickness (BASELINE.json)."""
import numpy as np
import pytest

from femo_alpha_amd.mesh import ShellMesh, plate_mesh, quads_to_triangles, wing_skin_mesh

pytestmark = pytest.mark.gpu

CLAMP = lambda x: np.less(x[0], 3e-16)


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-300)


def _mesh(kind):
    if kind == "plate":
        return plate_mesh(2.0, 10.0, 4, 12)
    if kind == "plate24":            # wide enough that the top tree levels take the panel-parallel solve kernels
        return plate_mesh(2.0, 5.0, 24, 60)
    if kind == "warped":
        return wing_skin_mesh(6, 14, shuffle=True)
    if kind == "tri":
        return quads_to_triangles(wing_skin_mesh(5, 9, shuffle=True))
    if kind == "delaunay":           # unstructured triangulation (valences 3..9, no mesh lines)
        from femo_alpha_amd.mesh import unstructured_skin_mesh
        return unstructured_skin_mesh(7, 15)
    if kind == "tee":                # branching surface: flange + web, edges shared by three cells
        from femo_alpha_amd.mesh import tee_beam_mesh
        return tee_beam_mesh(1.0, 0.5, 5.0, 4, 2, 10)
    raise ValueError(kind)


def _pair(kind, ewm=False, ewp=False, uhat=False, bc="penalty", beta=1e15, seed=0):
    from femo_alpha_amd.backend import ShellContext
    from oracle.rm_shell_oracle import ShellOracle, degree4_rule
    m = _mesh(kind)
    rng = np.random.default_rng(seed)
    nT = m.nel if ewm else m.nn
    nF = m.nel if ewp else m.nn
    fields = dict(thickness=0.05 * (1 + 0.3 * rng.uniform(-1, 1, nT)), E=3e7 * (1 + 0.2 * rng.uniform(-1, 1, nT)),
                  nu=0.3 + 0.05 * rng.uniform(-1, 1, nT), density=10 * (1 + 0.1 * rng.uniform(-1, 1, nT)),
                  F_solid=rng.uniform(-1, 1, (nF, 3)))
    if uhat:
        fields["uhat"] = 0.02 * rng.uniform(-1, 1, (m.nn, 3))
    marker = CLAMP if kind.startswith("plate") or kind == "tee" else (lambda x: np.less(x[1], 1e-12))
    pf = m.penalty_facets(marker) if bc == "penalty" else None
    sd = m.locate_dofs_geometrical(marker) if bc == "strong" else None
    o = ShellOracle(m, element_wise_material=ewm, elementwise_pressure=ewp, penalty_facets=pf, strong_dofs=sd, beta=beta)
    o.set_fields(h=fields["thickness"], E=fields["E"], nu=fields["nu"], rho=fields["density"], f=fields["F_solid"],
                 uhat=fields.get("uhat"))
    c = ShellContext(m, element_wise_material=ewm, elementwise_pressure=ewp)
    for k, v in fields.items():
        c.set_field(k, v)
    if pf is not None:
        c.set_penalty_facets(pf, beta)
    if sd is not None:
        c.set_strong_dofs(sd)
    return m, o, c, rng


CASES = [("plate", False, False, False, "penalty"), ("warped", False, False, False, "penalty"),
         ("warped", True, True, False, "strong"), ("warped", False, False, True, "penalty"),
         ("tri", False, False, False, "penalty"), ("tri", True, False, True, "strong"),
         ("tee", False, False, True, "penalty"), ("delaunay", False, False, True, "penalty"), ("delaunay", True, True, False, "strong")]


@pytest.mark.parametrize("kind,ewm,ewp,uhat,bc", CASES)
def test_operator_level_parity(kind, ewm, ewp, uhat, bc):
    m, o, c, rng = _pair(kind, ewm, ewp, uhat, bc)
    tol = 1e-11
    # element matrices
    Ke = c.element_matrices()
    Ko = o.element_matrices()
    assert rel(Ke, Ko) < tol
    # K x, with the Dirichlet treatment
    x = rng.uniform(-1, 1, m.ndof)
    K = o.assemble_K()
    assert rel(c.apply_K(x), K @ x) < tol
    assert rel(c.diagonal(), K.diagonal()) < tol
    assert rel(c.load_vector(), o.load_vector()) < tol
    w = rng.uniform(-1, 1, m.ndof) * 1e-3
    if bc == "strong":
        w[o.strong_dofs] = 0.0
    assert rel(c.residual(w), K @ w - o.load_vector()) < tol
    # functionals and their partials at a given state
    c.set_state(w)
    assert abs(c.functional("compliance") - o.compliance(w)) < tol * abs(o.compliance(w))
    assert abs(c.functional("mass") - o.mass()) < tol * abs(o.mass())
    assert abs(c.functional("elastic_energy") - o.elastic_energy(w)) < tol * abs(o.elastic_energy(w))
    assert rel(c.dfunctional("compliance", "disp_solid"), o.dcompliance_du(w)) < tol
    assert rel(c.dfunctional("compliance", "thickness"), o.dcompliance_dh(w)) < tol
    assert rel(c.dfunctional("mass", "thickness"), o.dmass_dh()) < tol
    assert rel(c.dfunctional("elastic_energy", "disp_solid"), o.apply_K(w, with_penalty=False)) < tol
    assert rel(c.dfunctional("elastic_energy", "thickness"), 0.5 * o.dRdfield_T("h", w, w)) < tol
    assert np.all(c.dfunctional("compliance", "F_solid") == 0.0)
    lam = rng.uniform(-1, 1, m.ndof)
    for arg, name in (("thickness", "h"), ("E", "E"), ("nu", "nu")):
        assert rel(c.dRdarg_T(arg, lam), o.dRdfield_T(name, w, lam)) < tol
    assert rel(c.dRdarg_T("F_solid", lam), o.dRdf_T(lam)) < tol


@pytest.mark.parametrize("kind,ewm,bc", [("plate", False, "penalty"), ("plate", True, "strong"), ("warped", False, "strong")])
def test_forward_adjoint_parity(kind, ewm, bc):
    """The parity triple of BASELINE.json: displacement, compliance, d compliance / d thickness."""
    m, o, c, rng = _pair(kind, ewm=ewm, bc=bc, beta=1e15)
    w_ref, J_ref, dJ_ref = o.forward_adjoint()
    c.set_solver(rtol=1e-12, maxit=400000, check_every=100)
    it, rr = c.solve_state(zero_guess=True)
    assert rr <= 1e-12
    w = c.get_state()
    assert rel(w, w_ref) < 1e-7
    assert abs(c.functional("compliance") - J_ref) < 1e-8 * abs(J_ref)
    dJ, it2, rr2 = c.total_gradient("compliance", "thickness")
    assert rel(dJ, dJ_ref) < 1e-7
    # operator protocol pieces compose to the same thing
    lam, _, _ = c.solve_linear(c.dfunctional("compliance", "disp_solid"))
    dJ2 = c.dfunctional("compliance", "thickness") - c.dRdarg_T("thickness", lam)
    assert rel(dJ2, dJ) < 1e-9


def test_errors_are_loud():
    from femo_alpha_amd._lib import FemoHipError
    from femo_alpha_amd.backend import ShellContext
    m = plate_mesh(2.0, 10.0, 2, 4)
    c = ShellContext(m)
    with pytest.raises(FemoHipError):
        c.set_field("no_such_field", np.ones(3))
    with pytest.raises(FemoHipError):
        c.set_field("thickness", np.ones(m.nn + 1))
    with pytest.raises(FemoHipError):
        c.functional("pnorm_of_nothing")
    c.set_field("thickness", np.array([0.1]))               # broadcast of a length-1 array
    assert np.all(c.get_field("thickness") == 0.1)


@pytest.mark.parametrize("kind,ewm,bc,uhat,wide_cnt", [("plate", False, "penalty", False, None), ("warped", True, "strong", False, 0),
                                                       ("warped", False, "penalty", True, None), ("tri", False, "penalty", False, 0),
                                                       ("plate24", False, "penalty", False, 0), ("plate", False, "strong", False, 0),
                                                       ("tee", False, "penalty", False, None), ("tee", True, "strong", True, 0),
                                                       ("plate24", False, "strong", False, None), ("delaunay", False, "penalty", True, None),
                                                       ("delaunay", True, "strong", False, 0)])
def test_multifrontal_preconditioner(kind, ewm, bc, uhat, wide_cnt):
    """PCG preconditioned by the multifrontal Cholesky factorisation: a handful of iterations and
    the same parity triple as the reference's direct (MUMPS LU) solve.  Levels with few fronts take the wide
    (many workgroups per front) solve kernels by default -- on these small meshes that is every level;
    wide_cnt = 0 sends them through the one-workgroup-per-front kernels instead."""
    m, o, c, rng = _pair(kind, ewm=ewm, bc=bc, uhat=uhat)
    opts = {}
    if kind == "plate24":
        opts["wide_np"] = 96                         # wide kernels by pivot-block size: a mix of both paths in one tree
    if wide_cnt is not None:
        opts["wide_cnt"] = wide_cnt
    if kind in ("tee", "tri") and wide_cnt == 0:
        opts["swork_slots"] = 2                      # levels with more fronts than scratch slots are factorised in chunks
    if kind in ("plate24", "tri"):
        c.set_option("trailing", 1)                  # left-looking rank-k updates (the default on small meshes is right-looking)
    if kind in ("warped", "plate24"):
        c.set_option("grid_chunk", 3)                # levels launched three fronts at a time (grid y/z extent limit)
    plan = c.enable_frontal(leaf_size=8, **opts)
    c.set_solver(preconditioner=2, rtol=1e-12, maxit=50, check_every=1)
    info = c.factorize()
    assert plan.ntree > 1
    if kind == "plate24":
        assert plan.npiv.max() > 192
    assert info["pivots_repaired"] == 0
    w_ref, J_ref, dJ_ref = o.forward_adjoint()
    it, rr = c.solve_state(zero_guess=True)
    assert it <= 6 and rr <= 1e-12
    assert rel(c.get_state(), w_ref) < 1e-8
    # 1e-8: the oracle's own LU solve of the 1e15-penalised system is only good to ~1e-9
    assert abs(c.functional("compliance") - J_ref) < 1e-8 * abs(J_ref)
    dJ, it2, rr2 = c.total_gradient("compliance", "thickness")
    assert it2 <= 6
    assert rel(dJ, dJ_ref) < 1e-8
    # the factorisation is refreshed when a field changes
    c.set_field("thickness", 1.1 * c.get_field("thickness"))
    it3, rr3 = c.solve_state(zero_guess=True)
    assert it3 <= 6 and rr3 <= 1e-12
    o.set_fields(h=1.1 * o.h)
    assert rel(c.get_state(), o.solve()) < 1e-8


@pytest.mark.parametrize("sp,force_right,ahead", [(256, False, 1), (256, True, 1), (256, True, 0), (512, True, 1)])
def test_super_panel_schedule_gives_the_same_factor(sp, force_right, ahead):
    """Right-looking levels may update the trailing matrix once per super-panel of 256 / 512 factor columns instead of
    once per 128 (option "super_panel"), with the bulk update on a second stream beside the next super-panel's panels
    (option "super_panel_ahead"): the factor is the same up to rounding, whatever the schedule -- same iteration count, same
    solution as the panel-by-panel schedule, same parity with the oracle.  The root front has 582 pivots: three
    super-panels of 256, so the second-stream dependencies (two bulk updates in flight) are exercised."""
    from femo_alpha_amd.backend import ShellContext
    from oracle.rm_shell_oracle import ShellOracle, degree4_rule
    m = plate_mesh(2.0, 5.0, 64, 64)
    rng = np.random.default_rng(5)
    fields = dict(thickness=0.05 * (1 + 0.3 * rng.uniform(-1, 1, m.nn)), E=np.full(m.nn, 3e7), nu=np.full(m.nn, 0.3),
                  density=np.full(m.nn, 10.0), F_solid=rng.uniform(-1, 1, (m.nn, 3)))
    pf = m.penalty_facets(CLAMP)
    sols = []
    for super_panel in (0, sp):
        c = ShellContext(m)
        for k, v in fields.items():
            c.set_field(k, v)
        c.set_penalty_facets(pf, 1e15)
        c.set_option("super_panel", super_panel)
        c.set_option("fused_schur", 1 if super_panel else 0)      # the reference run also fills the Schur columns in the extend-add
        c.set_option("super_panel_cnt", 64)
        c.set_option("super_panel_ahead", ahead)
        if force_right:
            c.set_option("trailing", 2)
        plan = c.enable_frontal(leaf_size=8)
        assert plan.npiv.max() > 2 * 256
        c.set_solver(preconditioner=2, rtol=1e-12, maxit=50, check_every=1)
        for _ in range(2):                              # twice: the streams must also be in order across factorisations
            info = c.factorize()
            assert info["pivots_repaired"] == 0
        it, rr = c.solve_state(zero_guess=True)
        assert it <= 6 and rr <= 1e-12
        sols.append((it, c.get_state()))
        c.close()
    assert sols[0][0] == sols[1][0]
    assert rel(sols[1][1], sols[0][1]) < 1e-9
    if sp == 256 and force_right and ahead:
        o = ShellOracle(m, penalty_facets=pf, beta=1e15)
        o.set_fields(h=fields["thickness"], E=fields["E"], nu=fields["nu"], rho=fields["density"], f=fields["F_solid"])
        assert rel(sols[1][1], o.solve()) < 1e-8


@pytest.mark.parametrize("shape,tri", [((1, 1), False), ((1, 2), False), ((1, 1), True), ((3, 3), False)])
def test_tiny_meshes_single_front(shape, tri):
    """One to nine cells: the elimination tree is a single front (no extend-add, no Schur complement); forward and
    adjoint solves against the oracle."""
    from femo_alpha_amd.backend import ShellContext
    from oracle.rm_shell_oracle import ShellOracle, degree4_rule
    m = plate_mesh(1.0, float(shape[1]), shape[0], shape[1])
    if tri:
        m = quads_to_triangles(m)
    f = np.tile([0.0, 0.0, 1.0], (m.nn, 1))
    c = ShellContext(m)
    for k, v in (("thickness", [0.1]), ("E", [1e6]), ("nu", [0.3]), ("density", [1.0]), ("F_solid", f)):
        c.set_field(k, v)
    pf = m.penalty_facets(CLAMP)
    c.set_penalty_facets(pf, 1e12)
    plan = c.enable_frontal()
    assert plan.ntree == 1
    c.set_solver(preconditioner=2, rtol=1e-12, maxit=20, check_every=1)
    it, rr = c.solve_state()
    o = ShellOracle(m, penalty_facets=pf, beta=1e12)
    o.set_fields(h=0.1, E=1e6, nu=0.3, rho=1.0, f=f)
    w, J, dJ = o.forward_adjoint()
    assert it <= 3 and rel(c.get_state(), w) < 1e-10
    g, it2, rr2 = c.total_gradient("compliance", "thickness")
    assert rel(g, dJ) < 1e-9


@pytest.mark.parametrize("kind,bc,precond", [("plate", "penalty", 2), ("warped", "strong", 2), ("thick", "strong", 0),
                                             ("thick", "penalty", 0)])
def test_bicgstab_matches_conjugate_gradients(kind, bc, precond):
    """The second Krylov method of the north star: right-preconditioned BiCGStab with the same preconditioners reaches
    the same state and the same adjoint gradient as CG.  With the Jacobi preconditioner it runs its full recurrence
    (tens of iterations) on a thick, well-conditioned plate -- on thin shells Jacobi-BiCGStab stagnates or breaks
    down, which is why the multifrontal preconditioner exists."""
    if kind == "thick":
        from femo_alpha_amd.backend import ShellContext
        from femo_alpha_amd.mesh import plate_mesh
        m = plate_mesh(2.0, 2.0, 4, 4)
        c = ShellContext(m)
        for name, v in (("thickness", [0.5]), ("E", [1e6]), ("nu", [0.3]), ("density", [1.0]),
                        ("F_solid", np.tile([0.0, 0.0, 1.0], (m.nn, 1)))):
            c.set_field(name, v)
        if bc == "penalty":
            c.set_penalty_facets(m.penalty_facets(CLAMP), 1e8)
        else:
            c.set_strong_dofs(m.locate_dofs_geometrical(CLAMP))
    else:
        m, o, c, rng = _pair(kind, bc=bc)
    if precond == 2:
        c.enable_frontal(leaf_size=8)
        c.set_solver(preconditioner=2, rtol=1e-12, maxit=50, check_every=1)
    else:
        c.set_solver(preconditioner=0, rtol=1e-11, maxit=200000, check_every=10)
    it_cg, rr_cg = c.solve_state(zero_guess=True)
    w_cg = c.get_state()
    g_cg, _, _ = c.total_gradient("compliance", "thickness")
    c.set_krylov("bicgstab")
    it_b, rr_b = c.solve_state(zero_guess=True)
    w_b = c.get_state()
    g_b, it_g, _ = c.total_gradient("compliance", "thickness")
    assert rr_b <= (1e-12 if precond == 2 else 1e-11)
    if precond == 2:
        assert it_b <= 3 and it_g <= 3
    else:
        assert it_b > 5                       # the recurrence really ran
    assert rel(w_b, w_cg) < (1e-9 if precond == 2 else 1e-6)
    assert rel(g_b, g_cg) < (1e-8 if precond == 2 else 1e-5)
    c.set_krylov("cg")
    with pytest.raises(KeyError):
        c.set_krylov("gmres")


@pytest.mark.parametrize("kind,bc", [("warped", "penalty"), ("tri", "penalty"), ("plate", "strong")])
def test_shape_sensitivities_vs_oracle_finite_differences(kind, bc):
    """d/d uhat of the outputs and (dR/d uhat)^T lambda: dual-number kernels against central finite
    differences of the CPU oracle (which evaluates F = I + grad(uhat), gradx and J in the primal)."""
    m, o, c, rng = _pair(kind, uhat=True, bc=bc, beta=1e6)
    w = rng.uniform(-1, 1, m.ndof) * 1e-3
    lam = rng.uniform(-1, 1, m.ndof)
    if bc == "strong":
        w[o.strong_dofs] = 0.0
        lam[o.strong_dofs] = 0.0
    c.set_state(w)
    u0 = o.uhat.copy()
    g_c = c.dfunctional("compliance", "uhat").reshape(-1, 3)
    g_m = c.dfunctional("mass", "uhat").reshape(-1, 3)
    g_e = c.dfunctional("elastic_energy", "uhat").reshape(-1, 3)
    g_r = c.dRdarg_T("uhat", lam).reshape(-1, 3)
    K = lambda: o.assemble_K(with_strong=False)

    def phis():
        return (o.compliance(w), o.mass(), o.elastic_energy(w), lam @ (K() @ w - o.load_vector()))

    step = 1e-6
    for v in rng.choice(m.nn, 3, replace=False):
        for comp in range(3):
            up = u0.copy(); up[v, comp] += step
            um = u0.copy(); um[v, comp] -= step
            o.set_fields(uhat=up); fp = phis()
            o.set_fields(uhat=um); fm = phis()
            o.set_fields(uhat=u0)
            fd = [(a - b) / (2 * step) for a, b in zip(fp, fm)]
            for name, g, d in (("compliance", g_c, fd[0]), ("mass", g_m, fd[1]), ("energy", g_e, fd[2]), ("residual", g_r, fd[3])):
                assert abs(g[v, comp] - d) <= 2e-6 * np.abs(g).max() + 1e-9 * abs(d), (name, v, comp, g[v, comp], d)


@pytest.mark.parametrize("kind,ewm,uhat", [("warped", False, False), ("warped", True, True), ("plate", False, False)])
def test_stress_outputs(kind, ewm, uhat):
    """p-norm aggregate and DG1 field of the top-surface von Mises stress, and the partial gradients of the
    aggregate, against the oracle (value) and finite differences of the oracle (gradients)."""
    from oracle.rm_shell_oracle import ShellOracle, degree4_rule
    m, o, c, rng = _pair(kind, ewm=ewm, uhat=uhat, beta=1e6)
    w = rng.uniform(-1, 1, m.ndof) * 1e-4
    c.set_state(w)
    o3 = ShellOracle(m, element_wise_material=ewm, nquad=degree4_rule(m))            # the degree-4 measure
    o3.set_fields(h=o.h, E=o.E, nu=o.nu, rho=o.rho, f=o.f, uhat=o.uhat)
    mval, rho = 1e-6, 6.0
    c.set_stress_params(mval, rho)
    assert abs(c.functional("pnorm_stress") - o3.pnorm_stress(w, mval, rho)) < 1e-10 * o3.pnorm_stress(w, mval, rho)
    if kind != "tri":
        assert rel(c.field_output("stress").reshape(m.nel, -1), o.stress_dg1(w)) < 1e-10
    # reference defaults (m = 1e-6, rho = 100) on the value
    c.set_stress_params(1e-6, 100.0)
    v100 = o3.pnorm_stress(w, 1e-6, 100)
    assert abs(c.functional("pnorm_stress") - v100) < 1e-9 * v100
    c.set_stress_params(mval, rho)
    area = o3.pnorm_stress(w * 0, mval, 0.0, alpha=1.0)                # int J dx with rho = 0 -> area (uhat included)
    o0 = ShellOracle(m, element_wise_material=ewm, nquad=degree4_rule(m)); alpha = o0.pnorm_stress(w * 0, 1.0, 0.0, alpha=1.0)
    P = lambda ww=w: o3.pnorm_stress(ww, mval, rho, alpha=alpha)
    g_w = c.dfunctional("pnorm_stress", "disp_solid")
    for i in rng.choice(m.ndof, 6, replace=False):
        st = 1e-6 * max(abs(w[i]), 1e-4)
        wp = w.copy(); wp[i] += st; wm = w.copy(); wm[i] -= st
        fd = (P(wp) - P(wm)) / (2 * st)
        assert abs(g_w[i] - fd) <= 2e-6 * np.abs(g_w).max() + 1e-7 * abs(fd), (i, g_w[i], fd)
    base = dict(h=o3.h.copy(), E=o3.E.copy(), nu=o3.nu.copy())
    for arg, key in (("thickness", "h"), ("E", "E"), ("nu", "nu")):
        g = c.dfunctional("pnorm_stress", arg)
        for i in rng.choice(g.size, 3, replace=False):
            v = base[key].copy(); st = 1e-6 * v[i]
            v[i] += st; o3.set_fields(**{key: v}); fp = P()
            v[i] -= 2 * st; o3.set_fields(**{key: v}); fm = P()
            o3.set_fields(**{key: base[key]})
            fd = (fp - fm) / (2 * st)
            assert abs(g[i] - fd) <= 5e-6 * np.abs(g).max() + 1e-7 * abs(fd), (arg, i, g[i], fd)
    g_u = c.dfunctional("pnorm_stress", "uhat").reshape(-1, 3)
    u0 = o3.uhat.copy()
    for v_ in rng.choice(m.nn, 2, replace=False):
        for comp in range(3):
            up = u0.copy(); up[v_, comp] += 1e-6; o3.set_fields(uhat=up); fp = P()
            up[v_, comp] -= 2e-6; o3.set_fields(uhat=up); fm = P()
            o3.set_fields(uhat=u0)
            fd = (fp - fm) / 2e-6
            assert abs(g_u[v_, comp] - fd) <= 5e-6 * np.abs(g_u).max() + 1e-7 * abs(fd), (v_, comp, g_u[v_, comp], fd)


@pytest.mark.parametrize("kind,ewm", [("warped", False), ("warped", True)])
def test_regularization_and_volume_outputs(kind, ewm):
    """The thickness regularisation the reference adds to the compliance ('H1' nodal, 'L2' element-wise,
    rm_shell_pde.py:64-89) and the volume int h J dx (rm_shell_pde.py:98-99) as outputs of their own."""
    m, o, c, rng = _pair(kind, ewm=ewm, uhat=True)
    w = rng.uniform(-1, 1, m.ndof) * 1e-3
    c.set_state(w)
    reg = o.regularization()
    assert abs(c.functional("regularization") - reg) <= 1e-12 * abs(reg)
    assert abs(c.functional("compliance") - o.compliance(w)) <= 1e-12 * abs(o.compliance(w))
    assert rel(c.dfunctional("regularization", "thickness"), o.dcompliance_dh(w)) < 1e-12
    assert np.all(c.dfunctional("regularization", "uhat") == 0.0)
    assert np.all(c.dfunctional("regularization", "disp_solid") == 0.0)
    vol = c.functional("volume")
    mass_unit = c.functional("mass")
    c.set_field("density", np.ones(c.field_size("density")))
    assert abs(c.functional("mass") - vol) <= 1e-13 * vol          # rho = 1: mass == volume
    assert mass_unit > 0


@pytest.mark.parametrize("kind,uhat", [("warped", True), ("tri", False)])
def test_stress_aggregate_on_subdomains(kind, uhat):
    """Per-tag stress aggregates (the reference's dxx(i) measure, rm_shell_model.py:242-253): value and partial
    gradients restricted to a sub-domain, each normalised by its own reference area; selecting -1 restores the mesh."""
    from oracle.rm_shell_oracle import ShellOracle, degree4_rule
    m, o, c, rng = _pair(kind, uhat=uhat, beta=1e6)
    w = rng.uniform(-1, 1, m.ndof) * 1e-4
    c.set_state(w)
    o3 = ShellOracle(m, nquad=degree4_rule(m))
    o3.set_fields(h=o.h, E=o.E, nu=o.nu, rho=o.rho, f=o.f, uhat=o.uhat)
    o0 = ShellOracle(m, nquad=degree4_rule(m))                                        # reference configuration: the frozen areas
    mval, rho = 1e-6, 6.0
    c.set_stress_params(mval, rho)
    perm = rng.permutation(m.nel)
    groups = [perm[: m.nel // 3], perm[m.nel // 3: m.nel // 2]]          # the rest stays untagged
    tags = -np.ones(m.nel, dtype=np.int32)
    for i, g in enumerate(groups):
        tags[g] = i
    c.set_cell_tags(tags, len(groups))
    whole = c.functional("pnorm_stress")
    assert abs(whole - o3.pnorm_stress(w, mval, rho)) < 1e-10 * whole
    for i, g in enumerate(groups):
        alpha = o0.pnorm_stress(w * 0, 1.0, 0.0, alpha=1.0, cells=g)
        P = lambda ww=w: o3.pnorm_stress(ww, mval, rho, alpha=alpha, cells=g)
        c.select_subdomain(i)
        assert abs(c.functional("pnorm_stress") - P()) < 1e-10 * P()
        g_w = c.dfunctional("pnorm_stress", "disp_solid")
        touched = np.unique(m.cell_dofs()[g])
        assert np.all(g_w[np.setdiff1d(np.arange(m.ndof), touched)] == 0.0)
        for k in rng.choice(touched, 4, replace=False):
            st = 1e-6 * max(abs(w[k]), 1e-4)
            wp = w.copy(); wp[k] += st; wm = w.copy(); wm[k] -= st
            fd = (P(wp) - P(wm)) / (2 * st)
            assert abs(g_w[k] - fd) <= 2e-6 * np.abs(g_w).max() + 1e-7 * abs(fd), (k, g_w[k], fd)
        g_h = c.dfunctional("pnorm_stress", "thickness")
        h0 = o3.h.copy()
        k = int(np.argmax(np.abs(g_h)))
        hp = h0.copy(); hp[k] *= 1 + 1e-6; o3.set_fields(h=hp); fp = P()
        hp[k] = h0[k] * (1 - 1e-6); o3.set_fields(h=hp); fm = P()
        o3.set_fields(h=h0)
        fd = (fp - fm) / (2e-6 * h0[k])
        assert abs(g_h[k] - fd) <= 5e-6 * abs(fd), (g_h[k], fd)
        g_u = c.dfunctional("pnorm_stress", "uhat").reshape(-1, 3)
        assert np.all(g_u[np.setdiff1d(np.arange(m.nn), np.unique(m.cells[g]))] == 0.0)
    # tip_disp = 0.5 int u.u J and area = int J over a sub-domain (rm_shell_pde.py:95-105): the groups partition the
    # tagged cells, so the pieces add up to the whole-mesh values
    c.select_subdomain(-1)
    comp_uu = c.functional("compliance") - c.functional("regularization")
    tag_rest = tags.copy(); tag_rest[tags < 0] = len(groups)
    c.set_cell_tags(tag_rest, len(groups) + 1)
    parts, areas = [], []
    for i in range(len(groups) + 1):
        c.select_subdomain(i)
        parts.append(c.functional("tip_disp")); areas.append(c.functional("area"))
        g_t = c.dfunctional("tip_disp", "disp_solid")
        cells_i = np.nonzero(tag_rest == i)[0]
        touched = np.unique(m.cell_dofs()[cells_i])
        assert np.all(g_t[np.setdiff1d(np.arange(m.ndof), touched)] == 0.0)
    c.select_subdomain(-1)
    assert abs(2.0 * sum(parts) - comp_uu) <= 1e-12 * comp_uu
    assert abs(sum(areas) - c.functional("area")) <= 1e-13 * sum(areas)
    g_all = c.dfunctional("compliance", "disp_solid")
    c.select_subdomain(0)
    # the compliance ignores the selection (atomic adds: the summation order may differ in the last bit)
    assert np.abs(c.dfunctional("compliance", "disp_solid") - g_all).max() <= 1e-14 * np.abs(g_all).max()
    c.set_cell_tags(tags, len(groups))
    c.select_subdomain(-1)
    assert c.functional("pnorm_stress") == whole
    with pytest.raises(Exception, match="unknown sub-domain"):
        c.select_subdomain(5)


@pytest.mark.parametrize("kind", ["plate", "warped", "tri"])
def test_csr_assembly(kind):
    """Wave-segmented scatter-add of the element matrices into CSR against the oracle's scipy assembly."""
    m, o, c, rng = _pair(kind)
    info = c.enable_csr()
    K = c.assemble_csr()
    Kref = o.assemble_K(with_penalty=False, with_strong=False)
    assert info["nnz"] == Kref.nnz
    assert abs(K - Kref).max() < 1e-11 * abs(Kref).max()
    K2 = c.assemble_csr()                                    # repeatable
    assert abs(K2 - K).max() <= 1e-13 * abs(Kref).max()
    # the device-built map against the numpy one: same pattern, same values
    rowptr, colidx = info["rowptr"].copy(), info["colidx"].copy()
    host = c.enable_csr(host_map=True)
    assert np.array_equal(rowptr, host["rowptr"]) and np.array_equal(colidx, host["colidx"])
    assert abs(c.assemble_csr() - K).max() <= 1e-13 * abs(Kref).max()


@pytest.mark.parametrize("case", ["quad CG2CG1", "quad CG1CG1", "tri CG2CG1", "tri CG2CR1", "tri CG1CG1"])
def test_element_matrices_against_the_symbolic_derivation(case):
    """The HIP element matrices DIRECTLY against the independent symbolic derivations (tests/golden/make_sympy_golden.py,
    make_sympy_golden_tri.py) -- not through the oracle: the CSR matrix of a one-cell mesh is the element matrix.  Affine cells,
    nodal thickness: the default rules integrate exactly."""
    import os
    from femo_alpha_amd.backend import ShellContext
    gd = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    shape, element = case.split()
    if shape == "quad" and element == "CG2CG1":
        g = np.load(os.path.join(gd, "sympy_element.npz"))
        X, h, E, nu, ref = g["A_X"], g["A_h"], g["A_E"][:1], g["A_nu"][:1], g["A_Ke"]
    else:
        g = np.load(os.path.join(gd, "sympy_triangle.npz"))
        if shape == "quad":
            X, h, E, nu, ref = g["Q_X"], g["Q_h"], g["Q_E"], g["Q_nu"], g["Q_Ke_cg1cg1"]
        else:
            X, h, E, nu, ref = g["T_X"], g["T_h"], g["T_E"], g["T_nu"], g["T_Ke_" + element.lower()]
    m = ShellMesh(X, np.arange(X.shape[0])[None, :], element)
    c = ShellContext(m)
    c.set_field("thickness", h); c.set_field("E", E); c.set_field("nu", nu); c.set_field("density", [1.0])
    c.enable_csr()
    K = c.assemble_csr().toarray()
    d = m.cell_dofs()[0]
    Ke = K[np.ix_(d, d)]
    assert Ke.shape == ref.shape and np.abs(Ke - ref).max() < 1e-12 * np.abs(ref).max()
    c.close()


def test_warped_element_with_mesh_motion_against_the_symbolic_derivation():
    """... and the general case: a warped, non-planar quadrilateral with uhat != 0 and nodal h / E / nu.  The golden integrates the
    symbolic point values with the 5 x 5 Gauss rule (make_sympy_golden_tri.py, case W); the HIP kernels with that rule reproduce the
    element matrix (CSR of the one-cell mesh) and the load vector -- frame, differentiated normal, gradx = grad F^-1, J(uhat) and the
    interpolated material without the oracle in between."""
    import os
    from femo_alpha_amd.backend import ShellContext
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sympy_triangle.npz"))
    m = ShellMesh(g["W_X"], np.array([[0, 1, 2, 3]]))
    c = ShellContext(m, nquad=int(g["W_n"][0]))
    c.set_field("thickness", g["W_h"]); c.set_field("E", g["W_E"]); c.set_field("nu", g["W_nu"]); c.set_field("density", [1.0])
    c.set_field("uhat", g["W_uhat"]); c.set_field("F_solid", g["W_f"])
    c.enable_csr()
    d = m.cell_dofs()[0]
    Ke = c.assemble_csr().toarray()[np.ix_(d, d)]
    assert np.abs(Ke - g["W_Ke"]).max() < 1e-11 * np.abs(g["W_Ke"]).max()
    Fe = c.load_vector()[d[:27]]
    assert np.abs(Fe - g["W_Fe"]).max() < 1e-12 * np.abs(g["W_Fe"]).max()
    # the functionals of a given state, integrated with the same rule from the symbolic densities
    w = np.zeros(m.ndof); w[d] = np.concatenate([g["W_U"].ravel(), g["W_TH"].ravel()])
    c.set_field("density", g["W_rho"])
    c.set_state(w)
    assert abs(c.functional("compliance") - g["W_compliance"][0]) < 1e-11 * g["W_compliance"][0]
    assert abs(c.functional("mass") - g["W_mass"][0]) < 1e-12 * g["W_mass"][0]
    e_ref = 0.5 * w[d] @ g["W_Ke"] @ w[d]
    assert abs(c.functional("elastic_energy") - e_ref) < 1e-11 * abs(e_ref)
    # the two halves of the north star's adjoint gradient on that state, for a given multiplier: (dR/dthickness)^T lam and
    # d compliance / d thickness (state_operation.py:174-184, output_operation.py:58-69)
    lam = np.zeros(m.ndof); lam[d] = np.concatenate([g["W_LU"].ravel(), g["W_LT"].ravel()])
    gR = c.dRdarg_T("thickness", lam)
    assert np.abs(gR - g["W_dRdh_T_lam"]).max() < 1e-11 * np.abs(g["W_dRdh_T_lam"]).max()
    gJ = c.dfunctional("compliance", "thickness")
    assert np.abs(gJ - g["W_dcompliance_dh"]).max() < 1e-11 * np.abs(g["W_dcompliance_dh"]).max()
    # ... and the other inputs' (rm_shell_model.py:216-232): E, nu, F_solid
    for name, key in (("E", "W_dRdE_T_lam"), ("nu", "W_dRdnu_T_lam")):
        assert np.abs(c.dRdarg_T(name, lam) - g[key]).max() < 1e-11 * np.abs(g[key]).max(), name
    gF = c.dRdarg_T("F_solid", lam).reshape(-1, 3)
    assert np.abs(gF - g["W_dRdf_T_lam"]).max() < 1e-12 * np.abs(g["W_dRdf_T_lam"]).max()
    # ... and the shape sensitivity (dR/duhat)^T lam through F(uhat), J(uhat), gradx (kinematics.py:12-44): the analytic HIP kernels
    # against three directional derivatives taken from the symbolic point values in 50-digit arithmetic (case W2)
    gU = c.dRdarg_T("uhat", lam).reshape(-1, 3)
    for D, ref in zip(g["W2_D"], g["W2_val"]):
        assert abs(np.sum(D * gU) - ref) < 1e-10 * abs(ref)
    # the inertia operator rho h (u.v + h_K^2 theta.eta) J dx (linear_shell_model.py:335-348): femo_op_apply_vec2 with aK = 0, aM = 1
    import torch
    Me = np.zeros((m.ndof, m.ndof))
    for j in range(m.ndof):
        c.vec_tensor("p").copy_(torch.from_numpy(np.eye(m.ndof)[j])); c.sync()
        c.op_apply_vec2("p", "Ap", 0.0, 1.0, with_penalty=False); c.sync()
        Me[:, j] = c.vec_tensor("Ap").cpu().numpy()
    Me = Me[np.ix_(d, d)]
    assert np.abs(Me - g["W_Me"]).max() < 1e-12 * np.abs(g["W_Me"]).max()
    c.close()


def test_stress_aggregate_against_the_symbolic_derivation():
    """1 / alpha int (m vm)^rho J dx of the top surface (rm_shell_pde.py:112-128, ShellStressRM with the thickness a field) on the warped
    quadrilateral with uhat != 0, nodal h / E / nu and a given state: the HIP aggregate against the symbolic von Mises stress integrated
    with the 3 x 3 rule of the degree-4 measure (make_sympy_golden_tri.py, case S; m = 2, rho = 4, alpha = 1)."""
    import os
    from femo_alpha_amd.backend import ShellContext
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sympy_triangle.npz"))
    m = ShellMesh(g["S_X"], np.array([[0, 1, 2, 3]]))
    c = ShellContext(m)
    c.set_field("thickness", g["S_h"]); c.set_field("E", g["S_E"]); c.set_field("nu", g["S_nu"]); c.set_field("density", [1.0])
    c.set_field("uhat", g["S_uhat"])
    d = m.cell_dofs()[0]
    w = np.zeros(m.ndof); w[d] = np.concatenate([g["S_U"].ravel(), g["S_TH"].ravel()])
    c.set_state(w)
    c.set_stress_params(m=2.0, rho=4.0)
    c.set_stress_alpha(1.0)
    assert abs(c.functional("pnorm_stress") - g["S_pnorm"][0]) < 1e-11 * g["S_pnorm"][0]
    c.close()
    # the six sums int sigma_ij J dx of the top-surface in-plane stress (sum_stress_subdomain, rm_shell_pde.py:130-150), 5 x 5 rule
    c = ShellContext(m, nquad=5)
    c.set_field("thickness", g["S_h"]); c.set_field("E", g["S_E"]); c.set_field("nu", g["S_nu"]); c.set_field("density", [1.0])
    c.set_field("uhat", g["S_uhat"])
    c.set_state(w)
    got = np.array([c.functional("sum_stress_" + k) for k in ("x", "y", "z", "xy", "xz", "yz")])
    assert np.abs(got - g["S_sum_stress"]).max() < 1e-11 * np.abs(g["S_sum_stress"]).max()
    c.close()


def test_triangle_rules_against_the_symbolic_derivation():
    """The triangle rules (round 6): a tilted triangle with uhat != 0 and nodal h / E / nu -- a nodal Poisson ratio makes the integrand
    rational, so the rule decides the number -- integrated from the symbolic point values with the symmetric rules of degree 6 / 9 / 12
    (make_sympy_golden_tri_rules.py).  The HIP kernels with the rule of that degree reproduce element matrix (CSR of the one-cell mesh),
    load vector, functionals and sensitivities; a context left to itself takes degree 9 (UFL's estimate for these forms) the moment such
    a field arrives; and the p-norm stress measure is the 6-point rule of the reference's quadrature_degree 4 (rm_shell_model.py:200-205)
    whatever the operator's rule -- with rho = 100 the 12-point rule of rounds 1-5 gives 447 times that number on this cell."""
    import os
    from femo_alpha_amd.backend import FemoHipError, ShellContext
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sympy_triangle_rules.npz"))
    m = ShellMesh(g["TR_X"], np.array([[0, 1, 2]]))
    d = m.cell_dofs()[0]
    w = np.zeros(m.ndof); w[d] = np.concatenate([g["TR_U"].ravel(), g["TR_TH"].ravel()])
    lam = np.zeros(m.ndof); lam[d] = np.concatenate([g["TR_LU"].ravel(), g["TR_LT"].ravel()])

    def fill(c):
        c.set_field("thickness", g["TR_h"]); c.set_field("E", g["TR_E"]); c.set_field("nu", g["TR_nu"]); c.set_field("density", g["TR_rho"])
        c.set_field("uhat", g["TR_uhat"]); c.set_field("F_solid", g["TR_f"])

    for deg in (6, 9, 12):
        c = ShellContext(m, nquad=deg)
        fill(c)
        assert c.nquad == deg and c.quadrature() == (deg, {6: 12, 9: 19, 12: 33}[deg])
        # the tables the kernels receive (femo_quadrature_tables, host only): weights of the rule sum to the area of the unit triangle
        import ctypes as C
        wts, npts = np.zeros(36), C.c_int32()
        assert c.lib.femo_quadrature_tables(3, deg, 0, 0, 0, C.byref(npts), wts.ctypes.data_as(C.POINTER(C.c_double)), *([None] * 7)) == 0
        assert npts.value == c.quadrature()[1] and abs(wts[:npts.value].sum() - 0.5) < 1e-16
        c.enable_csr()
        Ke = c.assemble_csr().toarray()[np.ix_(d, d)]
        ref = g[f"TR_Ke_d{deg}"]
        assert np.abs(Ke - ref).max() < 1e-11 * np.abs(ref).max(), deg
        Fe = c.load_vector()[d[:18]]
        assert np.abs(Fe - g[f"TR_Fe_d{deg}"]).max() < 1e-12 * np.abs(g[f"TR_Fe_d{deg}"]).max()
        c.set_state(w)
        assert abs(c.functional("compliance") - g[f"TR_compliance_d{deg}"][0]) < 1e-11 * g[f"TR_compliance_d{deg}"][0]
        assert abs(c.functional("mass") - g[f"TR_mass_d{deg}"][0]) < 1e-12 * g[f"TR_mass_d{deg}"][0]
        e_ref = 0.5 * w[d] @ ref @ w[d]
        assert abs(c.functional("elastic_energy") - e_ref) < 1e-11 * abs(e_ref)
        if deg == 9:
            for name, key in (("nu", "TR_dRdnu_d9"), ("thickness", "TR_dRdh_d9")):
                assert np.abs(c.dRdarg_T(name, lam) - g[key]).max() < 1e-11 * np.abs(g[key]).max(), name
        # the stress measure does not follow the operator's rule
        c.set_stress_params(m=2.0, rho=4.0); c.set_stress_alpha(1.0)
        assert abs(c.functional("pnorm_stress") - g["TR_pnorm4_d4"][0]) < 1e-11 * g["TR_pnorm4_d4"][0]
        c.set_stress_params(m=float(g["TR_m100"][0]), rho=100.0); c.set_stress_alpha(1.0)
        assert abs(c.functional("pnorm_stress") - g["TR_pnorm100_d4"][0]) < 1e-9 * g["TR_pnorm100_d4"][0]
        c.close()
    # left to itself: degree 6, and 9 from the moment the nodal Poisson ratio varies; back when it is uniform again; the rule can be
    # named after creation as well (femo_set_quadrature), and a degree without a rule is refused
    c = ShellContext(m)
    assert c.nquad == 6
    fill(c)
    assert c.nquad == 9 and c.quadrature() == (9, 19)
    c.enable_csr()
    Ke = c.assemble_csr().toarray()[np.ix_(d, d)]
    assert np.abs(Ke - g["TR_Ke_d9"]).max() < 1e-11 * np.abs(g["TR_Ke_d9"]).max()
    c.set_quadrature(12)
    Ke = c.assemble_csr().toarray()[np.ix_(d, d)]
    assert np.abs(Ke - g["TR_Ke_d12"]).max() < 1e-11 * np.abs(g["TR_Ke_d12"]).max()
    c.set_field("nu", [0.3])
    assert c.nquad == 6
    with pytest.raises(FemoHipError, match="degree"):
        c.set_quadrature(5)
    c.close()
    with pytest.raises(FemoHipError, match="degree"):
        ShellContext(m, nquad=5)


def test_penalty_term_of_cg2cr1_against_the_symbolic_facet_blocks():
    """CG2CR1: the penalty operator of the three facets of the affine triangle (3 x 3 rotation blocks: all three Crouzeix-Raviart functions
    have a trace on every facet) -- the HIP operator with the facets minus the one without, against the symbolic blocks (case PC)."""
    import os
    from femo_alpha_amd.backend import ShellContext
    from test_oracle import _penalty_reference_cr
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sympy_triangle.npz"))
    m = ShellMesh(g["PC_X"], np.array([[0, 1, 2]]), "CG2CR1")
    beta = 1e3
    d = m.cell_dofs()[0]

    def operator(with_facets):
        c = ShellContext(m)
        c.set_field("thickness", [0.05]); c.set_field("E", [2.0]); c.set_field("nu", [0.3]); c.set_field("density", [1.0])
        if with_facets:
            c.set_penalty_facets(np.array([[0, k] for k in range(3)]), beta=beta)
        K = np.stack([c.apply_K(np.eye(m.ndof)[j]) for j in range(m.ndof)], axis=1)
        c.close()
        return K[np.ix_(d, d)]
    P = operator(True) - operator(False)
    ref = _penalty_reference_cr(g, beta)
    assert np.abs(P - ref).max() < 1e-11 * np.abs(ref).max()


def test_penalty_term_against_the_symbolic_facet_blocks():
    """The penalty term of all four facets of the warped quadrilateral with uhat != 0 (linear_shell_model.py:323-333): the HIP operator with
    the facets minus the one without (unit vectors through femo_apply_K) against the symbolic blocks -- Nanson factor || J F^-T N ||,
    three-point facet rule, 1 / h_K (make_sympy_golden_tri.py, case P)."""
    import os
    from femo_alpha_amd.backend import ShellContext
    from test_oracle import _penalty_reference
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sympy_triangle.npz"))
    m = ShellMesh(g["P_X"], np.array([[0, 1, 2, 3]]))
    beta = 1e3
    d = m.cell_dofs()[0]

    def operator(with_facets):
        c = ShellContext(m)
        c.set_field("thickness", [0.05]); c.set_field("E", [2.0]); c.set_field("nu", [0.3]); c.set_field("density", [1.0])
        c.set_field("uhat", g["P_uhat"])
        if with_facets:
            c.set_penalty_facets(np.array([[0, k] for k in range(4)]), beta=beta)
        K = np.stack([c.apply_K(np.eye(m.ndof)[j]) for j in range(m.ndof)], axis=1)
        c.close()
        return K[np.ix_(d, d)]
    P = operator(True) - operator(False)
    ref = _penalty_reference(g, beta)
    assert np.abs(P - ref).max() < 1e-11 * np.abs(ref).max()


def test_non_convergence_and_indefinite_operators_raise():
    """The reference solves with a direct LU; here an iteration that stops short of rtol, or a Cholesky that meets a
    non-positive pivot, must not hand back numbers silently."""
    from femo_alpha_amd.backend import FemoConvergenceError, FemoNotPositiveDefiniteError
    m, o, c, rng = _pair("plate")
    c.set_solver(preconditioner=0, rtol=1e-12, maxit=5, check_every=5)       # starved Jacobi-PCG
    with pytest.raises(FemoConvergenceError):
        c.solve_state(zero_guess=True)
    c.set_option("strict", 0)                                                   # opting out: status 0, the caller judges relres
    it, rr = c.solve_state(zero_guess=True)
    assert it == 5 and rr > 1e-12
    c.set_option("strict", 1)
    c.use_direct_solver(leaf_size=8)
    it, rr = c.solve_state(zero_guess=True)
    assert it <= 6 and rr <= 1e-12
    good = c.get_field("thickness")
    c.set_field("thickness", -good)                  # negative thickness: membrane and shear blocks negative definite
    with pytest.raises(FemoNotPositiveDefiniteError):
        c.solve_state(zero_guess=True)
    with pytest.raises(FemoNotPositiveDefiniteError):
        c.factorize()
    c.set_option("allow_pivot_repair", 1)
    assert c.factorize()["pivots_repaired"] > 0
    c.set_option("allow_pivot_repair", 0)
    c.set_field("thickness", good)                   # and the context recovers
    it, rr = c.solve_state(zero_guess=True)
    assert it <= 6 and rel(c.get_state(), o.solve()) < 1e-8


def test_switching_preconditioners_after_a_field_change():
    """Two preconditioners, two staleness flags: a factorisation must not hide a stale Jacobi diagonal and vice versa;
    the density enters the operator only through the inertia term."""
    m, o, c, rng = _pair("plate")
    c.enable_frontal(leaf_size=8)
    c.set_solver(preconditioner=2, rtol=1e-12, maxit=50, check_every=1)
    c.solve_state(zero_guess=True)                                   # factor built, Jacobi diagonal never computed
    h2 = 1.3 * c.get_field("thickness")
    c.set_field("thickness", h2); o.set_fields(h=h2)
    w2 = o.solve()
    c.set_solver(preconditioner=0, rtol=1e-12, maxit=400000, check_every=100)
    it, rr = c.solve_state(zero_guess=True)                          # needs a fresh diagonal for the new thickness
    assert np.isfinite(rr) and rel(c.get_state(), w2) < 1e-7
    h3 = 0.8 * h2
    c.set_field("thickness", h3); o.set_fields(h=h3)
    c.solve_state(zero_guess=True)                                   # Jacobi again: clears only its own flag
    c.set_solver(preconditioner=2, rtol=1e-12, maxit=50, check_every=1)
    it, rr = c.solve_state(zero_guess=True)                          # must re-factorise (h3), not reuse the h2 factor
    assert it <= 6 and rel(c.get_state(), o.solve()) < 1e-8
    # density: irrelevant for the static operator (no re-factorisation), part of it once aM != 0
    f0 = c.frontal_info()["factor_ms"]
    c.set_field("density", 2.0 * c.get_field("density"))
    it, rr = c.solve_state(zero_guess=True)
    assert it <= 6
    c.set_operator(0.5, 200.0)
    x = rng.uniform(-1, 1, m.ndof)
    c.set_state(x)
    b1, _, _ = c.solve_linear(x)
    c.set_field("density", 3.0 * c.get_field("density"))            # operator changed: the factor must follow
    b2, it2, _ = c.solve_linear(x)
    assert it2 <= 6 and rel(b1, b2) > 1e-3


@pytest.mark.parametrize("kind,ewm,uhat", [("plate", False, False), ("warped", False, True), ("warped", True, False)])
def test_stress_on_the_mid_and_bottom_surfaces_and_global_component_sums(kind, ewm, uhat):
    """von_Mises_stress(surface='Mid' | 'Bot') (rm_shell_pde.py:153-165) and sum_stress_subdomain (:130-150) against the
    oracle's restatement of ShellStressRM (linear_shell_model.py:393-467); the sums over a tagged sub-domain and over the
    whole mesh."""
    m, o, c, rng = _pair(kind, ewm=ewm, uhat=uhat, beta=1e6)
    w = rng.uniform(-1, 1, m.ndof) * 1e-4
    c.set_state(w)
    for surface, name in (("Top", "stress"), ("Mid", "stress_mid"), ("Bot", "stress_bot")):
        assert rel(c.field_output(name).reshape(m.nel, -1), o.stress_dg1(w, surface)) < 1e-10, surface
    top, bot = c.field_output("stress"), c.field_output("stress_bot")
    assert rel(top, bot) > 1e-3                                   # bending present: the two surfaces differ
    names = ["sum_stress_" + k for k in ("x", "y", "z", "xy", "xz", "yz")]
    ref = o.sum_stress_subdomain(w)
    got = np.array([c.functional(n) for n in names])
    assert np.abs(got - ref).max() < 1e-10 * np.abs(ref).max()
    cells = rng.permutation(m.nel)[: m.nel // 3]
    tags = -np.ones(m.nel, dtype=np.int32); tags[cells] = 0
    c.set_cell_tags(tags, 1)
    c.select_subdomain(0)
    ref = o.sum_stress_subdomain(w, cells)
    got = np.array([c.functional(n) for n in names])
    assert np.abs(got - ref).max() < 1e-10 * np.abs(ref).max()
    c.select_subdomain(-1)
    with pytest.raises(Exception, match="unknown field output"):
        c.field_output("stress_side")


def test_csr_assembly_at_config2():
    """The CSR export at BASELINE config 2 (58 x 290 plate, 255 438 DOF, 18 685 548 stored entries) against the oracle's scipy
    assembly of the same matrix -- the matrix the full-size goldens were solved with."""
    from femo_alpha_amd.backend import ShellContext
    from femo_alpha_amd.mesh import plate_mesh
    from oracle.rm_shell_oracle import ShellOracle, degree4_rule
    m = plate_mesh(2.0, 10.0, 58, 290)
    rng = np.random.default_rng(0)
    h = 0.1 * (1 + 0.2 * rng.uniform(-1, 1, m.nn))
    c = ShellContext(m)
    for k, v in dict(thickness=h, E=[1e8], nu=[0.3]).items():
        c.set_field(k, v)
    info = c.enable_csr()
    assert info["nnz"] == 18685548
    K = c.assemble_csr()
    o = ShellOracle(m)
    o.set_fields(h=h, E=1e8, nu=0.3)
    Kref = o.assemble_K(with_penalty=False, with_strong=False)
    assert Kref.nnz == info["nnz"]
    assert abs(K - Kref).max() < 1e-11 * abs(Kref).max()
    c.close()


def test_pnorm_stress_with_a_given_alpha():
    """pnorm_stress(alpha=...) (rm_shell_pde.py:112-128): the caller's normalisation replaces the reference area."""
    from oracle.rm_shell_oracle import ShellOracle, degree4_rule
    m, o, c, rng = _pair("warped", beta=1e6)
    w = rng.uniform(-1, 1, m.ndof) * 1e-4
    c.set_state(w)
    o3 = ShellOracle(m, nquad=degree4_rule(m))
    o3.set_fields(h=o.h, E=o.E, nu=o.nu, rho=o.rho, f=o.f, uhat=o.uhat)
    c.set_stress_params(1e-6, 6.0)
    c.set_stress_alpha(2.5)
    ref = o3.pnorm_stress(w, 1e-6, 6.0, alpha=2.5)
    assert abs(c.functional("pnorm_stress") - ref) < 1e-10 * ref
    g = c.dfunctional("pnorm_stress", "disp_solid")
    c.set_stress_alpha(None)
    ref0 = o3.pnorm_stress(w, 1e-6, 6.0)
    assert abs(c.functional("pnorm_stress") - ref0) < 1e-10 * ref0
    g0 = c.dfunctional("pnorm_stress", "disp_solid")
    assert rel(g * 2.5, g0 * (ref0 / ref * 2.5)) < 1e-12 or rel(g * ref0, g0 * ref) < 1e-10


def test_penalty_with_prescribed_values():
    """Non-zero Dirichlet data g in the penalty term beta/h_E |J F^-T N| (w - g).v (linear_shell_model.py:323-333): load vector,
    state, residual and the shape sensitivity of the penalty term against the oracle; the state follows g on the clamped edge."""
    m, o, c, rng = _pair("warped", uhat=True)
    g = np.zeros(m.ndof)
    pf = m.penalty_facets(lambda x: np.less(x[1], 1e-12))
    cd = m.cell_dofs()
    edge_cells = np.unique(pf[:, 0])
    touched = np.unique(cd[edge_cells])
    g[touched] = 1e-3 * rng.uniform(-1, 1, touched.size)
    c.set_field("dirichlet", g); o.set_dirichlet_values(g)
    assert rel(c.load_vector(), o.load_vector()) < 1e-11
    c.use_direct_solver(leaf_size=8)
    it, rr = c.solve_state(zero_guess=True)
    w, w0 = c.get_state(), o.solve()
    assert it <= 6 and rel(w, w0) < 1e-8
    # the clamped DOFs (those of the facets' own nodes) follow the prescribed values
    fd = np.unique(np.concatenate([d for d, _ in o._penalty_blocks()]))
    assert np.abs(w[fd] - g[fd]).max() < 1e-6 * np.abs(g[fd]).max()
    assert np.abs(c.residual()).max() < 1e-6 * np.abs(o.apply_K(w0)).max()
    # shape sensitivity of lam . R with the penalty term P(uhat) (w - g): finite difference of the oracle along one direction
    lam = rng.uniform(-1, 1, m.ndof)
    gu = c.dRdarg_T("uhat", lam).reshape(-1, 3)
    u0 = o.uhat.copy()
    d = rng.uniform(-1, 1, u0.shape)
    eps = 1e-6
    def lamR(u):
        # lam . [K_el w + P(u) (w - g) - F]: the penalty term evaluated on the difference (P w and P g are 1e12 each)
        o.set_fields(uhat=u)
        o.set_dirichlet_values(None)
        r = o.apply_K(w, with_penalty=False) - o.load_vector()
        for dofs, blk in o._penalty_blocks():
            r[dofs] += blk @ (w - g)[dofs]
        o.set_dirichlet_values(g)
        return lam @ r
    fdv = (lamR(u0 + eps * d) - lamR(u0 - eps * d)) / (2 * eps)
    o.set_fields(uhat=u0)
    assert abs(np.sum(gu * d) - fdv) < 1e-5 * abs(fdv)
    c.set_field("dirichlet", np.zeros(m.ndof)); o.set_dirichlet_values(None)
    assert rel(c.load_vector(), o.load_vector()) < 1e-11


@pytest.mark.parametrize("ewm,uhat", [(False, True), (True, False)])
def test_pnorm_stress_with_the_thickness_regularisation(ewm, uhat):
    """pnorm_stress(regularization=True) (rm_shell_pde.py:120-122): value and the thickness / shape gradients of the added
    0.5e3 int h^rho J dx term against the oracle (finite differences for the gradients)."""
    from oracle.rm_shell_oracle import ShellOracle, degree4_rule
    m, o, c, rng = _pair("warped", ewm=ewm, uhat=uhat, beta=1e6)
    w = rng.uniform(-1, 1, m.ndof) * 1e-4
    c.set_state(w)
    o3 = ShellOracle(m, element_wise_material=ewm, nquad=degree4_rule(m))
    hh = 1.0 + 0.3 * rng.uniform(-1, 1, o.h.size)                       # thickness ~ 1 so that h^rho is neither 0 nor inf
    o3.set_fields(h=hh, E=o.E, nu=o.nu, rho=o.rho, f=o.f, uhat=o.uhat)
    c.set_field("thickness", hh)
    mval, rho = 1e-6, 6.0
    c.set_stress_params(mval, rho)
    c.set_option("stress_regularization", 0.5e3)
    alpha = ShellOracle(m, element_wise_material=ewm, nquad=degree4_rule(m)).pnorm_stress(w * 0, 1.0, 0.0, alpha=1.0)
    P = lambda: o3.pnorm_stress(w, mval, rho, alpha=alpha, regularization=True)
    ref = P()
    assert abs(c.functional("pnorm_stress") - ref) < 1e-10 * ref
    assert ref > 10 * o3.pnorm_stress(w, mval, rho, alpha=alpha)          # the added term is what is being tested
    g = c.dfunctional("pnorm_stress", "thickness")
    for i in rng.choice(g.size, 3, replace=False):
        v = hh.copy(); st = 1e-6 * v[i]
        v[i] += st; o3.set_fields(h=v); fp = P()
        v[i] -= 2 * st; o3.set_fields(h=v); fm = P()
        o3.set_fields(h=hh)
        assert abs(g[i] - (fp - fm) / (2 * st)) <= 1e-6 * np.abs(g).max()
    if uhat:
        gu = c.dfunctional("pnorm_stress", "uhat").reshape(-1, 3)
        u0 = o3.uhat.copy(); d = rng.uniform(-1, 1, u0.shape); eps = 1e-6
        o3.set_fields(uhat=u0 + eps * d); fp = P()
        o3.set_fields(uhat=u0 - eps * d); fm = P()
        o3.set_fields(uhat=u0)
        assert abs(np.sum(gu * d) - (fp - fm) / (2 * eps)) < 1e-5 * abs((fp - fm) / (2 * eps))
    c.set_option("stress_regularization", 0.0)


@pytest.mark.parametrize("kind,uhat,bc", [("warped", False, "strong"), ("tri", False, "strong"), ("warped", True, "strong"), ("warped", False, "penalty"),
                                          ("tri", True, "penalty")])
def test_cg1cg1_element(kind, uhat, bc):
    """ShellElement 'CG1CG1' (linear_shell_model.py:74-79: displacement AND rotation on the vertices; the reference's RMShellPDE
    never selects it, rm_shell_pde.py:27): the element kernels instantiated with the vertex tables for the displacement, against the
    oracle's CG1CG1 branch -- operator, load, functionals, partial gradients at 1e-11; forward solve and adjoint gradient through the
    multifrontal Cholesky; strong Dirichlet conditions and the penalty clamp (the linear edge block for the displacement too)."""
    from femo_alpha_amd.backend import ShellContext
    from oracle.rm_shell_oracle import ShellOracle, degree4_rule
    base = _mesh(kind)
    m = ShellMesh(base.nodes, base.cells, "CG1CG1")
    assert m.ndof == 6 * m.nn and m.ldof == 6 * m.nvc
    rng = np.random.default_rng(5)
    fields = dict(thickness=0.05 * (1 + 0.3 * rng.uniform(-1, 1, m.nn)), E=3e7 * (1 + 0.2 * rng.uniform(-1, 1, m.nn)),
                  nu=0.3 + 0.05 * rng.uniform(-1, 1, m.nn), density=10 * (1 + 0.1 * rng.uniform(-1, 1, m.nn)),
                  F_solid=rng.uniform(-1, 1, (m.nn, 3)))
    if uhat:
        fields["uhat"] = 0.02 * rng.uniform(-1, 1, (m.nn, 3))
    marker = lambda x: np.less(x[1], 1e-12)
    sd = m.locate_dofs_geometrical(marker) if bc == "strong" else None
    pf = m.penalty_facets(marker) if bc == "penalty" else None
    o = ShellOracle(m, strong_dofs=sd, penalty_facets=pf)
    o.set_fields(h=fields["thickness"], E=fields["E"], nu=fields["nu"], rho=fields["density"], f=fields["F_solid"], uhat=fields.get("uhat"))
    c = ShellContext(m)
    for k, v in fields.items():
        c.set_field(k, v)
    if sd is not None:
        c.set_strong_dofs(sd)
    else:
        c.set_penalty_facets(pf)
    tol = 1e-11
    x = rng.uniform(-1, 1, m.ndof)
    K = o.assemble_K()
    assert rel(c.apply_K(x), K @ x) < tol
    assert rel(c.diagonal(), K.diagonal()) < tol
    assert rel(c.load_vector(), o.load_vector()) < tol
    w = rng.uniform(-1, 1, m.ndof) * 1e-3
    if sd is not None:
        w[o.strong_dofs] = 0.0
    c.set_state(w)
    assert abs(c.functional("compliance") - o.compliance(w)) < tol * abs(o.compliance(w))
    assert abs(c.functional("mass") - o.mass()) < tol * abs(o.mass())
    assert abs(c.functional("elastic_energy") - o.elastic_energy(w)) < tol * abs(o.elastic_energy(w))
    assert rel(c.dfunctional("compliance", "disp_solid"), o.dcompliance_du(w)) < tol
    assert rel(c.dfunctional("compliance", "thickness"), o.dcompliance_dh(w)) < tol
    assert rel(c.dfunctional("elastic_energy", "thickness"), 0.5 * o.dRdfield_T("h", w, w)) < tol
    lam = rng.uniform(-1, 1, m.ndof)
    for arg, name in (("thickness", "h"), ("E", "E"), ("nu", "nu")):
        assert rel(c.dRdarg_T(arg, lam), o.dRdfield_T(name, w, lam)) < tol
    assert rel(c.dRdarg_T("F_solid", lam), o.dRdf_T(lam)) < tol
    # forward + adjoint through the direct solver
    w_ref, J_ref, dJ_ref = o.forward_adjoint()
    c.use_direct_solver(leaf_size=4)
    it, rr = c.solve_state(zero_guess=True)
    assert it <= 4 and rr <= 1e-12
    assert rel(c.get_state(), w_ref) < 1e-8
    assert abs(c.functional("compliance") - J_ref) < 1e-8 * abs(J_ref)
    dJ, it2, _ = c.total_gradient("compliance", "thickness")
    assert it2 <= 4 and rel(dJ, dJ_ref) < 1e-8
    c.close()


@pytest.mark.parametrize("kind", ["warped", "tri"])
def test_cg1cg1_stress_csr_and_shape_outputs(kind):
    """The rest of the operator surface on the CG1CG1 element: p-norm stress aggregate with its partial gradients, the DG1 stress
    field, the CSR export and the shape derivatives (d/d uhat of the outputs and (dR/d uhat)^T lambda), each against the oracle's
    CG1CG1 branch (values) or its central finite differences (shape, stress gradients), the penalty clamp's shape term included."""
    from femo_alpha_amd.backend import ShellContext
    from oracle.rm_shell_oracle import ShellOracle, degree4_rule
    base = _mesh(kind)
    m = ShellMesh(base.nodes, base.cells, "CG1CG1")
    rng = np.random.default_rng(9)
    fields = dict(thickness=0.05 * (1 + 0.3 * rng.uniform(-1, 1, m.nn)), E=3e7 * (1 + 0.2 * rng.uniform(-1, 1, m.nn)),
                  nu=0.3 + 0.05 * rng.uniform(-1, 1, m.nn), density=10 * (1 + 0.1 * rng.uniform(-1, 1, m.nn)),
                  F_solid=rng.uniform(-1, 1, (m.nn, 3)), uhat=0.02 * rng.uniform(-1, 1, (m.nn, 3)))
    sd = m.locate_dofs_geometrical(lambda x: np.less(x[1], 1e-12))

    def oracle(nquad=None):
        o_ = ShellOracle(m, strong_dofs=sd, nquad=nquad)
        o_.set_fields(h=fields["thickness"], E=fields["E"], nu=fields["nu"], rho=fields["density"], f=fields["F_solid"], uhat=fields["uhat"])
        return o_
    o, o3 = oracle(), oracle(degree4_rule(m))                        # o3: the degree-4 measure of the stress aggregate
    c = ShellContext(m)
    for k, v in fields.items():
        c.set_field(k, v)
    c.set_strong_dofs(sd)
    w = rng.uniform(-1, 1, m.ndof) * 1e-4
    w[o.strong_dofs] = 0.0
    lam = rng.uniform(-1, 1, m.ndof)
    lam[o.strong_dofs] = 0.0
    c.set_state(w)
    # CSR export
    info = c.enable_csr()
    Kref = o.assemble_K(with_penalty=False, with_strong=False)
    assert info["nnz"] == Kref.nnz
    assert abs(c.assemble_csr() - Kref).max() < 1e-11 * abs(Kref).max()
    # stress: aggregate, field, partial gradients
    mval, rho = 1e-6, 6.0
    c.set_stress_params(mval, rho)
    assert abs(c.functional("pnorm_stress") - o3.pnorm_stress(w, mval, rho)) < 1e-10 * o3.pnorm_stress(w, mval, rho)
    if kind != "tri":
        assert rel(c.field_output("stress").reshape(m.nel, -1), o.stress_dg1(w)) < 1e-10
    o0 = ShellOracle(m, nquad=degree4_rule(m))
    alpha = o0.pnorm_stress(w * 0, 1.0, 0.0, alpha=1.0)              # reference area (uhat = 0)
    P = lambda ww=w: o3.pnorm_stress(ww, mval, rho, alpha=alpha)
    g_w = c.dfunctional("pnorm_stress", "disp_solid")
    free = np.setdiff1d(np.arange(m.ndof), o.strong_dofs)
    for i in rng.choice(free, 5, replace=False):
        st = 1e-6 * max(abs(w[i]), 1e-4)
        wp = w.copy(); wp[i] += st; wm = w.copy(); wm[i] -= st
        fd = (P(wp) - P(wm)) / (2 * st)
        assert abs(g_w[i] - fd) <= 2e-6 * np.abs(g_w).max() + 1e-7 * abs(fd), (i, g_w[i], fd)
    g_h = c.dfunctional("pnorm_stress", "thickness")
    h0 = o3.h.copy()
    for i in rng.choice(g_h.size, 3, replace=False):
        v = h0.copy(); st = 1e-6 * v[i]
        v[i] += st; o3.set_fields(h=v); fp = P()
        v[i] -= 2 * st; o3.set_fields(h=v); fm = P()
        o3.set_fields(h=h0)
        fd = (fp - fm) / (2 * st)
        assert abs(g_h[i] - fd) <= 5e-6 * np.abs(g_h).max() + 1e-7 * abs(fd), (i, g_h[i], fd)
    # shape derivatives against central differences of the oracle
    g_c = c.dfunctional("compliance", "uhat").reshape(-1, 3)
    g_m = c.dfunctional("mass", "uhat").reshape(-1, 3)
    g_e = c.dfunctional("elastic_energy", "uhat").reshape(-1, 3)
    g_s = c.dfunctional("pnorm_stress", "uhat").reshape(-1, 3)
    g_r = c.dRdarg_T("uhat", lam).reshape(-1, 3)
    u0 = o.uhat.copy()

    def phis():
        return (o.compliance(w), o.mass(), o.elastic_energy(w), P(), lam @ (o.assemble_K(with_strong=False) @ w - o.load_vector()))
    step = 1e-6
    for v in rng.choice(m.nn, 3, replace=False):
        for comp in range(3):
            vals = []
            for sgn in (1, -1):
                u = u0.copy(); u[v, comp] += sgn * step
                o.set_fields(uhat=u); o3.set_fields(uhat=u)
                vals.append(phis())
            o.set_fields(uhat=u0); o3.set_fields(uhat=u0)
            fd = [(a - b) / (2 * step) for a, b in zip(*vals)]
            for g, d, name in ((g_c, fd[0], "compliance"), (g_m, fd[1], "mass"), (g_e, fd[2], "energy"), (g_s, fd[3], "pnorm"), (g_r, fd[4], "residual")):
                assert abs(g[v, comp] - d) <= 5e-6 * np.abs(g).max() + 1e-7 * abs(d), (name, v, comp, g[v, comp], d)
    c.close()
    # the shape derivative of the penalty clamp (linear edge block for the displacement)
    pf = m.penalty_facets(lambda x: np.less(x[1], 1e-12))
    op = ShellOracle(m, penalty_facets=pf, beta=1e6)
    op.set_fields(h=fields["thickness"], E=fields["E"], nu=fields["nu"], rho=fields["density"], f=fields["F_solid"], uhat=fields["uhat"])
    c = ShellContext(m)
    for k, v in fields.items():
        c.set_field(k, v)
    c.set_penalty_facets(pf, 1e6)
    w = rng.uniform(-1, 1, m.ndof) * 1e-3
    lam = rng.uniform(-1, 1, m.ndof)
    c.set_state(w)
    g_r = c.dRdarg_T("uhat", lam).reshape(-1, 3)
    on_clamp = np.nonzero(np.abs(m.nodes[:, 1]) < 1e-12)[0]
    for v in list(rng.choice(on_clamp, 2, replace=False)) + list(rng.choice(m.nn, 1)):
        for comp in range(3):
            vals = []
            for sgn in (1, -1):
                u = u0.copy(); u[v, comp] += sgn * step
                op.set_fields(uhat=u)
                vals.append(lam @ (op.assemble_K() @ w - op.load_vector()))
            op.set_fields(uhat=u0)
            d = (vals[0] - vals[1]) / (2 * step)
            assert abs(g_r[v, comp] - d) <= 5e-6 * np.abs(g_r).max() + 1e-7 * abs(d), ("penalty", v, comp, g_r[v, comp], d)
    c.close()


@pytest.mark.parametrize("kind,uhat,bc,ewm", [("tri", False, "strong", False), ("tri", True, "penalty", False), ("delaunay", False, "penalty", False),
                                              ("delaunay", True, "strong", True)])
def test_cg2cr1_element(kind, uhat, bc, ewm):
    """ShellElement 'CG2CR1' (linear_shell_model.py:68-73; triangles; the reference's RMShellPDE never selects it, rm_shell_pde.py:27):
    displacement on the P2 nodes, rotation on the EDGE MIDPOINTS with the Crouzeix-Raviart functions.  In the library the rotation's
    shape tables part from the tables of the geometry and the nodal fields (Tables::NR / dNR) and a rotation node is the cell's P2 node
    3 + k; the penalty clamp gets a 3 x 3 rotation block per facet (a Crouzeix-Raviart trace involves all three functions of the cell).
    Against the oracle's CG2CR1 branch -- which builds its B matrices from 1 - 2 L_(k+2) directly: operator, diagonal, load, functionals,
    partial gradients, stress outputs, CSR export at 1e-11; forward solve and adjoint gradient through the multifrontal Cholesky at 1e-8
    (the transient march: tests/test_gpu_dynamic.py); shape derivatives against finite differences of the oracle; element partitions are
    refused with a message."""
    from femo_alpha_amd.backend import FemoHipError, ShellContext
    from oracle.rm_shell_oracle import ShellOracle, degree4_rule
    base = _mesh(kind)
    m = ShellMesh(base.nodes, base.cells, "CG2CR1")
    assert m.nR == m.nE and m.ndof == 3 * (m.nV + m.nE) + 3 * m.nE and m.ldof == 27
    with pytest.raises(ValueError, match="Invalid element type"):
        ShellMesh(_mesh("warped").nodes, _mesh("warped").cells, "CG2CR1")          # a simplex element
    rng = np.random.default_rng(6)
    nT = m.nel if ewm else m.nn
    fields = dict(thickness=0.05 * (1 + 0.3 * rng.uniform(-1, 1, nT)), E=3e7 * (1 + 0.2 * rng.uniform(-1, 1, nT)),
                  nu=0.3 + 0.05 * rng.uniform(-1, 1, nT), density=10 * (1 + 0.1 * rng.uniform(-1, 1, nT)),
                  F_solid=rng.uniform(-1, 1, (m.nn, 3)))
    if uhat:
        fields["uhat"] = 0.02 * rng.uniform(-1, 1, (m.nn, 3))
    marker = lambda x: np.less(x[1], 1e-12)
    sd = m.locate_dofs_geometrical(marker) if bc == "strong" else None
    pf = m.penalty_facets(marker) if bc == "penalty" else None
    # (penalty 1e10, not the reference's 1e15: the oracle's float64 LU of a 1e15-penalised operator is itself only good to ~2e-7 on this
    #  element -- its 3 x 3 rotation blocks have rank 2 -- and the forward solve below is compared with it at 1e-8; the blocks themselves
    #  are checked entry by entry through apply_K and the diagonal, whatever beta is)
    beta = 1e10
    o = ShellOracle(m, element_wise_material=ewm, strong_dofs=sd, penalty_facets=pf, beta=beta)
    o.set_fields(h=fields["thickness"], E=fields["E"], nu=fields["nu"], rho=fields["density"], f=fields["F_solid"], uhat=fields.get("uhat"))
    c = ShellContext(m, element_wise_material=ewm)
    assert c.ndof == m.ndof
    for k, v in fields.items():
        c.set_field(k, v)
    if sd is not None:
        c.set_strong_dofs(sd)
    else:
        c.set_penalty_facets(pf, beta)
    tol = 1e-11
    x = rng.uniform(-1, 1, m.ndof)
    K = o.assemble_K()
    assert rel(c.apply_K(x), K @ x) < tol
    assert rel(c.diagonal(), K.diagonal()) < tol
    assert rel(c.load_vector(), o.load_vector()) < tol
    Ke = c.element_matrices(0, 3)
    assert rel(Ke, o.element_matrices(slice(0, 3))) < tol
    w = rng.uniform(-1, 1, m.ndof) * 1e-3
    if sd is not None:
        w[o.strong_dofs] = 0.0
    c.set_state(w)
    assert abs(c.functional("compliance") - o.compliance(w)) < tol * abs(o.compliance(w))
    assert abs(c.functional("mass") - o.mass()) < tol * abs(o.mass())
    assert abs(c.functional("elastic_energy") - o.elastic_energy(w)) < tol * abs(o.elastic_energy(w))
    assert rel(c.dfunctional("compliance", "disp_solid"), o.dcompliance_du(w)) < tol
    assert rel(c.dfunctional("compliance", "thickness"), o.dcompliance_dh(w)) < tol
    assert rel(c.dfunctional("elastic_energy", "thickness"), 0.5 * o.dRdfield_T("h", w, w)) < tol
    lam = rng.uniform(-1, 1, m.ndof)
    for arg, name in (("thickness", "h"), ("E", "E"), ("nu", "nu")):
        assert rel(c.dRdarg_T(arg, lam), o.dRdfield_T(name, w, lam)) < tol
    assert rel(c.dRdarg_T("F_solid", lam), o.dRdf_T(lam)) < tol
    # the stress outputs interpolate the rotation too: the p-norm aggregate of the top-surface von Mises stress
    os_ = ShellOracle(m, element_wise_material=ewm, nquad=degree4_rule(m))
    os_.set_fields(h=fields["thickness"], E=fields["E"], nu=fields["nu"], rho=fields["density"], uhat=fields.get("uhat"))
    c.set_stress_params(1e-6, 6.0)
    assert abs(c.functional("pnorm_stress") - os_.pnorm_stress(w, 1e-6, 6.0)) < 1e-10 * abs(os_.pnorm_stress(w, 1e-6, 6.0))
    # shape derivatives (dual-number kernels; the penalty clamp's 3-node rotation trace included) against central finite differences of
    # the oracle, where the mesh moves (uhat != 0)
    if uhat:
        u0 = o.uhat.copy()
        g_e = c.dfunctional("elastic_energy", "uhat").reshape(-1, 3)
        g_r = c.dRdarg_T("uhat", lam).reshape(-1, 3)
        phis = lambda: (o.elastic_energy(w), lam @ (o.assemble_K(with_strong=False) @ w - o.load_vector()))
        for v in rng.choice(m.nn, 2, replace=False):
            for comp in range(3):
                up = u0.copy(); up[v, comp] += 1e-6
                um = u0.copy(); um[v, comp] -= 1e-6
                o.set_fields(uhat=up); fp = phis()
                o.set_fields(uhat=um); fm = phis()
                o.set_fields(uhat=u0)
                for name, g, d in (("energy", g_e, (fp[0] - fm[0]) / 2e-6), ("residual", g_r, (fp[1] - fm[1]) / 2e-6)):
                    assert abs(g[v, comp] - d) <= 2e-6 * np.abs(g).max() + 1e-9 * abs(d), (name, v, comp, g[v, comp], d)
    # ghost entries (element partitions, round 6): three extra entries behind [u | theta(edges)], untouched by the element kernels
    cg = ShellContext(m, element_wise_material=ewm, nghost=3)
    assert cg.ndof == m.ndof + 3
    for k, v in fields.items():
        cg.set_field(k, v)
    if sd is not None:
        cg.set_strong_dofs(sd)
    else:
        cg.set_penalty_facets(pf, beta)
    xg = np.concatenate([w, [1.0, 2.0, 3.0]])
    assert np.abs(cg.apply_K(xg)[:m.ndof] - c.apply_K(w)).max() < 1e-13 * np.abs(c.apply_K(w)).max()
    cg.close()
    # the CSR export (pattern built on the device from the edge-midpoint rotation nodes) against the oracle's assembly
    info = c.enable_csr()
    Kel = o.assemble_K(with_penalty=False, with_strong=False)
    assert info["nnz"] == Kel.nnz and abs(c.assemble_csr() - Kel).max() < 1e-11 * abs(Kel).max()
    # forward + adjoint through the direct solver
    w_ref, J_ref, dJ_ref = o.forward_adjoint()
    c.use_direct_solver(leaf_size=4)
    it, rr = c.solve_state(zero_guess=True)
    assert it <= 4 and rr <= 1e-12
    assert rel(c.get_state(), w_ref) < 1e-8
    assert abs(c.functional("compliance") - J_ref) < 1e-8 * abs(J_ref)
    dJ, it2, _ = c.total_gradient("compliance", "thickness")
    assert it2 <= 4 and rel(dJ, dJ_ref) < 1e-8
    # and the matrix-free Jacobi-PCG agrees with it (the Krylov path of the north star on this element)
    c.set_solver(preconditioner=0, rtol=1e-12, maxit=200000, check_every=50)
    it3, rr3 = c.solve_state(zero_guess=True)
    assert rr3 <= 1e-12 and rel(c.get_state(), w_ref) < 1e-7
    c.close()
