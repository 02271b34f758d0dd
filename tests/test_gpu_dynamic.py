"""Transient path (BASELINE config 5 family): PlateSim march and its adjoint on the GPU against the CPU oracle's
restatement of the reference's time discretisation (parity unpinned: see femo_alpha_amd/dynamic_rm_shell/plate_sim.py)."""
import numpy as np
import pytest

from femo_alpha_amd import csdl
from femo_alpha_amd.mesh import plate_mesh

pytestmark = pytest.mark.gpu


def _gust(time_levels, nn, dt):
    # 1-cosine gust of ex_simple_dynamic_shell_opt.py:45-95 (V_p = 50, T0 = 0.02, T1 = 0.12), scaled
    t = np.arange(time_levels) * dt
    fz = np.where((t >= 0.02) & (t <= 0.14), 0.1 * 50 * (1 - np.cos(2 * np.pi * (t - 0.02) / 0.12)), 0.0)
    F = np.zeros((time_levels, nn, 3))
    F[:, :, 2] = fz[:, None]
    return F.reshape(time_levels, -1)


@pytest.mark.parametrize("ewt,element", [(False, "CG2CG1"), (True, "CG2CG1"), (False, "CG1CG1"), (False, "CG2CR1")])
def test_march_and_adjoint(ewt, element):
    from femo_alpha_amd.dynamic_rm_shell.operations import StateOperation, TotalStrainEnergyOperation, VolumeOperation
    from femo_alpha_amd.dynamic_rm_shell.plate_sim import PlateSim
    from oracle.rm_shell_oracle import ShellOracle
    mesh = plate_mesh(2.0, 10.0, 4, 12)
    if element == "CG2CR1":                 # Crouzeix-Raviart rotations (linear_shell_model.py:68-73): a simplex element
        from femo_alpha_amd.mesh import ShellMesh, quads_to_triangles
        tri = quads_to_triangles(mesh)
        mesh = ShellMesh(tri.nodes, tri.cells, element)
    elif element != "CG2CG1":               # the other quadrilateral element of ShellElement (linear_shell_model.py:74-79)
        from femo_alpha_amd.mesh import ShellMesh
        mesh = ShellMesh(mesh.nodes, mesh.cells, element)
    E, nu, rho, dt, N = 1e8, 0.3, 10.0, 0.01, 12
    ps = PlateSim(mesh, E, nu, rho, dt, N, element_wise_thickness=ewt, quad_deg=3, leaf_size=8)
    n_t = mesh.nel if ewt else mesh.nn
    rng = np.random.default_rng(0)
    t0 = 0.1 * (1 + 0.2 * rng.uniform(-1, 1, n_t))
    F = _gust(N + 1, mesh.nn, dt)

    o = ShellOracle(mesh, element_wise_material=ewt, strong_dofs=ps.bc_dofs, nred=2)
    def march(t):
        o.set_fields(h=t, E=E, nu=nu, rho=rho)
        return o.dynamic_history(F.reshape(N + 1, -1, 3), dt, N)
    W_ref = march(t0)

    rec = csdl.Recorder(inline=True); rec.start()
    grp = csdl.VariableGroup()
    grp.thickness = csdl.Variable(value=t0, name="thickness")
    grp.force_history = csdl.Variable(value=F, name="force_history")
    hist = StateOperation(ps).evaluate(grp)
    grp.disp_history = hist
    tse = TotalStrainEnergyOperation(ps).evaluate(grp)
    vol = VolumeOperation(ps).evaluate(grp)
    rec.stop()

    W = hist.value.reshape((mesh.ndof, N + 1), order="F")
    assert np.abs(W[:, 0]).max() == 0.0
    assert np.abs(W - W_ref).max() < 1e-8 * np.abs(W_ref).max()
    assert all(it <= 4 for it, rr in ps.solve_info)
    Kel = o.assemble_K(with_strong=False)
    tse_ref = sum(0.5 * W_ref[:, i] @ (Kel @ W_ref[:, i]) for i in range(N + 1))
    assert abs(tse.value[0] - tse_ref) < 1e-8 * tse_ref
    o.set_fields(rho=1.0); vref = o.mass(); o.set_fields(rho=rho)
    assert abs(vol.value[0] - vref) < 1e-12 * vref

    # adjoint of the total strain energy w.r.t. thickness against central finite differences of the oracle march
    g = rec.compute_totals(tse, grp.thickness)
    def J(t):
        Wt = march(t)
        o.set_fields(h=t)
        K = o.assemble_K(with_strong=False)
        return sum(0.5 * Wt[:, i] @ (K @ Wt[:, i]) for i in range(N + 1))
    for i in rng.choice(n_t, 3, replace=False):
        st = 1e-4 * t0[i]
        tp = t0.copy(); tp[i] += st; tm = t0.copy(); tm[i] -= st
        fd = (J(tp) - J(tm)) / (2 * st)
        assert abs(g[i] - fd) < 2e-5 * np.abs(g).max(), (i, g[i], fd)
    gv = rec.compute_totals(vol, grp.thickness)
    assert abs(gv.sum() - 20.0) < 1e-9                     # d/dt int t dx summed over the partition of unity = area
    # sensitivity to the load history through the same adjoint
    gF = rec.compute_totals(tse, grp.force_history)
    k, node = 5, mesh.nn // 2
    Fp = F.copy(); Fp[k, 3 * node + 2] += 1e-3
    def JF(Fh):
        Wt = o.dynamic_history(Fh.reshape(N + 1, -1, 3), dt, N)
        return sum(0.5 * Wt[:, i] @ (Kel @ Wt[:, i]) for i in range(N + 1))
    o.set_fields(h=t0)
    Fm = F.copy(); Fm[k, 3 * node + 2] -= 1e-3
    fd = (JF(Fp) - JF(Fm)) / 2e-3
    assert abs(gF[k, 3 * node + 2] - fd) < 1e-5 * np.abs(gF).max()


def test_self_weight_with_elementwise_thickness_and_postprocessing():
    """Self weight f_d = (0, 0, rho t g) (reference plate_sim.py:203-214) with element-wise thickness -- a cell-wise
    constant pressure applied as consistent P2 loads -- against the nodal-thickness path on a uniform plate (identical
    loads, so identical histories), its thickness gradient against finite differences of the march, and the
    post-processing calls of the gust examples (plate_sim.py:427-480)."""
    from femo_alpha_amd.dynamic_rm_shell.plate_sim import PlateSim
    mesh = plate_mesh(2.0, 10.0, 4, 12)
    E, nu, rho, dt, N = 1e8, 0.3, 10.0, 0.01, 8
    F = np.zeros((N + 1, 3 * mesh.nn))
    hist = {}
    for ewt in (False, True):
        ps = PlateSim(mesh, E, nu, rho, dt, N, element_wise_thickness=ewt, add_self_weight=True, quad_deg=3, leaf_size=8)
        ps.update_f_history(F)
        ps.update_t(np.full(mesh.nel if ewt else mesh.nn, 0.1))
        hist[ewt] = ps.solve_dynamic_problem()
        sims = ps if ewt else None
    assert np.abs(hist[False]).max() > 0
    assert np.abs(hist[True] - hist[False]).max() < 1e-10 * np.abs(hist[False]).max()
    # thickness gradient of J = sum_i U_i with the self weight depending on t, element-wise
    ps = sims
    rng = np.random.default_rng(1)
    t0 = 0.1 * (1 + 0.2 * rng.uniform(-1, 1, mesh.nel))

    def J(t):
        ps.update_t(t)
        ps.solve_dynamic_problem()
        return ps.energy_audit()[0].sum()
    J(t0)
    G = np.zeros((mesh.ndof, N + 1)); g_exp = np.zeros(mesh.nel)
    for i in range(N + 1):
        gt, gw = ps.strain_energy_gradients(ps.W[i].cpu().numpy())
        g_exp += gt; G[:, i] = gw
    g = g_exp - ps.residual_T_products(ps.adjoint_history(G))[0]
    d = rng.uniform(0, 1, mesh.nel)
    eps = 1e-6
    fd = (J(t0 + eps * d) - J(t0 - eps * d)) / (2 * eps)
    assert abs(g @ d - fd) < 2e-6 * abs(fd)
    # post-processing of a time level
    ps.update_t(t0); ps.solve_dynamic_problem()
    p_last, p_mid = ps.pnorm_stress(), ps.pnorm_stress(level=N // 2)
    assert p_last > 0 and p_mid > 0 and p_last != p_mid
    vm = ps.von_Mises_stress(level=N)
    assert vm.shape == (4 * mesh.nel,) and vm.max() > 0 and vm.mean() > 0      # an L2 projection onto DG1: vertex values may undershoot
    A = ps.construct_force_to_pressure_map()
    assert A.shape == (3 * mesh.nn, 3 * mesh.nn) and abs(A.sum() - 3 * 20.0) < 1e-10        # three components x plate area
    D = ps.construct_nodal_disp_map()
    w_last = ps.W[N].cpu().numpy()
    assert np.array_equal(D @ w_last, np.concatenate([w_last[c:3 * mesh.nn:3] for c in range(3)]))


@pytest.mark.parametrize("ewt,self_weight", [(False, False), (True, True), (False, True)])
def test_forward_mode_of_the_transient_operator(ewt, self_weight):
    """Forward mode (state_operation_dynamic.py:228-329, 534-605): the Jacobian-vector product of the whole-history residual
    against finite differences of the march's own residual identity, the tangent solve as its inverse, and forward against
    reverse mode through the dot-product test  <Lambda, J dY> = <J^T Lambda, dY>,  <Lambda, (dR/dt) dt> = <(dR/dt)^T Lambda, dt>."""
    from femo_alpha_amd.dynamic_rm_shell.operations import StateOperation
    from femo_alpha_amd.dynamic_rm_shell.plate_sim import PlateSim
    mesh = plate_mesh(2.0, 10.0, 4, 12)
    E, nu, rho, dt, N = 1e8, 0.3, 10.0, 0.01, 8
    ps = PlateSim(mesh, E, nu, rho, dt, N, element_wise_thickness=ewt, add_self_weight=self_weight, quad_deg=3, leaf_size=8, rtol=1e-12)
    n_t = mesh.nel if ewt else mesh.nn
    rng = np.random.default_rng(3)
    t0 = 0.1 * (1 + 0.2 * rng.uniform(-1, 1, n_t))
    F = _gust(N + 1, mesh.nn, dt)
    ps.update_f_history(F)

    def march(t, Fh=F):
        ps.update_f_history(Fh); ps.update_t(t)
        return ps.solve_dynamic_problem()
    # tangent of the march: dW/dt . d  =  - J^-1 (dR/dt) d   against central differences of the march itself.
    # The check must not depend on where PCG happens to stop (with a tolerance-controlled stop the rounding of a
    # march, amplified by the Newmark recursion and by 1 / eps, jumped between 1e-9 and 3e-6 with the iteration count): every solve of
    # the differenced marches runs a FIXED number of iterations (three applications of the exact factor: the residual is at the
    # rounding floor after two), and the two central differences at eps and eps / 2 are combined by Richardson extrapolation, so the
    # truncation error is O(eps^4) and the step need not be tuned against the rounding.
    d = rng.uniform(-1, 1, n_t) * t0
    ps.ctx.set_option("strict", 0)
    ps.ctx.set_solver(preconditioner=2, rtol=1e-300, maxit=3, check_every=1)
    cd = lambda e: (march(t0 + e * d) - march(t0 - e * d)) / (2 * e)
    eps = 2e-4
    fd1, fd2 = cd(eps), cd(eps / 2)
    fd = (4.0 * fd2 - fd1) / 3.0
    print(f"forward mode [{ewt}, {self_weight}]: central differences at {eps:g} and {eps / 2:g} differ by {np.abs(fd1 - fd2).max() / np.abs(fd).max():.1e}")
    W0 = march(t0)
    dRdt_d = ps.jacobian_products_fwd(dthickness=d)
    assert np.abs(dRdt_d[ps.bc_dofs]).max() == 0.0 and np.abs(dRdt_d[:, 0]).max() == 0.0
    dW = -ps.tangent_history(dRdt_d)
    print(f"forward mode [{ewt}, {self_weight}]: tangent against the extrapolated difference {np.abs(dW - fd).max() / np.abs(fd).max():.1e}")
    assert np.abs(dW - fd).max() < 2e-9 * np.abs(fd).max()          # measured 5e-11 .. 2e-10 (rounds 3-5 asserted 2e-6 with a tuned step)
    # the same for a perturbation of the load history
    dFh = rng.uniform(-1, 1, F.shape)
    fdF = (march(t0, F + 1e-3 * dFh) - march(t0, F - 1e-3 * dFh)) / 2e-3
    march(t0)
    dWF = -ps.tangent_history(ps.jacobian_products_fwd(dF=dFh))
    assert np.abs(dWF - fdF).max() < 1e-7 * np.abs(fdF).max()
    # J J^-1 = identity
    R = rng.uniform(-1, 1, W0.shape)
    back = ps.jacobian_products_fwd(dY=ps.tangent_history(R))
    defect = np.abs(back - R).max() / np.abs(R).max()
    assert defect < 1e-6          # not 1e-8: see below
    if not ewt and not self_weight:
        # WHY not 1e-8, tested.  J (J^-1 R) is evaluated level by level as a difference of terms of size
        # |A| |dY| that cancel down to R: its rounding is eps times the condition number of the step operator A = 2/dt^2 M + K/2.  The mass
        # shift regularises K/2, so the defect GROWS with the time step (round 3's comment had it falling like 2/dt^2, which this
        # measurement refuted): 1.1e-7 at dt = 0.01, 2.5e-7 at dt = 0.04.
        ps4 = PlateSim(mesh, E, nu, rho, 4 * dt, N, element_wise_thickness=ewt, add_self_weight=self_weight, quad_deg=3, leaf_size=8, rtol=1e-12)
        ps4.update_f_history(F); ps4.update_t(t0); ps4.solve_dynamic_problem()
        back4 = ps4.jacobian_products_fwd(dY=ps4.tangent_history(R))
        defect4 = np.abs(back4 - R).max() / np.abs(R).max()
        print(f"J J^-1 - I: {defect:.2e} at dt = {dt}, {defect4:.2e} at dt = {4 * dt}")
        assert defect < defect4 < 1e-5, (defect, defect4)
        ps4.ctx.close()
    # forward against reverse mode
    # (on the free rows: the adjoint keeps the Dirichlet entries of Lambda at zero -- they multiply rows of dR/dt and dR/df that are
    #  zero -- so the transpose relation is the one of the free-free blocks)
    Y, L = rng.uniform(-1, 1, W0.shape), rng.uniform(-1, 1, W0.shape)
    Y[ps.bc_dofs] = 0.0; L[ps.bc_dofs] = 0.0
    JY = ps.jacobian_products_fwd(dY=Y)
    Lam = ps.adjoint_history(L)                       # (J^T)^-1 L
    lhs = np.sum(Lam * JY)                            # <J^-T L, J Y> = <L, Y>
    assert abs(lhs - np.sum(L * Y)) < 1e-8 * abs(np.sum(L * Y)) + 1e-10 * np.abs(L).sum()
    g_t, g_f = ps.residual_T_products(Lam)
    assert abs(np.sum(Lam * dRdt_d) - g_t @ d) < 1e-8 * abs(g_t @ d)
    assert abs(np.sum(Lam * ps.jacobian_products_fwd(dF=dFh)) - np.sum(g_f * dFh)) < 1e-8 * abs(np.sum(g_f * dFh))
    # operator surface: mode='fwd' no longer raises
    op = StateOperation(ps)
    out = {op.state_name: None}
    op.apply_inverse_jacobian({}, {}, out, {op.state_name: R.ravel(order="F")}, "fwd")
    ref = ps.tangent_history(R)          # (the sweeps of the factor add with atomics: repeatable to rounding, not bit for bit)
    assert np.abs(out[op.state_name].reshape(W0.shape, order="F") - ref).max() < 1e-9 * np.abs(ref).max()
