"""Several right-hand sides through one pair of triangular sweeps (round 6: femo_solve_linear_multi, femo_total_gradients;
csrc/sweeps_multi.h).  The reference registers compliance, elastic_energy, pnorm_stress and one pnorm_stress_<tag> per sub-domain on
`disp_solid` (rm_shell_model.py:221-253) and solves one adjoint per output (state_operation.py:188-220); here their right-hand sides
share the sweeps in groups of up to four with the vectors interleaved.  Checked: against the oracle (nrhs 2 and 4), against the
one-at-a-time entry points (every group size, every sweep kernel: small-front levels, wide levels with both boundary forms), and at
BASELINE config 3's size."""
import numpy as np
import pytest

from femo_alpha_amd.mesh import plate_mesh, quads_to_triangles, wing_skin_mesh

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


def clamp(x):
    return np.less(x[1], 1e-12)


def _setup(m, leaf=None, **options):
    from femo_alpha_amd.backend import ShellContext
    rng = np.random.default_rng(5)
    fields = dict(thickness=0.05 * (1 + 0.3 * rng.uniform(-1, 1, m.nn)), E=3e7 * (1 + 0.2 * rng.uniform(-1, 1, m.nn)), nu=[0.3],
                  density=[10.0], F_solid=rng.uniform(-1, 1, (m.nn, 3)))
    c = ShellContext(m)
    for k, v in fields.items():
        c.set_field(k, v)
    c.set_penalty_facets(m.penalty_facets(clamp))
    c.enable_frontal(leaf, **options)
    c.set_solver(preconditioner=2, rtol=1e-12, maxit=30, check_every=1)
    return c, fields, rng


@pytest.mark.parametrize("kind", ["quad", "tri"])
@pytest.mark.parametrize("nrhs", [2, 4])
def test_multi_rhs_solves_against_the_oracle(kind, nrhs):
    """x_i = K^-1 b_i for nrhs = 2 and 4 random right-hand sides against the oracle's LU (SuperLU + one refinement step), 1e-8."""
    from oracle.rm_shell_oracle import ShellOracle
    m = wing_skin_mesh(10, 24)
    if kind == "tri":
        m = quads_to_triangles(m)
    c, fields, rng = _setup(m, leaf=4)
    o = ShellOracle(m, penalty_facets=m.penalty_facets(clamp))
    o.set_fields(h=fields["thickness"], E=fields["E"], nu=0.3, rho=10.0, f=fields["F_solid"])
    o.solve()
    B = rng.uniform(-1, 1, (nrhs, m.ndof))
    X, it, rr = c.solve_linear_multi(B)
    assert X.shape == B.shape and np.all(it <= 4) and np.all(rr <= 1e-12)
    for i in range(nrhs):
        assert rel(X[i], o.solve_adjoint(B[i])) < 1e-8, i
    c.close()


@pytest.mark.parametrize("options", [{"wide_np": 100000, "wide_cnt": 0}, {}, {"bnd_tiled_nb": 0}])
def test_multi_rhs_solves_equal_separate_solves(options):
    """Every group size (1 .. 6 right-hand sides: groups of 4 + 2, 3 riding as 4, ...) against femo_solve_linear one at a time: the same
    PCG recurrences, so the same iteration counts and the same solutions to the rounding of a different summation order.  ``options``
    route the sweeps through every multi-vector kernel: no level wide (one workgroup per front on every level), the default rule (on a
    mesh of this size every level has at most 512 fronts and is wide: tiled products with X and L21, the column-block form of L21^T x),
    and all levels wide with the tiled transposed product."""
    m = wing_skin_mesh(16, 40)
    pre = {k: v for k, v in options.items() if k in ("wide_np", "wide_cnt")}
    c, fields, rng = _setup(m, leaf=6, **pre)
    for k, v in options.items():
        if k not in pre:
            c.set_option(k, v)
    B = rng.uniform(-1, 1, (6, m.ndof))
    ref = [c.solve_linear(b) for b in B]
    for nrhs in (1, 2, 3, 4, 5, 6):
        X, it, rr = c.solve_linear_multi(B[:nrhs])
        for i in range(nrhs):
            assert it[i] == ref[i][1], (nrhs, i, it, [r[1] for r in ref])
            assert rel(X[i], ref[i][0]) < 1e-11, (nrhs, i)
            assert rr[i] <= 1e-12
    # the per-level timing of the grouped sweeps (femo_sweep_profile_multi): one forward and one backward time per tree level
    for nr in (2, 4):
        prof = c.sweep_profile_multi(nr)
        assert prof.shape == (c.plan.nlevels, 2) and np.all(prof >= 0) and prof.sum() > 0
    from femo_alpha_amd._lib import FemoHipError
    with pytest.raises(FemoHipError, match="2 or 4"):
        c.sweep_profile_multi(3)
    # option "multi_rhs" 0: one at a time through the same entry point
    c.set_option("multi_rhs", 0)
    X, it, rr = c.solve_linear_multi(B[:3])
    for i in range(3):
        assert rel(X[i], ref[i][0]) < 1e-13 and it[i] == ref[i][1]
    # a zero right-hand side among the others
    c.set_option("multi_rhs", 1)
    B0 = B[:4].copy(); B0[2] = 0.0
    X, it, rr = c.solve_linear_multi(B0)
    assert np.all(X[2] == 0.0) and it[2] == 0 and rel(X[3], ref[3][0]) < 1e-11
    c.close()


def test_total_gradients_of_the_registered_outputs():
    """d J_i / d thickness for the outputs the reference registers on `disp_solid` -- compliance, elastic_energy, pnorm_stress and two
    per-tag aggregates (rm_shell_model.py:221-253) -- by ONE grouped adjoint solve, against femo_total_gradient one functional at a time,
    and (compliance) against the oracle."""
    from oracle.rm_shell_oracle import ShellOracle
    m = wing_skin_mesh(12, 30)
    c, fields, rng = _setup(m, leaf=6)
    tags = -np.ones(m.nel, dtype=np.int32)
    x = m.nodes[m.cells].mean(axis=1)
    tags[x[:, 1] < np.median(x[:, 1])] = 0
    tags[x[:, 1] >= np.percentile(x[:, 1], 75)] = 1
    c.set_cell_tags(tags, 2)
    c.set_stress_params(m=1e-6, rho=6.0)
    it, rr = c.solve_state(zero_guess=True)
    names = ["compliance", "elastic_energy", "pnorm_stress", "pnorm_stress", "pnorm_stress"]
    subs = [-1, -1, -1, 0, 1]
    for arg in ("thickness", "E"):
        G, its, rrs = c.total_gradients(names, arg, subs)
        assert np.all(its <= 4) and np.all(rrs <= 1e-12)
        for i, (nm, sd) in enumerate(zip(names, subs)):
            c.select_subdomain(sd)
            g, it1, _ = c.total_gradient(nm, arg)
            assert rel(G[i], g) < 1e-10, (arg, nm, sd)
        c.select_subdomain(-1)
    o = ShellOracle(m, penalty_facets=m.penalty_facets(clamp))
    o.set_fields(h=fields["thickness"], E=fields["E"], nu=0.3, rho=10.0, f=fields["F_solid"])
    _, _, dJ = o.forward_adjoint()
    G, _, _ = c.total_gradients(names, "thickness", subs)
    assert rel(G[0], dJ) < 1e-8
    assert rel(G[3], G[4]) > 1e-3                      # the two sub-domains are different functionals
    # the selection of the context is left as it was
    c.select_subdomain(1)
    c.total_gradients(names[:2], "thickness")
    assert abs(c.functional("pnorm_stress") - c.functional("pnorm_stress")) == 0.0
    g_sel, _, _ = c.total_gradient("pnorm_stress", "thickness")
    assert rel(g_sel, G[4]) < 1e-10
    c.close()


def test_multi_rhs_at_config3_size():
    """BASELINE config 3 (1 015 470 DOF): four adjoint right-hand sides through the grouped sweeps against four separate solves --
    the same iteration counts, the same gradients (1e-9) -- and the time of the grouped call printed beside four single ones."""
    import time
    from bench import make_workload
    from femo_alpha_amd.backend import ShellContext
    m, fields, marker, _ = make_workload("wing1m")
    c = ShellContext(m)
    for k, v in fields.items():
        c.set_field(k, v)
    c.set_penalty_facets(m.penalty_facets(marker))
    c.use_direct_solver()
    c.solve_state(zero_guess=True)
    c.set_stress_params(m=1e-6, rho=6.0)
    names = ["compliance", "elastic_energy", "pnorm_stress", "tip_disp"]
    c.total_gradients(names, "thickness")              # warm-up (buffers)
    t0 = time.perf_counter()
    G, its, rrs = c.total_gradients(names, "thickness")
    t_multi = time.perf_counter() - t0
    t0 = time.perf_counter()
    single = [c.total_gradient(nm, "thickness") for nm in names]
    t_single = time.perf_counter() - t0
    for i, (g, it1, rr1) in enumerate(single):
        assert its[i] == it1 and rel(G[i], g) < 1e-9, (names[i], its, it1)
    print(f"wing1m: 4 total gradients grouped {t_multi * 1e3:.2f} ms, one at a time {t_single * 1e3:.2f} ms (host wall-clock incl. copies)")
    assert t_multi < 0.8 * t_single
    c.close()


def test_operator_surface_takes_the_seeds_of_several_outputs():
    """StateOperation.apply_inverse_jacobian (state_operation.py:188-220) with d_outputs[state] of shape (k, ndof): the seeds of k outputs
    in one grouped solve, Dirichlet rows zeroed in every one -- the same vectors as k separate calls, in both modes."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    from femo_alpha_amd import csdl
    from femo_alpha_amd.csdl_alpha_opt.state_operation import StateOperation
    from femo_alpha_amd.rm_shell.rm_shell_model import RMShellModel
    mesh = plate_mesh(2.0, 10.0, 6, 24)
    nn = mesh.nn
    marker = lambda x: np.less(x[0], 3e-16)
    rng = np.random.default_rng(2)
    recorder = csdl.Recorder(inline=True)
    recorder.start()
    pressure = csdl.Variable(value=np.tile([0.0, 0.0, 5.0], (nn, 1)), name="force_vector")
    thickness = csdl.Variable(value=0.1 * (1 + 0.2 * rng.uniform(-1, 1, nn)), name="thickness")
    E = csdl.Variable(value=1e8 * np.ones(nn), name="E")
    nu = csdl.Variable(value=0.3 * np.ones(nn), name="nu")
    density = csdl.Variable(value=10.0 * np.ones(nn), name="density")
    model = RMShellModel(mesh, shell_bc_func=marker, PENALTY_BC=False, record=False)        # strong Dirichlet rows: they must come back zero
    model.evaluate(pressure, thickness, E, nu, density, None, debug_mode=False, is_pressure=True)
    recorder.stop()
    fea = model.fea
    op = StateOperation(fea=fea, args_name_list=fea.states_dict["disp_solid"]["arguments"], state_name="disp_solid")
    inputs = {name: fea.inputs_dict[name]["function"].x.array.copy() for name in fea.states_dict["disp_solid"]["arguments"]}
    outputs = {}
    fea.opt_iter = 0
    op.solve_residual_equations(inputs, outputs)
    S = rng.uniform(-1, 1, (4, mesh.ndof))
    bc = np.concatenate([b.dof_indices()[0] for b in fea.bc])
    assert bc.size > 0
    for mode, (src, dst) in (("rev", ("d_outputs", "d_residuals")), ("fwd", ("d_residuals", "d_outputs"))):
        single = []
        for s in S:
            d = {"d_outputs": {}, "d_residuals": {}}
            d[src]["disp_solid"] = s.copy()
            op.apply_inverse_jacobian(inputs, outputs, d["d_outputs"], d["d_residuals"], mode)
            single.append(d[dst]["disp_solid"].copy())
        d = {"d_outputs": {}, "d_residuals": {}}
        d[src]["disp_solid"] = S.copy()
        op.apply_inverse_jacobian(inputs, outputs, d["d_outputs"], d["d_residuals"], mode)
        got = d[dst]["disp_solid"]
        assert got.shape == S.shape
        for i in range(4):
            assert rel(got[i], single[i]) < 1e-11, (mode, i)
        if mode == "rev":
            assert np.all(got[:, bc] == 0.0)
