"""The reference's operator protocol, end to end on the GPU, checked against the CPU oracle.

Reads like the reference's example driver (examples/advanced_examples/simple_shell_opt/
ex_simple_shell_opt.py:39-111): build RMShellModel on a 2x10 plate, evaluate, print-style sanity
against Euler-Bernoulli, then check_totals of compliance with respect to thickness."""
import numpy as np
import pytest

from femo_alpha_amd import csdl
from femo_alpha_amd.mesh import plate_mesh

pytestmark = pytest.mark.gpu

DOLFIN_EPS = 3e-16


def ClampedBoundary(x):
    return np.less(x[0], 0.0 + DOLFIN_EPS)


@pytest.mark.parametrize("element_wise_material,penalty", [(False, True), (True, True), (False, False)])
def test_rm_shell_model_protocol(element_wise_material, penalty):
    from femo_alpha_amd.rm_shell.rm_shell_model import RMShellModel
    from oracle.rm_shell_oracle import ShellOracle, degree4_rule
    mesh = plate_mesh(2.0, 10.0, 4, 20)
    nn, nel = mesh.nn, mesh.nel
    E_val, nu_val, h_val, rho_val, f_d = 1e8, 0.3, 0.1, 10.0, 5.0
    n_h = nel if element_wise_material else nn
    rng = np.random.default_rng(0)
    h0 = h_val * (1 + 0.2 * rng.uniform(-1, 1, n_h))

    recorder = csdl.Recorder(inline=True)
    recorder.start()
    pressure = csdl.Variable(value=np.zeros((nn, 3)), name="force_vector")
    pressure.value[:, 2] = f_d
    thickness = csdl.Variable(value=h0, name="thickness")
    E = csdl.Variable(value=E_val * np.ones(n_h), name="E")
    nu = csdl.Variable(value=nu_val * np.ones(n_h), name="nu")
    density = csdl.Variable(value=rho_val * np.ones(n_h), name="density")
    node_disp = csdl.Variable(value=np.zeros((nn, 3)), name="node_disp")
    model = RMShellModel(mesh, shell_bc_func=ClampedBoundary, element_wise_material=element_wise_material,
                         PENALTY_BC=penalty, record=False)
    out = model.evaluate(pressure, thickness, E, nu, density, node_disp, debug_mode=False, is_pressure=True)
    recorder.stop()

    # oracle on the same inputs
    o = ShellOracle(mesh, element_wise_material=element_wise_material,
                    penalty_facets=mesh.penalty_facets(ClampedBoundary) if penalty else None,
                    strong_dofs=None if penalty else mesh.locate_dofs_geometrical(ClampedBoundary))
    o.set_fields(h=h0, E=E_val, nu=nu_val, rho=rho_val, f=pressure.value)
    w_ref, J_ref, dJ_ref = o.forward_adjoint()

    # the drop-in surface solves like the reference's LU (utils_dolfinx.py:466,514-531): multifrontal factor + a few PCG steps
    its, relres = model.fea.last_solve
    assert its <= 4 and relres <= 1e-12
    assert out.disp_solid.shape == (mesh.ndof,)
    assert np.abs(out.disp_solid.value - w_ref).max() < 1e-7 * np.abs(w_ref).max()
    assert abs(out.compliance.value[0] - J_ref) < 1e-8 * abs(J_ref)
    assert abs(out.mass.value[0] - o.mass()) < 1e-11 * o.mass()
    assert abs(out.elastic_energy.value[0] - o.elastic_energy(w_ref)) < 1e-7 * o.elastic_energy(w_ref)
    assert out.disp_extracted.shape == (nn, 3)
    assert np.allclose(out.disp_extracted.value, w_ref[:3 * nn].reshape(nn, 3), rtol=0, atol=1e-7 * np.abs(w_ref).max())
    # Euler-Bernoulli sanity the reference prints (ex_simple_shell_opt.py:100-105)
    eb = f_d * 2.0 * 10.0 ** 4 / (8 * E_val * 2.0 * h_val ** 3 / 12)
    assert 0.8 * eb < np.abs(out.disp_solid.value[:mesh.ndof_u]).max() < 1.2 * eb

    # the post-processing idiom of the reference's example (ex_simple_shell_opt.py:142-147)
    w_fun = model.fea.states_dict["disp_solid"]["function"]
    u_mid = w_fun.sub(0).collapse().x.array
    theta = w_fun.sub(1).collapse().x.array
    assert u_mid.shape == (mesh.ndof_u,) and theta.shape == (3 * nn,)
    assert np.array_equal(np.concatenate([u_mid, theta]), w_fun.x.array)
    assert np.abs(u_mid - w_ref[: mesh.ndof_u]).max() < 1e-7 * np.abs(w_ref).max()
    assert out.F_solid.shape == (3 * nn,)

    # stress outputs (rm_shell_model.py:200-208, 230-239, 452-455)
    from oracle.rm_shell_oracle import ShellOracle as _SO, degree4_rule
    o3 = _SO(mesh, element_wise_material=element_wise_material, nquad=degree4_rule(mesh))
    o3.set_fields(h=h0, E=E_val, nu=nu_val)
    pn = o3.pnorm_stress(w_ref, 1e-6, 100)
    assert abs(out.pnorm_stress.value[0] - pn) < 1e-6 * pn
    assert abs(out.aggregated_stress.value[0] - 1e6 * pn ** 0.01) < 1e-7 * 1e6 * pn ** 0.01
    assert out.stress.shape == (4 * nel,)
    assert np.abs(out.stress.value.reshape(nel, 4) - o.stress_dg1(w_ref)).max() < 1e-6 * np.abs(o.stress_dg1(w_ref)).max()
    dA = recorder.compute_totals(out.aggregated_stress, thickness)
    i0 = int(np.argmax(np.abs(dA)))
    rows = recorder.check_totals(out.aggregated_stress, thickness, step=1e-4, indices=[i0])
    assert abs(rows[0][1] - rows[0][2]) < 1e-3 * abs(rows[0][2]), rows

    # shape derivative through the protocol: d compliance / d node_disp against a finite difference of the oracle
    dJu = recorder.compute_totals(out.compliance, node_disp)
    assert dJu.shape == (nn, 3)
    v = nn // 2
    def J_of(uh):
        o.set_fields(uhat=uh); return o.compliance(o.solve())
    up = np.zeros((nn, 3)); up[v, 2] = 1e-5
    fd = (J_of(up) - J_of(-up)) / 2e-5
    o.set_fields(uhat=np.zeros((nn, 3)))
    assert abs(dJu[v, 2] - fd) < 1e-4 * max(np.abs(dJu).max(), abs(fd))

    # total derivative through the operator protocol == oracle adjoint
    dJ = recorder.compute_totals(out.compliance, thickness)
    assert np.abs(dJ - dJ_ref).max() < 1e-7 * np.abs(dJ_ref).max()
    dM = recorder.compute_totals(out.mass, thickness)
    assert np.abs(dM - o.dmass_dh()).max() < 1e-11 * np.abs(o.dmass_dh()).max()
    # the reference's own verification: check_totals (finite differences through the GPU solve)
    rows = recorder.check_totals(out.compliance, thickness, step=1e-3, indices=[0, n_h // 3, n_h - 1])
    for i, ana, fd, err in rows:
        assert abs(ana - fd) < 2e-5 * np.abs(dJ).max(), (i, ana, fd)


def test_operator_error_behaviour():
    from femo_alpha_amd.csdl_alpha_opt.state_operation import StateOperation
    from femo_alpha_amd.rm_shell.rm_shell_model import RMShellModel
    mesh = plate_mesh(2.0, 10.0, 2, 4)
    with pytest.raises(ValueError, match="shell bc location"):
        RMShellModel(mesh, shell_bc_func=None)
    model = RMShellModel(mesh, shell_bc_func=ClampedBoundary, record=False)
    fea = model.fea
    with pytest.raises(ValueError, match="already been used"):
        fea.add_input("thickness", fea.inputs_dict["E"]["function"])
    op = StateOperation(fea=fea, args_name_list=fea.states_dict["disp_solid"]["arguments"], state_name="disp_solid")
    with pytest.raises(ValueError, match="not found in the FEA model"):
        op.evaluate(csdl.VariableGroup())
    with pytest.raises(ValueError, match="mode must be"):
        op.compute_jacvec_product({}, {}, {}, {}, {}, "sideways")
    with pytest.raises(ValueError, match="mode must be"):
        op.apply_inverse_jacobian({}, {}, {}, {}, "sideways")
    with pytest.raises(TypeError):
        StateOperation(fea="not an FEA", args_name_list=[], state_name="disp_solid")


def test_forces_instead_of_pressures():
    """is_pressure=False: nodal forces are turned into pressures with the consistent CG1 mass matrix
    (reference rm_shell_model.py:414-421, rm_shell_pde.py:194-209); a uniform pressure must be recovered from
    its own consistent nodal forces, and totals flow back through the map."""
    from femo_alpha_amd.rm_shell.rm_shell_model import RMShellModel
    mesh = plate_mesh(2.0, 10.0, 3, 9)
    nn = mesh.nn
    rec = csdl.Recorder(inline=True); rec.start()
    model = RMShellModel(mesh, shell_bc_func=ClampedBoundary, record=False)
    A = model.shell_pde.construct_force_to_pressure_map()
    p_uniform = np.zeros((nn, 3)); p_uniform[:, 2] = 5.0
    forces = csdl.Variable(value=(A @ p_uniform.ravel()).reshape(nn, 3), name="force_vector")
    assert abs(forces.value[:, 2].sum() - 5.0 * 20.0) < 1e-10          # total force = pressure x area
    mk = lambda v, n: csdl.Variable(value=v * np.ones(nn), name=n)
    out_f = model.evaluate(forces, mk(0.1, "thickness"), mk(1e8, "E"), mk(0.3, "nu"), mk(10.0, "density"), is_pressure=False)
    rec.stop()
    assert np.abs(out_f.F_solid.value.reshape(nn, 3) - p_uniform).max() < 1e-10
    g = rec.compute_totals(out_f.compliance, forces)
    assert g.shape == (nn, 3) and np.abs(g[:, 2]).max() > 0


@pytest.mark.parametrize("kind", ["warped_quads", "triangles"])
def test_force_to_pressure_solve_on_the_device(kind):
    """femo_force_to_pressure (Jacobi-PCG, the [CG1]^3 mass matrix applied cell by cell on the device) against a host sparse solve
    with the matrix of construct_force_to_pressure_map (rm_shell_pde.py:194-209) on warped quadrilaterals and on triangles; the
    operator is symmetric, so the same call is its own transpose (the reverse mode of rm_shell_model.py:420)."""
    import scipy.sparse.linalg as spla                     # test-side checker only
    from femo_alpha_amd.backend import ShellContext
    from femo_alpha_amd.mesh import quads_to_triangles, wing_skin_mesh
    from femo_alpha_amd.rm_shell.rm_shell_pde import force_to_pressure_map
    mesh = wing_skin_mesh(8, 24, shuffle=True)
    if kind == "triangles":
        mesh = quads_to_triangles(mesh)
    A = force_to_pressure_map(mesh)
    rng = np.random.default_rng(3)
    f = rng.uniform(-1, 1, 3 * mesh.nn)
    c = ShellContext(mesh)
    p = c.force_to_pressure(f)
    it, rr = c.last_force_to_pressure
    ref = spla.spsolve(A.tocsc(), f)
    assert it < 200 and rr <= 1e-13
    assert np.abs(p - ref).max() < 1e-11 * np.abs(ref).max()
    g = rng.uniform(-1, 1, 3 * mesh.nn)
    assert abs(g @ p - c.force_to_pressure(g) @ f) < 1e-11 * abs(g @ p)        # symmetry: g . A^-1 f = (A^-1 g) . f
    with pytest.raises(ValueError):
        c.force_to_pressure(f[:-1])
    c.close()


def test_mesh_tags_give_per_tag_stress_aggregates():
    """RMShellModel(mesh_tags=...) registers pnorm_stress_<tag> per sub-domain (rm_shell_model.py:101-133, 242-253)."""
    from femo_alpha_amd.rm_shell.rm_shell_model import RMShellModel
    from oracle.rm_shell_oracle import ShellOracle, degree4_rule
    mesh = plate_mesh(2.0, 10.0, 4, 20)
    nn, nel = mesh.nn, mesh.nel
    cx = mesh.nodes[mesh.cells].mean(axis=1)[:, 0]
    mesh_tags = {"root_bay": np.nonzero(cx < 3.0)[0].tolist(), 7: np.nonzero(cx > 6.0)[0].tolist()}
    recorder = csdl.Recorder(inline=True)
    recorder.start()
    pressure = csdl.Variable(value=np.zeros((nn, 3)), name="force_vector")
    pressure.value[:, 2] = 5.0
    thickness = csdl.Variable(value=0.1 * np.ones(nn), name="thickness")
    E = csdl.Variable(value=1e8 * np.ones(nn), name="E")
    nu = csdl.Variable(value=0.3 * np.ones(nn), name="nu")
    density = csdl.Variable(value=10.0 * np.ones(nn), name="density")
    model = RMShellModel(mesh, shell_bc_func=ClampedBoundary, record=False, mesh_tags=mesh_tags)
    assert model.association_table == {"root_bay": 0, 7: 1}
    out = model.evaluate(pressure, thickness, E, nu, density)
    recorder.stop()
    o = ShellOracle(mesh, penalty_facets=mesh.penalty_facets(ClampedBoundary))
    o.set_fields(h=0.1, E=1e8, nu=0.3, rho=10.0, f=pressure.value)
    w_ref = o.solve()
    o3 = ShellOracle(mesh, nquad=degree4_rule(mesh))
    o3.set_fields(h=0.1, E=1e8, nu=0.3)
    for tag, cells in mesh_tags.items():
        ref = o3.pnorm_stress(w_ref, 1e-6, 100, cells=cells)
        got = getattr(out, "pnorm_stress_" + str(tag)).value[0]
        assert abs(got - ref) < 1e-6 * ref, (tag, got, ref)
    whole = o3.pnorm_stress(w_ref, 1e-6, 100)
    assert abs(out.pnorm_stress.value[0] - whole) < 1e-6 * whole
    # the root bay carries the largest stresses of a cantilever
    assert out.pnorm_stress_root_bay.value[0] > getattr(out, "pnorm_stress_7").value[0]
    with pytest.raises(ValueError, match="one tag only"):
        RMShellModel(mesh, shell_bc_func=ClampedBoundary, record=False, mesh_tags={"a": [0, 1], "b": [1]})


def test_renumbered_model_answers_in_caller_order():
    """renumber=True: the solver works on a Morton-ordered copy of the mesh (as dolfinx reorders what it reads);
    inputs, nodal displacements and gradients keep the caller's numbering (rm_shell_model.py:396-438, 505-527)."""
    from femo_alpha_amd.mesh import wing_skin_mesh
    from femo_alpha_amd.rm_shell.rm_shell_model import RMShellModel
    from oracle.rm_shell_oracle import ShellOracle, degree4_rule
    mesh = wing_skin_mesh(8, 24, shuffle=True)
    nn, nel = mesh.nn, mesh.nel
    root = lambda x: np.less(x[1], 1e-12)
    rng = np.random.default_rng(3)
    h0 = 0.02 * (1 + 0.3 * rng.uniform(-1, 1, nn))
    f0 = rng.uniform(-1, 1, (nn, 3)) * 40.0
    recorder = csdl.Recorder(inline=True)
    recorder.start()
    pressure = csdl.Variable(value=f0, name="force_vector")
    thickness = csdl.Variable(value=h0, name="thickness")
    E = csdl.Variable(value=7e9 * np.ones(nn), name="E")
    nu = csdl.Variable(value=0.3 * np.ones(nn), name="nu")
    density = csdl.Variable(value=2700.0 * np.ones(nn), name="density")
    tags = {"outboard": np.nonzero(mesh.nodes[mesh.cells].mean(axis=1)[:, 1] > 3.0)[0].tolist()}
    model = RMShellModel(mesh, shell_bc_func=root, record=False, renumber=True, mesh_tags=tags)
    assert not np.array_equal(model.vertex_of_new, np.arange(nn))
    ctx = model.shell_pde.ctx
    ctx.set_solver(preconditioner=2, rtol=1e-12, maxit=50, check_every=1)
    out = model.evaluate(pressure, thickness, E, nu, density)
    recorder.stop()
    o = ShellOracle(mesh, penalty_facets=mesh.penalty_facets(root))
    o.set_fields(h=h0, E=7e9, nu=0.3, rho=2700.0, f=f0)
    w_ref, J_ref, dJ_ref = o.forward_adjoint()
    assert abs(out.compliance.value[0] - J_ref) < 1e-8 * abs(J_ref)
    assert abs(out.mass.value[0] - o.mass()) < 1e-11 * o.mass()
    u_ref = w_ref[:3 * nn].reshape(nn, 3)
    assert np.abs(out.disp_extracted.value - u_ref).max() < 1e-7 * np.abs(u_ref).max()
    dJ = recorder.compute_totals(out.compliance, thickness)
    assert np.abs(np.ravel(dJ) - dJ_ref).max() < 1e-7 * np.abs(dJ_ref).max()
    o3 = ShellOracle(mesh, nquad=degree4_rule(mesh))
    o3.set_fields(h=h0, E=7e9, nu=0.3)
    ref = o3.pnorm_stress(w_ref, 1e-6, 100, cells=tags["outboard"])
    assert abs(out.pnorm_stress_outboard.value[0] - ref) < 1e-6 * ref


def test_thickness_optimisation_loop():
    """The reference's example end to end (ex_simple_shell_opt.py:114-131): thickness design variable with bounds,
    mass held at its initial value, an SLSQP loop in which every function evaluation is a forward solve and every
    gradient an adjoint solve on the GPU.  A few iterations must lower the compliance markedly at constant mass."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    from optimize import slsqp
    from femo_alpha_amd.rm_shell.rm_shell_model import RMShellModel
    mesh = plate_mesh(2.0, 10.0, 4, 20)
    nn = mesh.nn
    recorder = csdl.Recorder(inline=True)
    recorder.start()
    pressure = csdl.Variable(value=np.zeros((nn, 3)), name="force_vector")
    pressure.value[:, 2] = 5.0
    thickness = csdl.Variable(value=0.1 * np.ones(nn), name="thickness")
    E = csdl.Variable(value=1e8 * np.ones(nn), name="E")
    nu = csdl.Variable(value=0.3 * np.ones(nn), name="nu")
    density = csdl.Variable(value=10.0 * np.ones(nn), name="density")
    model = RMShellModel(mesh, shell_bc_func=ClampedBoundary, record=False)
    ctx = model.shell_pde.ctx
    ctx.set_solver(preconditioner=2, rtol=1e-11, maxit=50, check_every=1)
    out = model.evaluate(pressure, thickness, E, nu, density)
    J0, m0 = float(out.compliance.value[0]), float(out.mass.value[0])
    thickness.set_as_design_variable(upper=0.2, lower=2e-2)
    out.mass.set_as_constraint(lower=m0, upper=m0)
    out.compliance.set_as_objective(scaler=1.0 / J0)
    res = slsqp(recorder, maxiter=8, ftol=1e-12)
    recorder.stop()
    J1, m1 = float(out.compliance.value[0]), float(out.mass.value[0])
    assert J1 < 0.7 * J0, (J0, J1, res.message)
    assert abs(m1 - m0) < 1e-6 * m0
    assert thickness.value.min() >= 2e-2 - 1e-12 and thickness.value.max() <= 0.2 + 1e-12
    # material moved towards the clamped root, as it must for a cantilever
    x = mesh.nodes[:, 0]
    assert thickness.value[x < 2.0].mean() > thickness.value[x > 8.0].mean()


def test_stale_factor_policy_gives_the_same_answers():
    """Option "stale_factor" (for optimisation loops): after a change of FIELDS only, the previous design's factor stays the PCG
    preconditioner; the operator is re-factorised at once if a field has moved more than "stale_rel" from the factor's design, and
    inside the solve if it has not converged after "stale_factor" iterations.  PCG iterates on the current matrix-free operator, so
    displacement, compliance and gradient must equal the always-refactorise run's (1e-8) whichever path a solve takes:
    femo_last_timing [3] = 0 fresh factor, 1 kept factor, 2 kept, then refreshed inside the solve.  A change of the Dirichlet data must
    discard the kept factor altogether."""
    from femo_alpha_amd.backend import ShellContext
    m = plate_mesh(2.0, 10.0, 24, 120)
    rng = np.random.default_rng(3)
    h0 = 0.1 * (1 + 0.2 * rng.uniform(-1, 1, m.nn))
    steps = (1e-3, 1e-2, 5e-2, 0.5)
    designs = [h0 * (1 + d * rng.uniform(-1, 1, m.nn)) for d in steps]

    def run(stale, rel=None):
        c = ShellContext(m)
        for k, v in dict(thickness=h0, E=[1e8], nu=[0.3], density=[10.0], F_solid=np.tile([0.0, 0.0, 5.0], (m.nn, 1))).items():
            c.set_field(k, v)
        c.set_penalty_facets(m.penalty_facets(ClampedBoundary))
        c.enable_frontal()
        c.set_solver(preconditioner=2, rtol=1e-12, maxit=60, check_every=1)
        c.set_option("stale_factor", stale)
        if rel is not None:
            c.set_option("stale_rel", rel)
        c.solve_state(zero_guess=True)
        assert c.last_timing()["factor_state"] == 0              # nothing to keep yet
        rows = []
        for h in designs:
            c.set_field("thickness", h)
            it, rr = c.solve_state(zero_guess=True)
            st = c.last_timing()["factor_state"]
            w = c.get_state()
            J = c.functional("compliance")
            g, it2, _ = c.total_gradient("compliance", "thickness")
            rows.append((it, st, it2, c.last_timing()["factor_state"], w, J, g))
        # Dirichlet data changed: the kept factor belongs to another problem and must not be used
        c.set_penalty_facets(m.penalty_facets(lambda x: np.less(x[0], 0.5)))
        c.solve_state(zero_guess=True)
        assert c.last_timing()["factor_state"] == 0
        c.close()
        return rows

    ref = run(0)
    assert all(r[1] == 0 for r in ref)
    for got in (run(6, rel=10.0), run(6)):                     # the gate off (every design tries the kept factor), then at its default 2e-3
        for (_, _, _, _, w0, J0, g0), (it, st, it2, st2, w, J, g), d in zip(ref, got, steps):
            assert np.abs(w - w0).max() < 1e-8 * np.abs(w0).max(), (d, st)
            assert abs(J - J0) < 1e-8 * abs(J0), (d, st)
            assert np.abs(g - g0).max() < 1e-8 * np.abs(g0).max(), (d, st)
        assert got[0][1] == 1 and got[0][0] <= 6 and got[0][3] == 1      # 0.1 % change: absorbed by the kept factor, forward and adjoint
    gate_off, gated = run(6, rel=10.0), run(6)
    assert gate_off[-1][1] == 2 and gate_off[-1][3] == 0 and gate_off[-1][2] <= 3     # 50 % change: refreshed INSIDE the solve, the adjoint finds the fresh factor
    assert gated[-1][1] == 0 and gated[-1][0] <= 3                                    # with the gate: refreshed at once, no wasted iterations


def test_stale_factor_gate_compares_with_the_design_of_the_stored_factor():
    """every path that completes a factorisation records the design it belongs to -- not only the solver's own.  A public
    ``factorize()`` at a DISTANT design followed by a small change of the fields must find the kept factor near (factor_state 1, a few
    iterations), and a ``factorize()`` followed by a return to the first design must find it far (refreshed at once, factor_state 0);
    switching the mesh motion on or off is a change whatever its size; and a kept factor never ends a solve unconverged at a small maxit."""
    from femo_alpha_amd.backend import ShellContext
    m = plate_mesh(2.0, 10.0, 16, 80)
    rng = np.random.default_rng(4)
    h0 = 0.1 * (1 + 0.2 * rng.uniform(-1, 1, m.nn))
    h_far = h0 * (1 + 0.5 * rng.uniform(-1, 1, m.nn))
    c = ShellContext(m)
    for k, v in dict(thickness=h0, E=[1e8], nu=[0.3], density=[10.0], F_solid=np.tile([0.0, 0.0, 5.0], (m.nn, 1))).items():
        c.set_field(k, v)
    c.set_penalty_facets(m.penalty_facets(ClampedBoundary))
    c.enable_frontal()
    c.set_solver(preconditioner=2, rtol=1e-12, maxit=60, check_every=1)
    c.set_option("stale_factor", 6)
    c.solve_state(zero_guess=True)                                   # snapshot: h0
    c.set_field("thickness", h_far)
    c.factorize()                                                    # the stored factor now belongs to h_far ...
    c.set_field("thickness", h_far * (1 + 1e-4 * rng.uniform(-1, 1, m.nn)))
    it, _ = c.solve_state(zero_guess=True)
    assert c.last_timing()["factor_state"] == 1 and it <= 6          # ... and a design next to h_far is near it
    w_near = c.get_state()
    c.set_field("thickness", h0)
    it, _ = c.solve_state(zero_guess=True)
    assert c.last_timing()["factor_state"] == 0 and it <= 3          # h0 is far from the factor's design: refreshed at once
    w0 = c.get_state()
    # mesh motion switched on: a different operator however small uhat is
    c.set_field("uhat", 1e-9 * rng.uniform(-1, 1, (m.nn, 3)))
    c.solve_state(zero_guess=True)
    assert c.last_timing()["factor_state"] == 0
    c.set_field("uhat", np.zeros((m.nn, 3)))
    c.solve_state(zero_guess=True)
    assert c.last_timing()["factor_state"] == 0
    assert np.abs(c.get_state() - w0).max() < 1e-8 * np.abs(w0).max()
    # maxit below stale_factor: the refresh comes in time for the solve to converge
    c.set_solver(preconditioner=2, rtol=1e-12, maxit=4, check_every=1)
    c.set_option("stale_rel", 10.0)
    c.set_field("thickness", h_far)
    it, rr = c.solve_state(zero_guess=True)
    assert rr <= 1e-12 and c.last_timing()["factor_state"] in (1, 2)
    # the answers are those of a context that always factorises
    c2 = ShellContext(m)
    for k, v in dict(thickness=h_far, E=[1e8], nu=[0.3], density=[10.0], F_solid=np.tile([0.0, 0.0, 5.0], (m.nn, 1))).items():
        c2.set_field(k, v)
    c2.set_penalty_facets(m.penalty_facets(ClampedBoundary))
    c2.use_direct_solver()
    c2.solve_state(zero_guess=True)
    assert np.abs(c.get_state() - c2.get_state()).max() < 1e-8 * np.abs(c2.get_state()).max()
    assert np.abs(w_near - c2.get_state()).max() < 1e-3 * np.abs(c2.get_state()).max()      # (a design 1e-4 away)
    c.close(); c2.close()
