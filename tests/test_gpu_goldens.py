"""HIP path against the committed full-size goldens (tests/golden/make_fullsize_goldens.py): BASELINE config 1
(8 046 DOF) and config 2 (255 438 DOF, nodal and element-wise thickness).  This is the north-star tolerance --
displacement, compliance and d compliance / d thickness to 1e-8 relative -- asserted at 250 k DOF through the
drop-in solver (multifrontal Cholesky + PCG refinement), not at the 3 k DOF of the seeded parity cases.

The goldens are the solution of the discrete problem itself: the CPU oracle's operator assembled in an extended arithmetic
(tests/golden/_extended.py: double-double since round 6) and an iterative refinement against it (the size of the last correction is
stored with them: <= 1e-18), so the numbers below are statements about the HIP path.  The float64-assembled oracle matrix -- what
rounds 1-3 refined against -- is itself 1e-8 (config 2) to 2e-7 (config 3) away: ``float64_matrix_distance_*`` in the files.

Round 6: the goldens of rounds 4-5 were refined in x87 extended precision, and their refinement stalled at corrections of 1e-11 .. 2e-10
(condition number x 2^-64).  The double-double goldens differ from them by 4e-11 / 4e-11 / 6e-11 in displacement / compliance / gradient
at config 3 -- exactly what rounds 4-5 reported as the HIP path's distance.  Against the double-double goldens the HIP path sits at
**2e-14 .. 3e-14 / 3e-16 .. 3e-15 / 5e-14 .. 9e-14** at 1 M DOF: the 1e-8 bar is asserted (TOL), and so is 1e-11 (SHARP), so that a
regression of the measured agreement does not hide three orders of magnitude below the bar."""
import os

import numpy as np
import pytest

from femo_alpha_amd.mesh import plate_mesh

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CLAMP = lambda x: np.less(x[0], 3e-16)
TOL = 1e-8          # BASELINE.json: "adjoint dJ/dt matching to 1e-8 rel"; the same bar for displacement and compliance
SHARP = 1e-11       # what the HIP path measures against the double-double goldens with room to spare (2e-14 .. 9e-14 at 1 M DOF)


def _triple(tag, w, J, dJ, g, tol=SHARP):
    """The north-star triple against a golden: printed, asserted at ``tol``."""
    ew = np.abs(w[g["w_sample_index"]] - g["w_sample"]).max() / float(g["w_maxabs"])
    eJ = abs(J - float(g["compliance"])) / abs(float(g["compliance"]))
    ref = g["dcompliance_dthickness"]
    eg = np.abs(dJ - ref).max() / np.abs(ref).max()
    print(f"{tag}: displacement {ew:.1e}, compliance {eJ:.1e}, gradient {eg:.1e} from the golden")
    assert ew < tol and eJ < tol and eg < tol, (tag, ew, eJ, eg)


@pytest.mark.parametrize("name", ["config1_plate_10x50_nodal", "config2_plate_58x290_nodal",
                                  "config2_plate_58x290_elementwise"])
def test_parity_triple_against_fullsize_golden(name):
    from femo_alpha_amd.backend import ShellContext
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    assert max(float(g["w_correction"]), float(g["lam_correction"])) < 1e-2 * TOL      # the golden is sharper than the bar
    m = plate_mesh(2.0, 10.0, int(g["nx"]), int(g["ny"]))
    assert m.ndof == int(g["ndof"])
    c = ShellContext(m, element_wise_material=bool(g["element_wise"]))
    for k, v in dict(thickness=g["thickness"], E=[1e8], nu=[0.3], density=[10.0],
                     F_solid=np.tile([0.0, 0.0, 5.0], (m.nn, 1))).items():
        c.set_field(k, v)
    c.set_penalty_facets(m.penalty_facets(CLAMP))
    c.use_direct_solver(rtol=1e-13)
    it, rr = c.solve_state(zero_guess=True)
    assert it <= 5 and rr <= 1e-13
    w = c.get_state()
    assert abs(np.abs(w).max() - float(g["w_maxabs"])) < TOL * float(g["w_maxabs"])
    assert np.abs(w[g["w_sample_index"]] - g["w_sample"]).max() < TOL * float(g["w_maxabs"])
    J = c.functional("compliance")
    assert abs(J - float(g["compliance"])) < TOL * abs(float(g["compliance"]))
    assert abs(c.functional("mass") - float(g["mass"])) < 1e-12 * float(g["mass"])
    assert abs(c.functional("elastic_energy") - float(g["elastic_energy"])) < TOL * float(g["elastic_energy"])
    dJ, it2, rr2 = c.total_gradient("compliance", "thickness")
    ref = g["dcompliance_dthickness"]
    assert it2 <= 5
    assert np.abs(dJ - ref).max() < TOL * np.abs(ref).max()
    _triple(name, w, J, dJ, g)
    # and entry by entry wherever the gradient is not small (99 % of the entries): no cancellation hides behind the max norm
    big = np.abs(ref) > 1e-3 * np.abs(ref).max()
    assert np.abs(dJ[big] / ref[big] - 1.0).max() < 1e-6
    c.close()


@pytest.mark.parametrize("nquad", [None, 4])
def test_parity_triple_against_the_one_million_dof_golden(nquad):
    """BASELINE config 3 -- the bench workload itself, 1 015 470 DOF (tests/golden/make_config3_golden.py: CPU multifrontal
    Cholesky of the oracle's matrix, refined in double-double to 1e-18).

    nquad None: the rule the mesh asks for and the bench runs -- 5 x 5 Gauss points on these warped cells, within 1e-9 of the
    reference's (nearly) exact integration (linear_shell_model.py:88-103; golden config3_wing1m.npz, made with n = 5).
    nquad 4: the rule of rounds 1-3 against ITS golden (config3_wing1m_n4.npz); the two goldens differ by 7.5e-8 in the
    gradient, which is why n = 4 is no longer the default on warped meshes.

    Tolerance 1e-8, the north-star bar, at the north-star size.  Rounds 1-3 asserted 1e-7 here and blamed the conditioning of the
    1.27 mm skin; the floor was the GOLDEN's: it solved the float64-ASSEMBLED matrix, whose entry rounding alone moves this
    solution by 2.4e-7 (stored with the golden: float64_matrix_distance_*; a one-ulp change of a Gauss weight moves it by 3.5e-7).
    The goldens now solve the discrete problem itself (operator assembled in an extended arithmetic, tests/golden/_extended.py),
    and the HIP path -- which never forms a matrix and iterates on its own matrix-free residual -- sits 3e-14 / 3e-15 / 9e-14 from that
    (rounds 4-5 measured 5e-11: the x87 golden's own refinement floor, see the module docstring)."""
    from bench import make_workload
    from femo_alpha_amd.backend import ShellContext
    tol = TOL
    g = np.load(os.path.join(GOLDEN, "config3_wing1m.npz" if nquad is None else f"config3_wing1m_n{nquad}.npz"))
    assert max(float(g["w_correction"]), float(g["lam_correction"])) < 1e-2 * tol          # the golden is sharper than the bar
    m, fields, marker, _ = make_workload("wing1m")
    assert m.ndof == int(g["ndof"])
    c = ShellContext(m, nquad=nquad)
    assert c.nquad == (5 if nquad is None else nquad) and int(g["nquad"] if "nquad" in g.files else 4) == c.nquad
    for k, v in fields.items():
        c.set_field(k, v)
    c.set_penalty_facets(m.penalty_facets(marker))
    c.use_direct_solver()
    it, rr = c.solve_state(zero_guess=True)
    assert it <= 4 and rr <= 1e-12
    w = c.get_state()
    assert abs(np.abs(w).max() - float(g["w_maxabs"])) < tol * float(g["w_maxabs"])
    assert np.abs(w[g["w_sample_index"]] - g["w_sample"]).max() < tol * float(g["w_maxabs"])
    J = c.functional("compliance")
    assert abs(J - float(g["compliance"])) < tol * abs(float(g["compliance"]))
    assert abs(c.functional("mass") - float(g["mass"])) < 1e-12 * float(g["mass"])
    dJ, it2, _ = c.total_gradient("compliance", "thickness")
    ref = g["dcompliance_dthickness"]
    assert it2 <= 4
    assert np.abs(dJ - ref).max() < tol * np.abs(ref).max()
    _triple(f"wing1m n = {c.nquad}", w, J, dJ, g)
    c.close()


@pytest.mark.parametrize("workload", ["uskin1m", "wing1m_tri", "uquad1m", "uquad1m_n5"])
def test_parity_triple_on_the_unstructured_and_triangle_skins(workload):
    """The surface of config 3 on other meshes, against the exact discrete solution (tests/golden/make_config3_golden.py auto <workload>: the
    C++ oracle's operator assembled in double-double, refined to 1e-18): displacement, compliance and the full
    d compliance / d thickness vector at 1e-8.
      uskin1m     an UNSTRUCTURED triangulation (134 560 CG2xCG1 triangles by Delaunay, vertex valences 3..9, 1 015 470 DOF);
      wing1m_tri  the triangle variant SURVEY.md section 8d defines (183 x 365 quads split: 133 590 triangles, 1 006 863 DOF);
      uquad1m     an UNSTRUCTURED ALL-QUADRILATERAL skin (67 398 kites: every Delaunay triangle cut into three quadrilaterals, vertex
                  valences 3..9, strongly non-affine and warped cells, 1 016 124 DOF) -- what the reference's real wings are
                  (ex_lpc_gust_response_opt.py:142-153).  6 x 6 Gauss points, the rule ShellMesh.recommended_nquad gives for such cells
                  (5 x 5 is 2.1e-8 from the limit in the gradient there, profiles/r5_quadrature_uquad1m.txt); ``uquad1m_n5``: the same
                  mesh with 5 x 5 points against ITS exact discrete solution.
    The Delaunay triangulations come from scipy / qhull; the goldens carry theirs (``triangulation``), so the meshes here are the goldens'
    meshes whatever qhull is installed, and the checksum below can only fail if the generator itself changes."""
    import hashlib
    from bench import make_workload
    from femo_alpha_amd.backend import ShellContext
    g = np.load(os.path.join(GOLDEN, f"config3_{workload}.npz"))
    workload, nquad = (workload[:-3], int(workload[-1])) if workload.endswith(("_n5",)) else (workload, None)
    m, fields, marker, _ = make_workload(workload, tri=g["triangulation"] if "triangulation" in g.files else None)
    sha = hashlib.sha256(np.ascontiguousarray(m.cells, dtype=np.int64).tobytes() + np.ascontiguousarray(m.nodes).tobytes()).hexdigest()
    assert sha == str(g["mesh_sha256"]), "the mesh generator no longer produces the mesh of the golden"
    assert m.ndof == int(g["ndof"]) == {"uskin1m": 1015470, "wing1m_tri": 1006863, "uquad1m": 1016124}[workload]
    assert m.is_quad == (workload == "uquad1m")
    assert max(float(g["w_correction"]), float(g["lam_correction"])) < 1e-3 * SHARP     # the golden is sharper than either bar (double-double: 1e-18)
    c = ShellContext(m, nquad=nquad)
    assert c.nquad == int(g["nquad"]) == {"uquad1m": 6 if nquad is None else nquad}.get(workload, int(g["nquad"]))
    for k, v in fields.items():
        c.set_field(k, v)
    c.set_penalty_facets(m.penalty_facets(marker))
    c.use_direct_solver()
    it, rr = c.solve_state(zero_guess=True)
    assert it <= 4 and rr <= 1e-12
    w = c.get_state()
    assert abs(np.abs(w).max() - float(g["w_maxabs"])) < TOL * float(g["w_maxabs"])
    assert np.abs(w[g["w_sample_index"]] - g["w_sample"]).max() < TOL * float(g["w_maxabs"])
    J = c.functional("compliance")
    assert abs(J - float(g["compliance"])) < TOL * abs(float(g["compliance"]))
    assert abs(c.functional("mass") - float(g["mass"])) < 1e-12 * float(g["mass"])
    dJ, it2, _ = c.total_gradient("compliance", "thickness")
    ref = g["dcompliance_dthickness"]
    assert it2 <= 4 and np.abs(dJ - ref).max() < TOL * np.abs(ref).max()
    _triple(f"{workload} n = {c.nquad}", w, J, dJ, g)
    c.close()


@pytest.mark.parametrize("rtol", [1e-13, None])
def test_config5_march_against_the_full_size_golden(rtol):
    """BASELINE config 5 at full size (82 x 410 plate, 508 734 DOF, 100 midpoint / Newmark steps under the 1-cosine gust) against
    tests/golden/config5_plate500k_dynamic.npz -- the CPU restatement's march (C++/OpenMP element kernels, multifrontal Cholesky) with
    every step's solve refined to 1e-14 (make_config5_golden.py): tip deflection at every time level, samples of the last state and
    velocity, total strain energy, at 1e-8.  The march runs inside the library (femo_newmark_*), factorised once.
      rtol 1e-13  two applications of the factor per time step (a refinement step on the matrix-free residual): 6e-15 from the golden;
      rtol None   the PRODUCT DEFAULT (PlateSim rtol 1e-8): PCG stops after the FIRST application of the exact factor, one direct solve
                  per time step as the reference's single Newton iteration with LU (nonlinear_utils.py:220-229).  Measured 7e-10 (tip
                  history) / 1e-9 (last state) from the golden: inside the 1e-8 bar as well -- the round-4 docstring's "~1e-7" was a guess."""
    from bench import dynamic_case
    from femo_alpha_amd.dynamic_rm_shell.plate_sim import PlateSim
    g = np.load(os.path.join(GOLDEN, "config5_plate500k_dynamic.npz"))
    N = int(g["nsteps"])
    mesh, dt, F = dynamic_case(nsteps=N)
    assert mesh.ndof == int(g["ndof"]) == 508734 and abs(dt - float(g["dt"])) < 1e-15
    ps = PlateSim(mesh, 1e8, 0.3, 10.0, dt, N, quad_deg=3, **({} if rtol is None else {"rtol": rtol}))
    assert ps.rtol == (1e-8 if rtol is None else rtol)
    ps.update_f_history(F)
    ps.update_t(np.full(mesh.nn, 0.1))
    W = ps.solve_dynamic_problem()                                   # (ndof, N + 1)
    assert max(i for i, _ in ps.solve_info) == (1 if rtol is None else 2)      # applications of the factor per time step
    tip = int(g["tip_vertex"])
    hist = W[3 * tip + 2, :]
    ref = g["tip_history"]
    assert np.abs(hist - ref).max() < TOL * np.abs(ref).max()
    wl = W[:, -1]
    assert np.abs(wl[g["sample_index"]] - g["w_last_sample"]).max() < TOL * float(g["w_last_maxabs"])
    assert abs(np.abs(wl).max() - float(g["w_last_maxabs"])) < TOL * float(g["w_last_maxabs"])
    # the velocity of the last level (the recursion of the midpoint rule on the history) and the summed strain energy through the
    # library's own operator
    b = 2.0 / dt
    wd = np.zeros(mesh.ndof)
    for i in range(1, N + 1):
        wd = b * (W[:, i] - W[:, i - 1]) - wd
    assert np.abs(wd[g["sample_index"]] - g["wdot_last_sample"]).max() < 100 * TOL * float(g["wdot_last_maxabs"])      # 100 steps of differences
    U, T, work = ps.energy_audit()
    assert abs(U.sum() - float(g["total_strain_energy"])) < TOL * float(g["total_strain_energy"])


def test_distance_of_the_transient_march_from_the_golden_at_both_solver_settings():
    """What bench.py --workload plate500k_dynamic prints as ``distance_from_golden``: the product default (one application of the factor
    per time step) and the parity setting (two) against the full-size golden, through bench.golden_distance_dynamic.  Measured on the
    MI355X: 6.8e-10 / 1.0e-9 (tip history / last state) and 6e-15 / 6e-14.  The bounds asserted leave a factor ~10."""
    from bench import dynamic_case, golden_distance_dynamic
    from femo_alpha_amd.dynamic_rm_shell.plate_sim import PlateSim
    g = np.load(os.path.join(GOLDEN, "config5_plate500k_dynamic.npz"))
    N = int(g["nsteps"])
    mesh, dt, F = dynamic_case(nsteps=N)
    for rtol, its, bound in ((1e-8, 1, 1e-8), (1e-13, 2, 1e-12)):
        ps = PlateSim(mesh, 1e8, 0.3, 10.0, dt, N, quad_deg=3, rtol=rtol)
        ps.update_f_history(F)
        ps.update_t(np.full(mesh.nn, 0.1))
        W = ps.solve_dynamic_problem()                                   # (ndof, N + 1)
        assert max(i for i, _ in ps.solve_info) == its
        d = golden_distance_dynamic(np.ascontiguousarray(W.T), mesh)
        print(f"rtol {rtol:g} ({its} application(s) per time step): tip history {d['tip_history']:.2e}, last state {d['last_state_samples']:.2e} from the golden")
        assert d["tip_history"] < bound and d["last_state_samples"] < bound
        ps.ctx.close()
