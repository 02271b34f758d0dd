"""HIP path against the committed full-size goldens (tests/golden/make_fullsize_goldens.py): BASELINE config 1
(8 046 DOF) and config 2 (255 438 DOF, nodal and element-wise thickness).  This is the north-star tolerance --
displacement, compliance and d compliance / d thickness to 1e-8 relative -- asserted at 250 k DOF through the
drop-in solver (multifrontal Cholesky + PCG refinement), not at the 3 k DOF of the seeded parity cases.

The goldens come from the CPU oracle polished by extended-precision iterative refinement (their own distance from
the exact discrete solution is stored with them, ~1e-13), so the 1e-8 below is a statement about the HIP path."""
import os

import numpy as np
import pytest

from femo_alpha_amd.mesh import plate_mesh

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CLAMP = lambda x: np.less(x[0], 3e-16)
TOL = 1e-8          # BASELINE.json: "adjoint dJ/dt matching to 1e-8 rel"; the same bar for displacement and compliance


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
    # and entry by entry wherever the gradient is not small (99 % of the entries): no cancellation hides behind the max norm
    big = np.abs(ref) > 1e-3 * np.abs(ref).max()
    assert np.abs(dJ[big] / ref[big] - 1.0).max() < 1e-6
    c.close()


@pytest.mark.parametrize("nquad", [None, 4])
def test_parity_triple_against_the_one_million_dof_golden(nquad):
    """BASELINE config 3 -- the bench workload itself, 1 015 470 DOF (tests/golden/make_config3_golden.py: CPU multifrontal
    Cholesky of the oracle's matrix, polished in extended precision to 1e-10).

    nquad None: the rule the mesh asks for and the bench runs -- 5 x 5 Gauss points on these warped cells, within 1e-9 of the
    reference's (nearly) exact integration (linear_shell_model.py:88-103; golden config3_wing1m.npz, made with n = 5).
    nquad 4: the rule of rounds 1-3 against ITS golden (config3_wing1m_n4.npz); the two goldens differ by 7.5e-8 in the
    gradient, which is why n = 4 is no longer the default on warped meshes.

    Tolerance 1e-7, not 1e-8: at this size and slenderness (1.27 mm skin, 6 m span) the discrete solution itself is only
    defined to ~7e-8 in double precision -- changing the oracle's stiffness entries by ONE unit in the last place moves the
    displacement by 7e-8 and the compliance by 8e-8 (scripts/conditioning_floor.py).  The HIP path and the oracle agree to
    1.7e-8 (displacement), 4e-9 (compliance), 2.6e-8 (gradient): closer than two correct fp64 evaluations of the operator
    can be asked to.  (A single direct solve, the reference's own procedure, is another 5-8e-8 away: the golden's first
    refinement step.)  The 1e-8 bar itself is asserted on config 2 above."""
    from bench import make_workload
    from femo_alpha_amd.backend import ShellContext
    tol = 1e-7
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
    c.close()
