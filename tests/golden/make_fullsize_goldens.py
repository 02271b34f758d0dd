"""Generator of the full-size goldens for BASELINE configs 1 and 2.

    python tests/golden/make_fullsize_goldens.py          (about 3 minutes of one host core, 3 GB)

The CPU oracle (oracle/rm_shell_oracle.py) is run on
  config 1   2 x 10 plate, 10 x 50 quads, 8 046 DOF, h = 0.1 uniform (nodal)
  config 2   2 x 10 plate, 58 x 290 quads, 255 438 DOF, h_i = 0.1 (1 + 0.2 U(-1,1)), default_rng(0),
             nodal AND element-wise thickness
with E = 1e8, nu = 0.3, rho = 10, pressure (0,0,5), clamp x0 <= 3e-16 by the 1e15 penalty
(reference examples/advanced_examples/simple_shell_opt/ex_simple_shell_opt.py:39-65), and the parity triple of
BASELINE.json is stored: displacement (max |w| and 4096 seeded samples), compliance, mass, and the full
d compliance / d thickness vector.

Accuracy of the stored numbers: they are the solution of the discrete problem itself, not of a float64 matrix.  The stiffness
matrix and the load vector are formed in double-double arithmetic (tests/golden/_extended.py, oracle/cpu_kernels.cpp
cpu_assemble_csr_dd; x87 long double, cpu_assemble_csr_ld, until round 5 and still with FEMO_GOLDEN_ARITH=x87); SuperLU on the float64 matrix is only the preconditioner of an iterative refinement whose residual is
accumulated in extended precision on that operator.  (Round 1-3 goldens refined against the float64-assembled matrix: the rounding
of its entries alone moved config 2 by ~1e-8 -- swapping numpy's 4-point Gauss table for the exact one, a change of 1e-16, moved
compliance by 8e-9.)  ``w_correction`` / ``lam_correction`` record the size of the last correction relative to the solution.

These files pin the HIP path to the oracle at the north-star tolerance at 250 k DOF; they do NOT pin the oracle to
FEniCSx (parity stays "unpinned" in that sense: DESIGN.md section 2).
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from femo_alpha_amd.mesh import plate_mesh                      # noqa: E402
from oracle import cpu_baseline as cb                            # noqa: E402
from oracle.rm_shell_oracle import ShellOracle                   # noqa: E402
sys.path.insert(0, HERE)
from _extended import as_float64, extended_system, operator_from_float64, refine                    # noqa: E402

CLAMP = lambda x: np.less(x[0], 3e-16)


def run(nx, ny, element_wise, random_thickness):
    m = plate_mesh(2.0, 10.0, nx, ny)
    n_h = m.nel if element_wise else m.nn
    h = 0.1 * (1 + 0.2 * np.random.default_rng(0).uniform(-1, 1, n_h)) if random_thickness else np.full(n_h, 0.1)
    f = np.tile([0.0, 0.0, 5.0], (m.nn, 1))
    o = ShellOracle(m, element_wise_material=element_wise, penalty_facets=m.penalty_facets(CLAMP))
    o.set_fields(h=h, E=1e8, nu=0.3, rho=10.0, f=f)
    t0 = time.time()
    lu = o.factorize()
    cs = cb.CpuShell(o)
    Kx, bx = extended_system(cs, cb.host_cores())
    w, cw = refine(Kx, lu.solve, bx, lu.solve(as_float64(bx)))
    J = o.compliance(w)
    rhs = o.dcompliance_du(w)
    lam, cl = refine(Kx, lu.solve, rhs, lu.solve(rhs))
    dJ = o.dcompliance_dh(w) - o.dRdfield_T("h", w, lam)
    # for the record: how far the solution of the FLOAT64-assembled matrix (the round 1-3 goldens; what any float64 code can hope
    # to reproduce) sits from the one above
    K64 = o._K.tocsr(); K64.sort_indices()
    w64, _ = refine(operator_from_float64(K64, cs, cb.host_cores()), lu.solve, o.load_vector(), lu.solve(o.load_vector()))
    d64 = (float(np.abs(w64 - w).max() / np.abs(w).max()), float(abs(o.compliance(w64) - J) / abs(J)))
    sample = np.sort(np.random.default_rng(7).choice(m.ndof, size=min(4096, m.ndof), replace=False))
    print(f"{nx}x{ny} element_wise={element_wise}: ndof {m.ndof}  {time.time() - t0:.0f} s  J={J:.15e}  "
          f"corrections w {cw:.1e} lam {cl:.1e}  "
          f"float64-matrix solution off by {d64[0]:.1e} (displacement) {d64[1]:.1e} (compliance)")
    return dict(nx=nx, ny=ny, element_wise=element_wise, ndof=m.ndof, thickness=h, compliance=J, mass=o.mass(),
                elastic_energy=o.elastic_energy(w), w_maxabs=np.abs(w).max(), u_maxabs=np.abs(w[:m.ndof_u]).max(),
                w_sample_index=sample, w_sample=w[sample], dcompliance_dthickness=dJ,
                w_correction=cw, lam_correction=cl, float64_matrix_distance_w=d64[0], float64_matrix_distance_compliance=d64[1])


def main():
    cases = {"config1_plate_10x50_nodal": run(10, 50, False, False),
             "config2_plate_58x290_nodal": run(58, 290, False, True),
             "config2_plate_58x290_elementwise": run(58, 290, True, True)}
    for name, d in cases.items():
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)


if __name__ == "__main__":
    main()
