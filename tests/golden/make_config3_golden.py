"""Generator of the full-size golden for BASELINE config 3: the 1 015 470-DOF wing skin of bench.py (workload "wing1m").

    python tests/golden/make_config3_golden.py [nquad]    (a few minutes of the host's cores, ~20 GB)

``nquad`` Gauss points per direction; default: what the mesh asks for (ShellMesh.recommended_nquad: 5 on the warped cells of
this skin -- the reference integrates its static forms (nearly) exactly, linear_shell_model.py:88-103, and n = 5 is within
1e-9 of that limit here, n = 4 is 7.5e-8 away in the gradient).  ``4`` writes config3_wing1m_n4.npz, the round-1..3 golden
kept as the secondary rule.

SuperLU cannot factorise this matrix (32-bit fill indices), so the state and the adjoint come from the CPU restatement's
own multifrontal Cholesky (oracle/cpu_baseline.py: C++/OpenMP element matrices, LAPACK/BLAS on dense fronts) and are then
polished by iterative refinement with the residual b - K x accumulated in x87 extended precision on the CSR matrix the C++
restatement assembles -- the same recipe as make_fullsize_goldens.py.  ``w_correction`` / ``lam_correction`` record the
size of the last correction relative to the solution.  Stored: compliance, mass, max |w|, 4096 seeded samples of the state,
and the full d compliance / d thickness vector: the north-star triple at the north-star size.

Like the other goldens this pins the HIP path to the CPU oracle, not the oracle to FEniCSx.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
NQUAD = int(sys.argv[1]) if len(sys.argv) > 1 else None
sys.argv = [sys.argv[0]]

from bench import make_workload                                  # noqa: E402
from femo_alpha_amd.solver.symbolic import build_plan            # noqa: E402
from oracle import cpu_baseline as cb                            # noqa: E402
from oracle.rm_shell_oracle import ShellOracle                   # noqa: E402


def refine(K, solve, b, x, steps=10, tol=2e-16):
    if np.finfo(np.longdouble).nmant < 63:
        raise RuntimeError("numpy longdouble is not the x87 80-bit type on this machine")
    data = K.data.astype(np.longdouble)
    xl = x.astype(np.longdouble)
    bl = b.astype(np.longdouble)
    rel = np.inf
    for k in range(steps):
        r = bl - np.add.reduceat(data * xl[K.indices], K.indptr[:-1])
        dx = solve(np.asarray(r, dtype=np.float64))
        xl += dx
        rel = float(np.abs(dx).max() / np.abs(xl).max())
        print(f"    refinement {k}: correction {rel:.1e}", flush=True)
        if rel < tol:
            break
    return np.asarray(xl, dtype=np.float64), rel


def main():
    t0 = time.time()
    m, fields, marker, desc = make_workload("wing1m")
    cores = cb.host_cores()
    nquad = m.recommended_nquad() if NQUAD is None else NQUAD
    o = ShellOracle(m, nquad=nquad, penalty_facets=m.penalty_facets(marker))
    o.set_fields(h=fields["thickness"], E=fields["E"], nu=fields["nu"], rho=fields["density"], f=fields["F_solid"])
    cs = cb.CpuShell(o)
    cs.pattern()
    K = cs.assemble_K(cores).tocsr()
    K.sort_indices()
    b = cs.load_vector(cores)
    mf = cb.CpuMultifrontal(cs, build_plan(m, 12), cores)
    mf.factorize()
    print(f"{desc}\nassembled and factorised in {time.time() - t0:.0f} s", flush=True)
    w, cw = refine(K, mf.solve, b, mf.solve(b))
    J = o.compliance(w)
    rhs = o.dcompliance_du(w)
    lam, cl = refine(K, mf.solve, rhs, mf.solve(rhs))
    dJ = o.dcompliance_dh(w) - cs.assemble_drdfield("h", w, cores).T @ lam
    sample = np.sort(np.random.default_rng(7).choice(m.ndof, size=4096, replace=False))
    print(f"ndof {m.ndof}  J={J:.15e}  corrections w {cw:.1e} lam {cl:.1e}  total {time.time() - t0:.0f} s")
    name = "config3_wing1m.npz" if nquad == m.recommended_nquad() else f"config3_wing1m_n{nquad}.npz"
    np.savez_compressed(os.path.join(os.environ.get("FEMO_GOLDEN_OUT", HERE), name), ndof=m.ndof, nn=m.nn, nel=m.nel, nquad=nquad, compliance=J, mass=o.mass(),
                        w_maxabs=np.abs(w).max(), w_sample_index=sample, w_sample=w[sample],
                        dcompliance_dthickness=dJ, w_correction=cw, lam_correction=cl)


if __name__ == "__main__":
    main()
