"""Generator of the full-size golden for BASELINE config 3: the 1 015 470-DOF wing skin of bench.py (workload "wing1m").

    python tests/golden/make_config3_golden.py [nquad] [workload]    (a few minutes of the host's cores, ~20 GB)

``workload`` (default wing1m): any single-mesh workload of bench.py -- ``uskin1m`` writes config3_uskin1m.npz, the same surface with an
unstructured (Delaunay) triangulation; that file also stores a checksum of the connectivity, because the triangulation comes from
scipy / qhull and the golden is only valid for the very same mesh.

``nquad`` Gauss points per direction; default: what the mesh asks for (ShellMesh.recommended_nquad: 5 on the warped cells of
this skin -- the reference integrates its static forms (nearly) exactly, linear_shell_model.py:88-103, and n = 5 is within
1e-9 of that limit here, n = 4 is 7.5e-8 away in the gradient).  ``4`` writes config3_wing1m_n4.npz, the round-1..3 golden
kept as the secondary rule.

SuperLU cannot factorise this matrix (32-bit fill indices), so the preconditioner of the refinement is the CPU restatement's own
multifrontal Cholesky (oracle/cpu_baseline.py: C++/OpenMP element matrices, LAPACK/BLAS on dense fronts); the operator the residual
b - K x is formed with is assembled in an extended arithmetic (tests/golden/_extended.py: double-double since round 6 -- portable, and
its refinement converges to 1e-20 where the x87 one stalls at ~5e-11 on this skin; FEMO_GOLDEN_ARITH=x87 for the old one) -- the same recipe as
make_fullsize_goldens.py: the stored numbers are the solution of the discrete problem, not of a float64-rounded matrix.  ``w_correction`` / ``lam_correction`` record the
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
NQUAD = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] not in ("", "auto") else None
WORKLOAD = sys.argv[2] if len(sys.argv) > 2 else "wing1m"
sys.argv = [sys.argv[0]]

from bench import make_workload                                  # noqa: E402
from femo_alpha_amd.solver.symbolic import build_plan            # noqa: E402
from oracle import cpu_baseline as cb                            # noqa: E402
from oracle.rm_shell_oracle import ShellOracle                   # noqa: E402
sys.path.insert(0, HERE)
from _extended import as_float64, extended_system, operator_from_float64, refine                    # noqa: E402


def main():
    t0 = time.time()
    # the unstructured skins: the Delaunay triangulation comes from scipy / qhull and goes into the golden, so that the test runs on the
    # golden's own mesh whatever qhull is installed there
    tri = None
    if WORKLOAD in ("uskin1m", "uquad1m"):
        from femo_alpha_amd.mesh import skin_triangulation
        tri = skin_triangulation(*{"uskin1m": (116, 580), "uquad1m": (47, 239)}[WORKLOAD])[2]
    m, fields, marker, desc = make_workload(WORKLOAD, tri=tri)
    cores = cb.host_cores()
    nquad = m.recommended_nquad() if NQUAD is None else NQUAD
    o = ShellOracle(m, nquad=nquad, penalty_facets=m.penalty_facets(marker))
    o.set_fields(h=fields["thickness"], E=fields["E"], nu=fields["nu"], rho=fields["density"], f=fields["F_solid"])
    cs = cb.CpuShell(o)
    Kx, bx = extended_system(cs, cores)
    mf = cb.CpuMultifrontal(cs, build_plan(m, m.recommended_leaf_size()), cores)
    mf.factorize()
    print(f"{desc}\nassembled (extended precision) and factorised in {time.time() - t0:.0f} s", flush=True)
    say = lambda msg: print(msg, flush=True)
    w, cw = refine(Kx, mf.solve, bx, mf.solve(as_float64(bx)), log=say)
    J = o.compliance(w)
    rhs = o.dcompliance_du(w)
    lam, cl = refine(Kx, mf.solve, rhs, mf.solve(rhs), log=say)
    dJ = o.dcompliance_dh(w) - cs.assemble_drdfield("h", w, cores).T @ lam
    # for the record: the solution of the FLOAT64-assembled matrix (the round 1-3 goldens) against the one above
    K64 = cs.assemble_K(cores).tocsr(); K64.sort_indices()
    b64 = cs.load_vector(cores)
    w64, _ = refine(operator_from_float64(K64, cs, cores), mf.solve, b64, mf.solve(b64), steps=6)
    d64 = (float(np.abs(w64 - w).max() / np.abs(w).max()), float(abs(o.compliance(w64) - J) / abs(J)))
    print(f"float64-assembled matrix: solution off by {d64[0]:.1e} (displacement) {d64[1]:.1e} (compliance)", flush=True)
    sample = np.sort(np.random.default_rng(7).choice(m.ndof, size=4096, replace=False))
    print(f"ndof {m.ndof}  J={J:.15e}  corrections w {cw:.1e} lam {cl:.1e}  total {time.time() - t0:.0f} s")
    name = f"config3_{WORKLOAD}.npz" if nquad == m.recommended_nquad() else f"config3_{WORKLOAD}_n{nquad}.npz"
    import hashlib
    mesh_sha = hashlib.sha256(np.ascontiguousarray(m.cells, dtype=np.int64).tobytes() + np.ascontiguousarray(m.nodes).tobytes()).hexdigest()
    np.savez_compressed(os.path.join(os.environ.get("FEMO_GOLDEN_OUT", HERE), name), ndof=m.ndof, nn=m.nn, nel=m.nel, nquad=nquad, compliance=J, mass=o.mass(),
                        w_maxabs=np.abs(w).max(), w_sample_index=sample, w_sample=w[sample],
                        dcompliance_dthickness=dJ, w_correction=cw, lam_correction=cl,
                        float64_matrix_distance_w=d64[0], float64_matrix_distance_compliance=d64[1], mesh_sha256=mesh_sha,
                        arithmetic=Kx.__class__.__name__.strip("_").lower(), **({} if tri is None else {"triangulation": np.asarray(tri, dtype=np.int32)}))


if __name__ == "__main__":
    main()
