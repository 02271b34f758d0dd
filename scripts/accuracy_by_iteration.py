"""How good is the solution after 1, 2, 3 PCG iterations on the multifrontal factor?  (a) the 255 k-DOF plate against its
committed golden (displacement samples, compliance, d compliance / d thickness), (b) the 1 M-DOF wing against its own
converged solution."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench
from femo_alpha_amd.backend import ShellContext
from femo_alpha_amd.mesh import plate_mesh

g = np.load(os.path.join(ROOT, "tests", "golden", "config2_plate_58x290_nodal.npz"))
m = plate_mesh(2.0, 10.0, int(g["nx"]), int(g["ny"]))
c = ShellContext(m)
for k, v in dict(thickness=g["thickness"], E=[1e8], nu=[0.3], density=[10.0], F_solid=np.tile([0.0, 0.0, 5.0], (m.nn, 1))).items():
    c.set_field(k, v)
c.set_penalty_facets(m.penalty_facets(lambda x: np.less(x[0], 3e-16)))
c.enable_frontal()
c.set_option("strict", 0)
for it in (1, 2, 3):
    c.set_solver(preconditioner=2, rtol=1e-30, maxit=it, check_every=1)
    n, rr = c.solve_state(zero_guess=True)
    w = c.get_state()
    ew = np.abs(w[g["w_sample_index"]] - g["w_sample"]).max() / float(g["w_maxabs"])
    J = c.functional("compliance")
    dJ, n2, rr2 = c.total_gradient("compliance", "thickness")
    ref = g["dcompliance_dthickness"]
    print(f"plate250k  {it} it: relres {rr:.1e}  |w - golden| {ew:.1e}  J {abs(J / float(g['compliance']) - 1):.1e}  dJ/dh {np.abs(dJ - ref).max() / np.abs(ref).max():.1e} (adjoint relres {rr2:.1e})", flush=True)
c.close()

m, fields, marker, desc = bench.make_workload("wing1m")
c = ShellContext(m)
for k, v in fields.items():
    c.set_field(k, v)
c.set_penalty_facets(m.penalty_facets(marker))
c.enable_frontal()
c.set_option("strict", 0)
sol = {}
for it in (4, 1, 2, 3):
    c.set_solver(preconditioner=2, rtol=1e-30, maxit=it, check_every=1)
    n, rr = c.solve_state(zero_guess=True)
    w = c.get_state()
    J = c.functional("compliance")
    dJ, n2, rr2 = c.total_gradient("compliance", "thickness")
    sol[it] = (w, J, dJ)
    if it != 4:
        w4, J4, dJ4 = sol[4]
        nu = m.ndof_u if hasattr(m, "ndof_u") else len(w)
        print(f"wing1m  {it} it: relres {rr:.1e}  |w - w4|/|w4| {np.abs(w - w4).max() / np.abs(w4).max():.1e}  J {abs(J / J4 - 1):.1e}  dJ/dh {np.abs(dJ - dJ4).max() / np.abs(dJ4).max():.1e}", flush=True)
