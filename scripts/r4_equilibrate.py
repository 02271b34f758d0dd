"""does factorising the symmetrically equilibrated operator D K D, D = diag(K)^-1/2, make ONE application of
the factor accurate enough to drop the second PCG iteration?  For config 2 (against its golden) and config 3 (against its golden):
option "equilibrate" 0 (off), 1 (D = diag^-1/2) and 2 (D rounded to powers of two), after 1 and 2 iterations: the relative
residual of the recurrence, and the distance of displacement / compliance / gradient from the golden.
    python scripts/r4_equilibrate.py > profiles/r4_equilibrate.txt"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench
from femo_alpha_amd.backend import ShellContext
from femo_alpha_amd.mesh import plate_mesh


def table(name, c, g):
    c.enable_frontal()
    c.set_option("strict", 0)
    ref = g["dcompliance_dthickness"]
    for mode in (0, 1, 2):
        c.set_option("equilibrate", mode)
        for it in (1, 2):
            c.set_solver(preconditioner=2, rtol=1e-30, maxit=it, check_every=1)
            n, rr = c.solve_state(zero_guess=True)
            w = c.get_state()
            ew = np.abs(w[g["w_sample_index"]] - g["w_sample"]).max() / float(g["w_maxabs"])
            J = c.functional("compliance")
            dJ, n2, rr2 = c.total_gradient("compliance", "thickness")
            print(f"{name}  equilibrate {mode}  {it} iteration(s): relres {rr:.2e}  |w - golden| {ew:.2e}  compliance {abs(J / float(g['compliance']) - 1):.2e}  "
                  f"gradient {np.abs(dJ - ref).max() / np.abs(ref).max():.2e}  (adjoint relres {rr2:.2e})", flush=True)
    c.close()


g = np.load(os.path.join(ROOT, "tests", "golden", "config2_plate_58x290_nodal.npz"))
m = plate_mesh(2.0, 10.0, int(g["nx"]), int(g["ny"]))
c = ShellContext(m)
for k, v in dict(thickness=g["thickness"], E=[1e8], nu=[0.3], density=[10.0], F_solid=np.tile([0.0, 0.0, 5.0], (m.nn, 1))).items():
    c.set_field(k, v)
c.set_penalty_facets(m.penalty_facets(lambda x: np.less(x[0], 3e-16)))
table("plate250k", c, g)

g = np.load(os.path.join(ROOT, "tests", "golden", "config3_wing1m.npz"))
m, fields, marker, desc = bench.make_workload("wing1m")
c = ShellContext(m)
for k, v in fields.items():
    c.set_field(k, v)
c.set_penalty_facets(m.penalty_facets(marker))
table("wing1m", c, g)
