"""A/B of option "sweep_graph" (the preconditioner application of the PCG loop replayed as a HIP graph): PCG loop time of the forward
solve at 1 M DOF, factor reused.   python scripts/r3_graph_ab.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_workload
from femo_alpha_amd.backend import ShellContext

m, fields, marker, desc = make_workload("wing1m")
c = ShellContext(m)
for k, v in fields.items():
    c.set_field(k, v)
c.set_penalty_facets(m.penalty_facets(marker))
c.enable_frontal(12)
c.set_solver(preconditioner=2, rtol=1e-10, maxit=30, check_every=1)
c.factorize()
for g in (0, 1, 0, 1):
    c.set_option("sweep_graph", g)
    ts = []
    for _ in range(8):
        it, rr = c.solve_state(True)
        ts.append(c.last_timing()["krylov_ms"])
    print(f"sweep_graph={g}: {it} iterations, PCG loop ms: " + " ".join(f"{t:.3f}" for t in ts), flush=True)
w = c.get_state()
print("state norm", float(np.linalg.norm(w)))
