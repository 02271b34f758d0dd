"""Per-step times of consecutive cold forward solves on the native single-GPU path (looking for periodic stalls)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]]
import bench
from femo_alpha_amd.backend import ShellContext
m, fields, marker, desc = bench.make_workload("wing1m")
c = ShellContext(m)
for k, v in fields.items():
    c.set_field(k, v)
c.set_penalty_facets(m.penalty_facets(marker))
c.enable_frontal(12)
c.set_solver(preconditioner=2, rtol=1e-10, maxit=30, check_every=1)
h = c.get_field("thickness")
out = []
for rep in range(24):
    c.set_field("thickness", h)
    t0 = time.perf_counter(); c.solve_state(True); t1 = time.perf_counter()
    out.append(1e3 * (t1 - t0))
print(" ".join(f"{x:.1f}" for x in out))
