"""Relative residual of the preconditioned CG after 1, 2, 3 iterations (how accurate is one application of the factor?)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
sys.argv = [sys.argv[0]]
import bench
for wl in ("plate250k", "wing1m", "wing4m"):
    m, fields, marker, desc = bench.make_workload(wl)
    from femo_alpha_amd.backend import ShellContext
    c = ShellContext(m)
    for k, v in fields.items():
        c.set_field(k, v)
    c.set_penalty_facets(m.penalty_facets(marker))
    c.enable_frontal()
    out = []
    for it in (1, 2, 3):
        c.set_solver(preconditioner=2, rtol=1e-30, maxit=it, check_every=1)
        out.append(c.solve_state(zero_guess=True)[1])
    print(wl, ["%.1e" % r for r in out], flush=True)
    c.close()
