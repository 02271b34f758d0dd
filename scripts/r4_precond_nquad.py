"""The rule of the FRONT ASSEMBLY (option precond_nquad) against iteration counts, accuracy and time at BASELINE config 3: the factor
is a preconditioner, PCG iterates on the operator's own (5 x 5) residual.
    python scripts/r4_precond_nquad.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench
from femo_alpha_amd.backend import ShellContext
g = np.load(os.path.join(ROOT, "tests", "golden", "config3_wing1m.npz"))
m, fields, marker, desc = bench.make_workload("wing1m")
c = ShellContext(m)
for k, v in fields.items():
    c.set_field(k, v)
c.set_penalty_facets(m.penalty_facets(marker))
c.enable_frontal(12)
c.set_solver(preconditioner=2, rtol=1e-10, maxit=30, check_every=1)
h = c.get_field("thickness")
ref = g["dcompliance_dthickness"]
for pn in (0, 4, 3, 2):
    c.set_option("precond_nquad", pn)
    ts, ta = [], []
    for _ in range(6):
        c.set_field("thickness", h)
        t0 = time.perf_counter(); it, rr = c.solve_state(True); t1 = time.perf_counter()
        dJ, it2, rr2 = c.total_gradient("compliance", "thickness"); t2 = time.perf_counter()
        ts.append(t1 - t0); ta.append(t2 - t1)
    w = c.get_state()
    ew = np.abs(w[g["w_sample_index"]] - g["w_sample"]).max() / float(g["w_maxabs"])
    eJ = abs(c.functional("compliance") / float(g["compliance"]) - 1)
    eg = np.abs(dJ - ref).max() / np.abs(ref).max()
    info = c.frontal_info()
    print(f"precond_nquad {pn}: iterations {it}/{it2}  relres {rr:.1e}/{rr2:.1e}  forward {np.median(ts[1:]) * 1e3:.2f} ms  adjoint {np.median(ta[1:]) * 1e3:.2f} ms  "
          f"assemble {info['assemble_ms']:.2f}  factor {info['factor_ms']:.2f} ms | against the golden: displacement {ew:.1e} compliance {eJ:.1e} gradient {eg:.1e}", flush=True)
