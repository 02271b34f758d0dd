"""Plan parameters against time on the GPU: leaf size, axis rule and gap coefficient of the bisection (solver/symbolic.py).  Per
setting: the un-instrumented factorisation (median of 6), one preconditioner application, the forward solve.
    python scripts/r4_plan_sweep.py wing1m"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
which = sys.argv[1] if len(sys.argv) > 1 else "wing1m"
sys.argv = [sys.argv[0]]
from bench import make_workload
from femo_alpha_amd.backend import ShellContext
from femo_alpha_amd.solver.symbolic import build_plan
m, fields, marker, desc = make_workload(which)
settings = [(12, 2, 0.75), (12, 1, 0.75), (12, 2, 0.75), (12, 1, 0.75)] if os.environ.get("FEMO_PLAN_SWEEP") == "rule2" else [(12, 2, 0.75), (12, 1, 0.75), (12, 0, 0.0), (12, 0, 0.75), (12, 1, 0.5), (12, 1, 1.0), (8, 1, 0.75), (16, 1, 0.75), (20, 1, 0.75), (24, 1, 0.75), (32, 1, 0.75)]
for leaf, ar, gap in settings:
    c = ShellContext(m)
    for k, v in fields.items():
        c.set_field(k, v)
    c.set_penalty_facets(m.penalty_facets(marker))
    plan = build_plan(m, leaf, axis_rule=ar, gap=gap)
    c.enable_frontal(leaf, plan=plan)
    c.set_solver(preconditioner=2, rtol=1e-10, maxit=30, check_every=1)
    c.factorize(); c.factorize()
    tf = np.median([c.factorize()["factor_ms"] for _ in range(6)])
    ta = c.factorize()["assemble_ms"]
    sw = np.min([c.sweep_profile().sum() for _ in range(3)])
    h = c.get_field("thickness")
    ts = []
    for _ in range(4):
        c.set_field("thickness", h)
        t0 = time.perf_counter(); it, rr = c.solve_state(True); ts.append(time.perf_counter() - t0)
    s = plan.summary()
    pan = [int(np.ceil(plan.npiv[n].max() / 128)) for n in plan.level_nodes]
    print(f"leaf {leaf:2d} axis {ar} gap {gap:4.2f}: levels {s['levels']:2d} GF {s['factor_gflop']:6.1f} front GB {s['front_GB']:5.2f} panels {sum(pan):3d} | assemble {ta:5.2f} factor {tf:6.2f} "
          f"sweeps {sw:5.2f} forward {np.median(ts[1:]) * 1e3:6.2f} ms ({it} it)", flush=True)
    c.close()
