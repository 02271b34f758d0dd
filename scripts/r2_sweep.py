"""Sweep of the schedule switches on one workload: median forward solve and preconditioner application per setting."""
import sys, os, time, itertools
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_workload
from femo_alpha_amd.backend import ShellContext

which = sys.argv[1] if len(sys.argv) > 1 else "wing1m"
m, fields, marker, desc = make_workload(which)


def run(leaf, opts):
    c = ShellContext(m)
    for k, v in fields.items():
        c.set_field(k, v)
    c.set_penalty_facets(m.penalty_facets(marker))
    pre = {k: v for k, v in opts.items() if k in ("wide_np", "wide_cnt", "swork_slots")}
    c.enable_frontal(leaf, **pre)
    for k, v in opts.items():
        if k not in pre:
            c.set_option(k, v)
    c.set_solver(preconditioner=2, rtol=1e-10, maxit=50, check_every=1)
    h = c.get_field("thickness")
    ts, its = [], 0
    for _ in range(6):
        c.set_field("thickness", h)
        t0 = time.perf_counter(); its, rr = c.solve_state(True); ts.append(time.perf_counter() - t0)
    tm = c.last_timing()
    g, it2, _ = c.total_gradient("compliance", "thickness")
    t0 = time.perf_counter(); c.total_gradient("compliance", "thickness"); ta = time.perf_counter() - t0
    sw = np.min([c.sweep_profile().sum() for _ in range(3)])
    c.close()
    return np.median(ts[1:]) * 1e3, tm["setup_ms"], tm["krylov_ms"], ta * 1e3, sw, its


base = dict()
cases = [(12, {}), (12, dict(left_min=32)), (12, dict(left_min=128)), (12, dict(left_min=256, super_panel_cnt=256)), (12, dict(super_panel=256)), (12, dict(super_panel=384)),
         (12, dict(super_panel=768)), (12, dict(left_max=4096)), (12, dict(left_max=1024)), (10, {}), (14, {}), (16, {}), (12, dict(fused_schur=0))]
if len(sys.argv) > 2 and sys.argv[2] == "all":
    cases += [(8, {}), (16, {}), (24, {}),
         (12, dict(left_max=8192)), (12, dict(left_max=100000)), (12, dict(left_min=8)), (12, dict(left_min=32)),
         (12, dict(lookahead_cnt=8)), (12, dict(lookahead_cnt=32)), (12, dict(lookahead=0)),
         (12, dict(wide_cnt=1024)), (12, dict(wide_cnt=256)), (12, dict(wide_cnt=2048)), (12, dict(wide_cnt=4096)),
         (12, dict(xinv_small_cnt=8)), (12, dict(xinv_small_cnt=128))]
for leaf, opts in cases:
    f, su, kr, ad, sw, its = run(leaf, opts)
    print(f"leaf {leaf:3d} {str(opts):60s} forward {f:7.2f} ms (factor {su:6.2f} + pcg {kr:5.2f}, {its} it)  adjoint {ad:6.2f} ms  apply {sw:5.2f} ms", flush=True)
