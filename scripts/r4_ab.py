"""A/B of schedule options inside ONE process: the un-instrumented factorisation time (femo_frontal_info after femo_factorize) for
several option sets, interleaved over several rounds so that clock and box drift hit all of them alike.
    python scripts/r4_ab.py wing1m "rows_fine_wg=0,narrow_fine_wg=0" "rows_fine_wg=96,narrow_fine_wg=64" ..."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
which = sys.argv[1]
sets = [dict((kv.split("=")[0], float(kv.split("=")[1])) for kv in a.split(",") if kv) for a in sys.argv[2:]]
sys.argv = [sys.argv[0]]
from bench import make_workload
from femo_alpha_amd.backend import ShellContext
m, fields, marker, desc = make_workload(which)
c = ShellContext(m)
for k, v in fields.items():
    c.set_field(k, v)
c.set_penalty_facets(m.penalty_facets(marker))
c.enable_frontal(12)
c.set_solver(preconditioner=2, rtol=1e-10, maxit=30, check_every=1)
defaults = {}
for s in sets:
    for k in s:
        defaults.setdefault(k, None)
times = [[] for _ in sets]
for rnd in range(6):
    for i, s in enumerate(sets):
        for k, v in s.items():
            c.set_option(k, v)
        c.factorize()
        ts = [c.factorize()["factor_ms"] for _ in range(4)]
        if rnd:
            times[i] += ts
for s, t in zip(sets, times):
    print(f"{s}: factor_ms median {np.median(t):.3f}  min {np.min(t):.3f}  (n = {len(t)})", flush=True)
