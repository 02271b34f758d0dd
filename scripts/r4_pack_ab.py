"""A/B of option "assemble_pack" (several elements per 128-thread workgroup in the front assembly): assemble_ms / factor_ms and the
compliance of the solved state, off / on / off / on in one process."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from femo_alpha_amd.backend import ShellContext
for w in sys.argv[1:] or ("plate250k", "wing1m", "uskin1m"):
    m, fields, marker, _ = bench.make_workload(w)
    c = ShellContext(m)
    for k, v in fields.items(): c.set_field(k, v)
    c.set_penalty_facets(m.penalty_facets(marker))
    c.enable_frontal()
    c.set_solver(preconditioner=2, rtol=1e-10, maxit=30, check_every=1)
    res = {}
    for pack in (0, 1, 0, 1):
        c.set_option("assemble_pack", pack)
        c.factorize()
        t = [c.factorize() for _ in range(6)]
        it, rr = c.solve_state(True)
        res.setdefault(pack, []).append((np.median([x["assemble_ms"] for x in t]), np.median([x["factor_ms"] for x in t]), it, c.functional("compliance")))
    print(w, {k: [(round(a, 3), round(b, 3), i, repr(j)) for a, b, i, j in v] for k, v in res.items()}, flush=True)
    c.close()
