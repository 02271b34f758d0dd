"""Per-level times of one preconditioner application (forward and backward sweep, both launches of the wide levels) with the
factor bytes each level reads: where the application is below the HBM roof.   python scripts/r3_sweeps.py [wing1m] [key=value ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_workload
from femo_alpha_amd.backend import ShellContext

which = sys.argv[1] if len(sys.argv) > 1 else "wing1m"
opts = {a.split("=")[0]: float(a.split("=")[1]) for a in sys.argv[2:]}
m, fields, marker, desc = make_workload(which)
c = ShellContext(m)
for k, v in fields.items():
    c.set_field(k, v)
c.set_penalty_facets(m.penalty_facets(marker))
pre = {k: v for k, v in opts.items() if k in ("wide_np", "wide_cnt", "swork_slots")}
plan = c.enable_frontal(12, **pre)
for k, v in opts.items():
    if k not in pre:
        c.set_option(k, v)
print("options", opts)
c.set_solver(preconditioner=2, rtol=1e-10, maxit=30, check_every=1)
c.factorize()
t = np.min([c.sweep_profile(detail=True) for _ in range(5)], axis=0)
print("level  cnt   max_np  max_nb   factor MB | fwd a   fwd b  | bwd a   bwd b  (us) |  GB/s fwd   GB/s bwd")
tot_b = 0
for L in range(plan.nlevels):
    ts = plan.level_nodes[L]
    nf, npv = plan.nf[ts].astype(float), plan.npiv[ts].astype(float)
    by = float((nf * npv).sum() * 8)
    tot_b += by
    f, b = t[L, 0] + t[L, 1], t[L, 2] + t[L, 3]
    print(f"{L:5d} {len(ts):5d} {int(npv.max()):7d} {int((nf - npv).max()):7d} {by / 1e6:10.1f} | {t[L,0]*1e3:6.1f} {t[L,1]*1e3:6.1f} | {t[L,2]*1e3:6.1f} {t[L,3]*1e3:6.1f}      | {by / f / 1e6:8.0f} {by / b / 1e6:8.0f}")
ts = [c.factorize()["factor_ms"] for _ in range(4)]
print("factor_ms:", " ".join(f"{x:.2f}" for x in ts))
print(f"total: forward {t[:, :2].sum():.3f} ms, backward {t[:, 2:].sum():.3f} ms, factor {tot_b / 1e9:.2f} GB per sweep -> {2 * tot_b / t.sum() / 1e6:.0f} GB/s")
