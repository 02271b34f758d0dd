"""the triangular sweeps with 1 / 2 / 4 right-hand sides (csrc/sweeps_multi.h) -- ms per application of the factor
and the factor bytes per second it corresponds to, and the adjoint of 4 outputs in one grouped solve beside one output.

    python scripts/r6_sweeps_nrhs.py [workload] > profiles/r6_sweeps_nrhs.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
WORKLOAD = sys.argv[1] if len(sys.argv) > 1 else "wing1m"
sys.argv = [sys.argv[0]]
from bench import make_workload                                  # noqa: E402
from femo_alpha_amd.backend import ShellContext                  # noqa: E402

m, fields, marker, desc = make_workload(WORKLOAD)
c = ShellContext(m)
for k, v in fields.items():
    c.set_field(k, v)
c.set_penalty_facets(m.penalty_facets(marker))
c.use_direct_solver()
c.solve_state(zero_guess=True)
info = c.frontal_info()
factor_gb = 8.0 * float(np.sum(c.plan.nf.astype(np.int64) * c.plan.npiv)) / 1e9
print(f"{WORKLOAD}: {desc}\nfactor {factor_gb:.2f} GB (read twice per application: forward and backward sweep)\n")
print("vectors   ms per application   ms per vector   factor GB/s   of 8 TB/s")
base = None
for nr in (1, 2, 4):
    ms = c.bench_kernel(f"sweeps{nr}", reps=20)
    base = base or ms
    print(f"{nr:7d} {ms:18.3f} {ms / nr:15.3f} {2 * factor_gb / ms * 1e3:13.0f} {2 * factor_gb / ms * 1e3 / 8000:11.2f}     ({ms / base:.2f} x one vector)")
p1 = c.sweep_profile(); p2 = c.sweep_profile_multi(2); p4 = c.sweep_profile_multi(4)
print("\nper level (ms, event pairs: launch gaps inside):  fronts  wide   1 vector fwd/bwd    2 vectors fwd/bwd    4 vectors fwd/bwd    4 vectors / 1 vector")
for L in range(c.plan.nlevels):
    t = c.plan.level_nodes[L]
    npv = c.plan.npiv[t]
    print(f"  level {L:2d} {len(t):6d}  max npiv {int(npv.max()):5d}   {p1[L, 0]:7.3f} {p1[L, 1]:7.3f}     {p2[L, 0]:7.3f} {p2[L, 1]:7.3f}     {p4[L, 0]:7.3f} {p4[L, 1]:7.3f}"
          f"      {p4[L, 0] / max(p1[L, 0], 1e-9):5.2f} {p4[L, 1] / max(p1[L, 1], 1e-9):5.2f}")
print(f"  sum                               {p1[:, 0].sum():7.3f} {p1[:, 1].sum():7.3f}     {p2[:, 0].sum():7.3f} {p2[:, 1].sum():7.3f}     {p4[:, 0].sum():7.3f} {p4[:, 1].sum():7.3f}")
c.set_stress_params(m=1e-6, rho=6.0)
names = ["compliance", "elastic_energy", "pnorm_stress", "mass"]
c.total_gradients(names, "thickness"); c.total_gradient("compliance", "thickness")
def timed(fn, reps=10):
    ts = []
    for _ in range(reps):
        c.sync(); t0 = time.perf_counter(); fn(); c.sync(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3
t1 = timed(lambda: c.total_gradient("compliance", "thickness"))
t4 = timed(lambda: c.total_gradients(names, "thickness"))
t4s = timed(lambda: [c.total_gradient(nm, "thickness") for nm in names])
print(f"\nadjoint gradient d J / d thickness (host wall-clock, the device-to-host copy of the gradients inside):\n"
      f"  one output (compliance)                    {t1:7.2f} ms\n"
      f"  four outputs, one grouped solve            {t4:7.2f} ms   = {t4 / t1:.2f} x one output\n"
      f"  four outputs, one at a time                {t4s:7.2f} ms   = {t4s / t1:.2f} x one output")
c.close()
