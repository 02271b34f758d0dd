"""Per-level times of one factorisation by kernel class under a given set of schedule options.
    python scripts/r2_levels.py wing1m [key=value ...]"""
import os, re, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_workload
from femo_alpha_amd.backend import ShellContext

which = sys.argv[1] if len(sys.argv) > 1 else "wing1m"
opts = {a.split("=")[0]: float(a.split("=")[1]) for a in sys.argv[2:]}
if "node_order" in opts:                     # rows of a front by ascending node id (0, rounds 1-5) or along the separators (1): a plan option
    from femo_alpha_amd.solver import symbolic as _sym
    _sym.NODE_ORDER = int(opts.pop("node_order"))
    print("node_order", _sym.NODE_ORDER)
m, fields, marker, desc = make_workload(which)
c = ShellContext(m)
for k, v in fields.items():
    c.set_field(k, v)
c.set_penalty_facets(m.penalty_facets(marker))
pre = {k: v for k, v in opts.items() if k in ("wide_np", "wide_cnt", "swork_slots")}
plan = c.enable_frontal(int(opts.pop("leaf", 12)), **pre)
for k, v in opts.items():
    if k not in pre:
        c.set_option(k, v)
c.set_solver(preconditioner=2, rtol=1e-10, maxit=30, check_every=1)
c.factorize(); c.factorize()
c.set_option("profile_verbose", 1)
sys.stderr.flush()
tmp = tempfile.TemporaryFile(mode="w+b")
old = os.dup(2); os.dup2(tmp.fileno(), 2)
p = c.factorize_profile()
os.dup2(old, 2); os.close(old)
c.set_option("profile_verbose", 0)
tmp.seek(0)
lev = {}
for line in tmp.read().decode().splitlines():
    mm = re.match(r"prof level (\d+) class (\d+) ([\d.]+) us", line)
    if mm:
        L, cl, us = int(mm[1]), int(mm[2]), float(mm[3])
        a = lev.setdefault(L, [[0.0, 0] for _ in range(7)])
        a[cl][0] += us; a[cl][1] += 1
names = ["rows", "diag", "trail", "extend", "assemble", "zero", "xinv"]
print("options", opts)
print("level  cnt  " + " ".join(f"{n:>13s}" for n in names) + "   trailing TF/s (approx)")
for L in sorted(lev):
    t = plan.level_nodes[L]
    nf, npv = plan.nf[t].astype(float), plan.npiv[t].astype(float)
    fl = (npv * nf * nf - npv * npv * nf + npv ** 3 / 3).sum() - (npv * 128 * 128).sum() - ((nf - npv / 2) * npv * 128).sum()
    tr = lev[L][2][0]
    print(f"{L:5d} {len(t):5d} " + " ".join(f"{a[0]:8.0f}/{a[1]:<4d}" for a in lev[L]) + (f"   {fl / 1e6 / tr:6.1f}" if tr > 0 else ""))
print("class totals (ms):", {n: round(p[k]["ms"], 3) for n, k in zip(names, ["panel_rows", "panel_diag", "trailing", "extend_add", "front_assemble", "memset", "l11_inverse"])})
print("trailing %.1f GF -> %.2f TFLOP/s" % (p["trailing_flops"] / 1e9, p["trailing_flops"] / (p["trailing"]["ms"] * 1e-3) / 1e12))
ts = []
for _ in range(5):
    ts.append(c.factorize()["factor_ms"])
print("factor_ms (unprofiled):", " ".join(f"{t:.2f}" for t in ts))
ts = []
for _ in range(5):
    it, rr = c.solve_state(zero_guess=True)
    ts.append(c.last_timing())
print("forward solve (assemble + factorise + PCG, ms):", " ".join(f"{t['setup_ms'] + t['krylov_ms']:.2f}" for t in ts), " PCG:", " ".join(f"{t['krylov_ms']:.2f}" for t in ts), f" iterations {it}")
