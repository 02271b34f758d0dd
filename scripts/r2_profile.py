"""Round-2 profile of the frontal path on one workload: per-class and per-level times of one factorisation
(HIP events around every launch) and per-level times of the two triangular sweeps.
    python scripts/r2_profile.py [wing1m|plate250k] [out.json]"""
import json, os, re, sys, subprocess, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_workload
from femo_alpha_amd.backend import ShellContext

which = sys.argv[1] if len(sys.argv) > 1 else "wing1m"
out = sys.argv[2] if len(sys.argv) > 2 else None
m, fields, marker, desc = make_workload(which)
c = ShellContext(m)
for k, v in fields.items():
    c.set_field(k, v)
c.set_penalty_facets(m.penalty_facets(marker))
plan = c.enable_frontal(12)
c.set_solver(preconditioner=2, rtol=1e-10, maxit=30, check_every=1)
c.factorize(); c.factorize()
# per-launch lines go to stderr of this process: capture through a temp file by dup2
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
print("level  cnt  " + " ".join(f"{n:>13s}" for n in names))
for L in sorted(lev):
    cnt = len(plan.level_nodes[L]) if L < plan.nlevels else 0
    print(f"{L:5d} {cnt:5d} " + " ".join(f"{a[0]:8.0f}/{a[1]:<4d}" for a in lev[L]))
tot = {n: p[k]["ms"] for n, k in zip(names, ["panel_rows", "panel_diag", "trailing", "extend_add", "front_assemble", "memset", "l11_inverse"])}
print("class totals (ms):", {k: round(v, 3) for k, v in tot.items()}, "sum", round(sum(tot.values()), 3))
print("flops: trailing %.1f GF rows %.1f GF diag %.1f GF" % (p["trailing_flops"] / 1e9, p["panel_rows_flops"] / 1e9, p["panel_diag_flops"] / 1e9))
print("trailing TFLOP/s:", p["trailing_flops"] / (p["trailing"]["ms"] * 1e-3) / 1e12)
info = c.factorize()
print("factorize:", info)
for thr in (0, 1 << 30):
    c.set_option("bnd_tiled_nb", thr)
    s4 = np.min([c.sweep_profile(detail=True) for _ in range(3)], axis=0)
    print(f"bnd_tiled_nb {thr}: L21^T x per level (us):", " ".join(f"{x*1e3:.0f}" for x in s4[:, 2]), " total apply ms", s4.sum())
c.set_option("bnd_tiled_nb", 1 << 30)
sw4 = np.min([c.sweep_profile(detail=True) for _ in range(3)], axis=0)
sw = np.stack([sw4[:, 0] + sw4[:, 1], sw4[:, 2] + sw4[:, 3]], axis=1)
print("sweeps per level (us): fwd (X b | L21 y)  bwd (L21^T x | X^T s)")
for L in range(plan.nlevels):
    nf, npv = plan.nf[plan.level_nodes[L]].astype(float), plan.npiv[plan.level_nodes[L]].astype(float)
    gb = (nf * npv).sum() * 8 / 1e9
    print(f"{L:5d} cnt {len(nf):5d} maxnp {int(npv.max()):5d} factor {gb:6.3f} GB  fwd {sw4[L,0]*1e3:7.1f} {sw4[L,1]*1e3:7.1f}  bwd {sw4[L,2]*1e3:7.1f} {sw4[L,3]*1e3:7.1f}   {2*gb/(sw[L].sum()*1e-3)/1e3:6.2f} TB/s")
print("sweep total ms", sw.sum(), "fwd", sw[:, 0].sum(), "bwd", sw[:, 1].sum())
it, rr = c.solve_state(True)
print("solve", it, rr, c.last_timing())
if out:
    json.dump(dict(workload=which, classes_ms=tot, levels_us={str(L): [[a[0], a[1]] for a in lev[L]] for L in lev},
                   trailing_flops=p["trailing_flops"], sweeps_ms=sw.tolist(), info={k: float(v) for k, v in info.items()}),
              open(out, "w"), indent=1)
