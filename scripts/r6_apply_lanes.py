"""k_apply4 with four or five lanes per element (option apply_lanes): interleaved A/B of the operator application (k_apply4 +
k_gather_sum, femo_bench_kernel "apply") on the bench workloads, and the largest difference of K x between the two layouts.
VERDICT r5 item 6 (north-star SpMV): the 25 points of the 5 x 5 rule are 7 + 6 + 6 + 6 on a quad of lanes, 5 + 5 + 5 + 5 + 5 on five."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]] + sys.argv[1:]
names = sys.argv[1:] or ["wing1m", "plate250k", "uquad1m"]
sys.argv = [sys.argv[0]]
from bench import make_workload                                    # noqa: E402
from femo_alpha_amd.backend import ShellContext                    # noqa: E402

for name in names:
    m, fields, marker, desc = make_workload(name)
    ctx = ShellContext(m)
    for k, v in fields.items():
        ctx.set_field(k, v)
    ctx.set_penalty_facets(m.penalty_facets(marker))
    rule = ctx.quadrature()
    rng = np.random.default_rng(7)
    x = rng.standard_normal(m.ndof)
    ys, rows = {}, {4: [], 5: []}
    for rep in range(5):
        for lanes in (4, 5):
            ctx.set_option("apply_lanes", lanes)
            if rep == 0:
                ys[lanes] = ctx.apply_K(x)
            rows[lanes].append(ctx.bench_kernel("apply", 200) * 1e3)
    d = np.abs(ys[4] - ys[5]).max() / np.abs(ys[4]).max()
    print(f"{name}: {m.ndof} DOF, rule {rule}: apply (us) four lanes {np.round(rows[4], 1)}  five lanes {np.round(rows[5], 1)}  "
          f"median {np.median(rows[4]):.1f} -> {np.median(rows[5]):.1f};  K x differs by {d:.1e} (max norm, relative)", flush=True)
    ctx.close()
