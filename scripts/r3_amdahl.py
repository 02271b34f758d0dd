"""Strong-scaling bound of BASELINE config 4 from single-GPU measurements (no multi-GPU node is available to the builder).

For P in 2, 4, 8 partitions of the 1 M-DOF wing skin every rank's LOCAL work is run alone on the one GPU (its subtree:
assembly + factorisation of levels [0, nl), its share of the sweeps and of the element operator), then rank 0 receives the
packed Schur complements of the other subtree roots (device-to-device, same process) and factorises the REPLICATED top of the
tree, which every rank repeats.  From these times, the bytes of the Schur all-gather and an assumed cost per collective the
script composes the forward solve at P GPUs and the part of it that does not shrink with P.

    python scripts/r3_amdahl.py [wing1m] > profiles/r3_amdahl_wing1m.json      (markdown table on stderr)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_workload                                   # noqa: E402
from femo_alpha_amd.parallel import HipEngine                     # noqa: E402
from femo_alpha_amd.solver.symbolic import analyse, rank_plan     # noqa: E402

COLLECTIVE_US = 30.0        # assumed latency of one small RCCL all-reduce over xGMI at 8 ranks (not measurable here)
LINK_GBS = 100.0            # assumed achieved rate of one xGMI link (peak ~153 GB/s per direction)
which = sys.argv[1] if len(sys.argv) > 1 else "wing1m"
weak = len(sys.argv) > 2 and sys.argv[2] == "weak"          # weak scaling: P times the span, one 1 M-DOF partition per rank
m, fields, marker, _ = make_workload(which)
leaf = 12


def weak_workload(P):
    from femo_alpha_amd.mesh import wing_skin_mesh
    mm = wing_skin_mesh(116, 580 * P, span=6.0 * P).renumbered()[0]
    ff = dict(thickness=[1.27e-3], E=[73.1e9], nu=[0.33], density=[2780.0], F_solid=np.tile([0.0, 0.0, -2780.0 * 1.27e-3 * 9.81], (mm.nn, 1)))
    return mm, ff


def engine(P, tree, rank):
    sub, plan, info = rank_plan(m, tree, rank, P)
    eng = HipEngine(sub, plan, info)
    sel = info["vertices"]
    for k in ("thickness", "E", "nu", "density"):
        v = np.asarray(fields[k], dtype=np.float64).ravel()
        eng.set_field(k, v if v.size == 1 else v[sel])
    eng.set_field("F_solid", np.asarray(fields["F_solid"]).reshape(-1, 3)[sel])
    eng.set_penalty_facets(sub.penalty_facets(marker), 1e15)
    eng.dist_setup(info["top_local"], P, info["n_local_levels"], sel)
    return eng, sub, plan, info


def timed(fn, reps=5):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


rows = []
for P in (1, 2, 4, 8):
    if weak:
        m, fields = weak_workload(P)
    tree = analyse(m, leaf, min_depth=int(np.log2(P)))
    packed, local_ms, cap = {}, [], 0
    eng0 = None
    for rank in range(P - 1, -1, -1):                 # rank 0 last: its engine stays for the top of the tree
        eng, sub, plan, info = engine(P, tree, rank)
        nl, nlev = info["n_local_levels"], plan.nlevels
        sizes = info["schur_sizes"]
        cap = max(max(n * (n + 1) // 2 for n in sizes), 1)
        eng.factor(0, nl, True)                      # warm-up
        t = timed(lambda: eng.factor(0, nl, True), 3)
        local_ms.append(t)
        buf = eng.new_tensor(cap)
        eng.schur_pack(info["root_front"], buf)
        eng.ctx.sync()
        packed[rank] = buf
        if rank == 0:
            eng0, info0, plan0, sub0 = eng, info, plan, sub
        else:
            eng.ctx.close()
    nl, nlev = info0["n_local_levels"], plan0.nlevels
    for q in range(1, P):
        eng0.block_unpack(info0["stub_fronts"][q], packed[q])

    def top():
        for q in range(1, P):
            eng0.block_unpack(info0["stub_fronts"][q], packed[q])
        eng0.factor(nl, nlev, False)
    top()
    top_ms = timed(top, 3) if P > 1 else 0.0
    c = eng0.ctx
    sw = {}
    for name, (l0, l1) in (("local", (0, nl)), ("top", (nl, nlev))):
        f = timed(lambda: c.frontal_sweep("z", l0, l1, False), 5) if l1 > l0 else 0.0
        b = timed(lambda: c.frontal_sweep("z", l0, l1, True), 5) if l1 > l0 else 0.0
        sw[name] = f + b
    apply_ms = c.bench_kernel("apply", 50)
    top_fronts = [int(t) for L in range(nl, nlev) for t in plan0.level_nodes[L]]
    top_gflop = float(sum(plan0.npiv[t] * float(plan0.nf[t]) ** 2 - float(plan0.npiv[t]) ** 2 * plan0.nf[t] + float(plan0.npiv[t]) ** 3 / 3 for t in top_fronts)) / 1e9
    gather_bytes = (P - 1) * cap * 8 if P > 1 else 0
    gather_ms = (cap * 8 / (LINK_GBS * 1e9) * 1e3 + COLLECTIVE_US * 1e-3) if P > 1 else 0.0          # P - 1 peers over P - 1 links at once
    ncoll = 3 if P > 1 else 0
    iters = 2
    precond = sw["local"] + sw["top"]
    pcg = iters * (precond + apply_ms + 0.15 + ncoll * COLLECTIVE_US * 1e-3) + (sw["local"] / 2 + COLLECTIVE_US * 1e-3 if P > 1 else sw["local"] / 2)
    fwd = max(local_ms) + gather_ms + top_ms + pcg
    serial = top_ms + gather_ms + iters * (sw["top"] + ncoll * COLLECTIVE_US * 1e-3)
    rows.append(dict(P=P, ndof=int(m.ndof), cells_per_rank=int(sub0.nel), replicated_dofs=int(info0["n_top"]), local_levels=int(nl), top_levels=int(nlev - nl),
                     local_assemble_factor_ms_max=max(local_ms), local_assemble_factor_ms_all=local_ms,
                     top_factor_ms=top_ms, top_fronts=len(top_fronts), top_gflop=top_gflop,
                     schur_allgather_bytes_received=int(gather_bytes), schur_packed_doubles=int(cap), schur_allgather_ms_assumed=gather_ms,
                     sweeps_local_ms=sw["local"], sweeps_top_ms=sw["top"], apply_ms=apply_ms,
                     pcg_ms_composed=pcg, forward_ms_composed=fwd, replicated_or_latency_ms=serial))
    eng0.ctx.close()
base = rows[0]["forward_ms_composed"]
for r in rows:
    r["speedup_vs_1"] = base / r["forward_ms_composed"] * (r["P"] if weak else 1)      # weak: throughput relative to one GPU
out = dict(workload=which + (" x P (weak scaling)" if weak else ""), ndof=int(m.ndof), assumptions=dict(collective_us=COLLECTIVE_US, xgmi_link_GBs=LINK_GBS, pcg_iterations=2,
           note="local work of every rank run ALONE on one MI355X; collectives are not executed, their cost is the assumption above"), rows=rows)
print(json.dumps(out, indent=1))
hdr = "| P | cells / rank | replicated DOFs | local assembly + factorisation (max over ranks) | replicated top: factorisation | Schur all-gather received | sweeps local / top | operator | forward composed | speed-up | does not shrink with P |"
print(hdr, file=sys.stderr)
print("|" + "---|" * 11, file=sys.stderr)
for r in rows:
    print(f"| {r['P']} | {r['cells_per_rank']} | {r['replicated_dofs']} | {r['local_assemble_factor_ms_max']:.2f} ms | {r['top_factor_ms']:.2f} ms ({r['top_fronts']} fronts, {r['top_gflop']:.1f} GFLOP) | "
          f"{r['schur_allgather_bytes_received'] / 1e6:.1f} MB ({r['schur_allgather_ms_assumed']:.2f} ms) | {r['sweeps_local_ms']:.2f} / {r['sweeps_top_ms']:.2f} ms | {r['apply_ms'] * 1e3:.0f} us | "
          f"{r['forward_ms_composed']:.2f} ms | {r['speedup_vs_1']:.2f}x | {r['replicated_or_latency_ms']:.2f} ms |", file=sys.stderr)
