"""one cheap diagnostic before any more kernels for the gathering rank-k updates of levels 1-5 (0.31 of HBM).

The extend-add is a gather (frontal.h: k_extend_gather, k_trailing_mfma<true>): a lane that owns parent row r reads entry
(cinv[r], cinv[c]) of a child's Schur block.  Along a parent column the reads of consecutive lanes are contiguous in the child
exactly where cinv is a run  cinv[r + 1] = cinv[r] + 1.  This script takes the plan of a workload (host only, no GPU) and prints,
per tree level of the PARENT, the distribution of those run lengths over the rows that receive a child entry.

    python scripts/r6_cinv_runs.py [workload] > profiles/r6_cinv_runs.txt

Decision rule of the verdict: median run < 32 entries -> store each child's Schur block in parent order when its rank-k update
writes it and A/B; runs already long -> write the negative result down and close this line of work."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
WORKLOAD = sys.argv[1] if len(sys.argv) > 1 else "wing1m"
NODE_ORDER = int(sys.argv[2]) if len(sys.argv) > 2 else None      # 0: rows of a front by ascending node id (rounds 1-5); 1: along the separators
sys.argv = [sys.argv[0]]

from bench import make_workload                                  # noqa: E402
from femo_alpha_amd.solver.symbolic import build_plan           # noqa: E402


def main():
    m, fields, marker, desc = make_workload(WORKLOAD)
    p = build_plan(m, m.recommended_leaf_size(), node_order=NODE_ORDER)
    level_of = np.asarray(p.height, dtype=np.int64)
    print(f"node_order {NODE_ORDER}  {WORKLOAD}: {desc}\n{p.ntree} fronts, {p.nlevels} levels; runs of consecutive child rows along the parent's rows (cinv[r+1] = cinv[r] + 1)\n")
    print("parent  fronts   child rows     runs   mean   median   p10   p90   entries in runs >= 16 / >= 32 / >= 64    rows weighted: median run a row sits in")
    for L in range(1, p.nlevels):
        parents = np.nonzero(level_of == L)[0]
        lens = []
        for t in parents:
            for ch in (p.left[t], p.right[t]):
                if ch < 0:
                    continue
                o = int(p.dof_off[ch]); np_, nf_ = int(p.npiv[ch]), int(p.nf[ch])
                up = np.asarray(p.up_map[o + np_: o + nf_], dtype=np.int64)          # child boundary row k -> parent row
                if up.size == 0:
                    continue
                # cinv over the parent's rows: sort the (parent row, child row) pairs by parent row
                order = np.argsort(up, kind="stable")
                pr, cr = up[order], (np_ + order)
                brk = np.nonzero((np.diff(pr) != 1) | (np.diff(cr) != 1))[0]
                edges = np.concatenate([[0], brk + 1, [pr.size]])
                lens.append(np.diff(edges))
        if not lens:
            continue
        l = np.concatenate(lens).astype(np.float64)
        tot = l.sum()
        w = np.repeat(l, l.astype(np.int64))                                      # a row's own run length
        print(f"{L:6d} {parents.size:7d} {int(tot):12d} {l.size:8d} {l.mean():6.1f} {np.median(l):8.0f} {np.percentile(l, 10):5.0f} {np.percentile(l, 90):5.0f}"
              f"      {l[l >= 16].sum() / tot:6.3f} / {l[l >= 32].sum() / tot:6.3f} / {l[l >= 64].sum() / tot:6.3f}"
              f"                      {np.median(w):6.0f}")


if __name__ == "__main__":
    main()
