"""world_size 2, 4 and 8 under gloo on the CPU: the element-partitioned driver (femo_alpha_amd/parallel.py)
with the numpy stand-in engine reproduces the single-domain oracle solution, compliance, mass and
d compliance / d thickness.  Covers the N > 1 path of bench.py by construction."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch.multiprocessing as mp

import dist_helpers as H


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,kind", [(1, "wing"), (2, "wing"), (4, "wing"), (8, "wing"), (2, "plate")])
def test_partitioned_solve_matches_single_domain(world, kind):
    m, marker, fields = H.make_case(kind)
    w0, J0, dJ0, M0 = H.reference_solution(m, marker, fields)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "res.npz")
        mp.spawn(H.worker, args=(world, _free_port(), kind, "numpy", path), nprocs=world, join=True)
        r = np.load(path)
    assert int(r["it"]) <= 4 and float(r["rel"]) < 1e-12
    assert np.abs(r["w"] - w0).max() < 1e-8 * np.abs(w0).max()
    assert abs(float(r["J"]) - J0) < 1e-7 * abs(J0)        # the oracle LU of the 1e15-penalised thin plate is good to ~1e-8
    assert abs(float(r["M"]) - M0) < 1e-12 * M0
    assert np.abs(r["g"] - dJ0).max() < 1e-7 * np.abs(dJ0).max()
    if world > 1:
        assert int(r["ntop"]) > 0


def test_rank_plans_partition_the_mesh():
    from femo_alpha_amd.solver.symbolic import analyse, rank_plan
    m, _, _ = H.make_case("wing")
    T = analyse(m, 4, min_depth=2)
    seen = np.zeros(m.nel, int)
    owned = np.zeros(m.ndof, int)
    for r in range(4):
        sub, plan, info = rank_plan(m, T, r, 4)
        seen[info["cells"]] += 1
        nvec = sub.ndof + info["nghost"]
        piv = np.concatenate([plan.front_dofs[plan.dof_off[t]:plan.dof_off[t] + plan.npiv[t]] for t in range(plan.ntree)])
        assert np.array_equal(np.sort(piv), np.arange(nvec))           # every local entry eliminated exactly once
        is_top = np.zeros(nvec, bool); is_top[info["top_local"]] = True
        owned[info["l2g_dof"][~is_top]] += 1                             # interior DOFs belong to one rank only
        assert len(info["top_local"]) == info["n_top"]
        # the replicated entries come in the same global order on every rank
        assert np.all(np.diff(info["l2g_dof"][info["top_local"]]) > 0)
    assert np.all(seen == 1)
    assert owned.max() == 1
