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


@pytest.mark.parametrize("world,kind", [(1, "wing"), (2, "wing"), (4, "wing"), (8, "wing"), (2, "plate"), (2, "wing_cr"), (4, "wing_cr"), (2, "wing_tri_nu")])
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
    # the rule of the whole mesh: triangles with a nodal Poisson ratio that varies take UFL's degree 9 on every rank
    assert int(r["nquad"]) == {"wing_tri_nu": 9, "wing_cr": 6}.get(kind, m.recommended_nquad())


@pytest.mark.parametrize("kind", ["wing", "wing_cr"])
def test_rank_plans_partition_the_mesh(kind):
    from femo_alpha_amd.solver.symbolic import analyse, rank_plan
    m, _, _ = H.make_case(kind)
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


def test_bench_launcher_with_two_ranks_on_the_cpu():
    """``python bench.py --gpus 2`` without a launcher and without a GPU: bench.py's own ``launch_ranks`` starts two ranks
    (torch.distributed.run --standalone), the ranks rendezvous over gloo, run the partitioned forward + adjoint step of
    ``main_distributed`` on a small skin with the numpy stand-in engine (``--rehearsal-engine``: a test hook, the line says
    REHEARSAL) and rank 0 prints the one JSON line of the contract -- with the element-partition label, and with the OTHER scaling
    as a sub-object (``weak`` beside the default strong run)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FEMO_BENCH_NC="8", FEMO_BENCH_NS="24", OMP_NUM_THREADS="2",
               PYTHONPATH=os.pathsep.join([os.path.join(root, "tests"), root, os.environ.get("PYTHONPATH", "")]))
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--rehearsal-engine", "dist_helpers:NumpyEngine"], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["unit"] == "DOF/s" and j["value"] > 0 and j["steps"] == 1
    assert "REHEARSAL" in j["data"] and "roofline" not in j
    assert j["config"]["parallelism"].startswith("element partition over 2 GPUs")
    assert j["config"]["pcg_iterations_forward"] <= 4 and j["config"]["pcg_iterations_adjoint"] <= 4
    assert j["config"]["replicated_separator_dofs"] > 0
    w = j["weak"]
    assert w["scaling"] == "weak" and w["ndof"] > j["config"]["ndof"] and w["ndof_per_gpu"] >= 0.8 * j["config"]["ndof_per_gpu"]
    assert w["pcg_iterations_forward"] <= 4
