"""The element-partitioned driver with the real HIP engine: world_size 1 and 2 sharing the box's one
GPU, collectives over gloo (host-staged) because RCCL wants one device per rank.  On the 8-GPU node the
same driver runs with backend nccl (= RCCL over xGMI) from bench.py."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch.multiprocessing as mp

import dist_helpers as H

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,kind", [(1, "wing"), (2, "wing"), (4, "wing"), (2, "wing_cr"), (4, "wing_cr")])
def test_hip_partitioned_solve_matches_oracle(world, kind):
    """``wing_cr``: ShellElement 'CG2CR1' (linear_shell_model.py:68-73) in element partitions -- the rotation DOFs of a replicated
    separator belong to its edge midpoints (round 6; rounds 1-5 refused this element here)."""
    m, marker, fields = H.make_case(kind)
    w0, J0, dJ0, M0 = H.reference_solution(m, marker, fields)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "res.npz")
        mp.spawn(H.worker, args=(world, _free_port(), kind, "hip", path), nprocs=world, join=True)
        r = np.load(path)
    assert int(r["it"]) <= 4 and float(r["rel"]) < 1e-12
    assert np.abs(r["w"] - w0).max() < 1e-8 * np.abs(w0).max()
    assert abs(float(r["J"]) - J0) < 1e-8 * abs(J0)
    assert abs(float(r["M"]) - M0) < 1e-12 * M0
    assert np.abs(r["g"] - dJ0).max() < 1e-7 * np.abs(dJ0).max()


def test_single_rank_over_rccl_matches_the_gloo_run():
    """Backend nccl (= RCCL), world 1 (RCCL wants a device per rank; the box has one): every collective of the driver goes through
    RCCL on the DEVICE tensors -- the code path of bench.py --gpus N on a multi-GPU node, which the gloo tests (host-staged copies)
    do not take -- and must be ordered with the library's own stream.  Same result as the gloo run to rounding."""
    m, marker, fields = H.make_case("wing")
    w0, J0, dJ0, M0 = H.reference_solution(m, marker, fields)
    res = {}
    for backend in ("gloo", "nccl"):
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "res.npz")
            mp.spawn(H.worker, args=(1, _free_port(), "wing", "hip", path, backend), nprocs=1, join=True)
            res[backend] = {k: v for k, v in np.load(path).items()}
    r, g = res["nccl"], res["gloo"]
    assert int(r["it"]) == int(g["it"]) and int(r["it2"]) == int(g["it2"])
    assert np.abs(r["w"] - g["w"]).max() < 1e-12 * np.abs(g["w"]).max()
    assert np.abs(r["g"] - g["g"]).max() < 1e-12 * np.abs(g["g"]).max()
    assert np.abs(r["w"] - w0).max() < 1e-8 * np.abs(w0).max()
    assert np.abs(r["g"] - dJ0).max() < 1e-7 * np.abs(dJ0).max()


def test_partitions_agree_with_the_single_rank_run():
    """SURVEY.md section 8e "correctness check": the P-rank result against the 1-rank result of the SAME driver --
    displacement <= 1e-12 relative, identical PCG iteration counts (measured: 4e-15 on displacement, 4e-16 on the
    compliance, 8e-16 on the gradient for 2 and 4 ranks).  Also exercises the packed lower-triangle exchange of the
    subtree Schur complements."""
    res = {}
    for world in (1, 2, 4):
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "res.npz")
            mp.spawn(H.worker, args=(world, _free_port(), "wing", "hip", path), nprocs=world, join=True)
            res[world] = {k: v for k, v in np.load(path).items()}
    one = res[1]
    for world in (2, 4):
        r = res[world]
        ew = np.abs(r["w"] - one["w"]).max() / np.abs(one["w"]).max()
        eJ = abs(float(r["J"]) - float(one["J"])) / abs(float(one["J"]))
        eg = np.abs(r["g"] - one["g"]).max() / np.abs(one["g"]).max()
        print(f"world {world} vs 1: iterations {int(r['it'])}/{int(r['it2'])} vs {int(one['it'])}/{int(one['it2'])}, "
              f"displacement {ew:.2e}, compliance {eJ:.2e}, gradient {eg:.2e}")
        assert int(r["it"]) == int(one["it"]) and int(r["it2"]) == int(one["it2"])
        assert ew < 1e-12 and eJ < 1e-12 and eg < 1e-12
        assert int(r["ntop"]) > 0


# ---------------------------------------------------------------------------------------------- BASELINE config 4 at full size
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config3_wing1m.npz")
PARTITION_TOL = 1e-12


def _check_against_config3_golden(r, tol=1e-8):
    """The config-3 tolerance (tests/test_gpu_goldens.py: 1e-8 against the exact discrete solution)."""
    g = np.load(GOLDEN)
    w = r["w"]
    assert w.size == int(g["ndof"])
    assert abs(np.abs(w).max() - float(g["w_maxabs"])) < tol * float(g["w_maxabs"])
    assert np.abs(w[g["w_sample_index"]] - g["w_sample"]).max() < tol * float(g["w_maxabs"])
    assert abs(float(r["J"]) - float(g["compliance"])) < tol * abs(float(g["compliance"]))
    assert abs(float(r["M"]) - float(g["mass"])) < 1e-12 * float(g["mass"])
    ref = g["dcompliance_dthickness"]
    assert np.abs(r["g"] - ref).max() < tol * np.abs(ref).max()


def test_config4_one_million_dof_skin_in_2_4_8_partitions():
    """BASELINE config 4 as written, minus the hardware: the 1 015 470-DOF wing skin of the bench in 2, 4 and 8 element
    partitions on the HIP engine, all partitions sharing the box's one GPU (threads: a box admits 6 GPU processes, config 4
    needs 8 ranks).  Every partitioning against the committed config-3 golden at the config-3 tolerance, and against the
    1-partition run of the same driver: PCG iteration counts within one, displacement / compliance / gradient to 1e-12
    (measured at 2 partitions: 6e-14 / 1e-16 / 5e-14 -- PCG on the true residual to 1e-12 pins the
    solution although one unit in the last place of the operator moves it by 7e-8, scripts/conditioning_floor.py)."""
    m, marker, fields = H.make_case("wing1m")
    res = {world: H.run_threads(world, m, marker, fields) for world in (1, 2, 4, 8)}
    one = res[1]
    assert int(one["it"]) <= 4 and int(one["it2"]) <= 4
    for world in (1, 2, 4, 8):
        _check_against_config3_golden(res[world])
    for world in (2, 4, 8):
        r = res[world]
        ew = np.abs(r["w"] - one["w"]).max() / np.abs(one["w"]).max()
        eJ = abs(float(r["J"]) - float(one["J"])) / abs(float(one["J"]))
        eg = np.abs(r["g"] - one["g"]).max() / np.abs(one["g"]).max()
        print(f"1M DOF, {world} partitions vs 1: iterations {int(r['it'])}/{int(r['it2'])} vs {int(one['it'])}/{int(one['it2'])}, "
              f"displacement {ew:.2e}, compliance {eJ:.2e}, gradient {eg:.2e}, replicated DOFs {int(r['ntop'])}")
        # "identical iteration counts +-1" (SURVEY.md section 8e): at rtol 1e-12 the last iteration of the adjoint solve sits on
        # the threshold (measured: 2 partitions take 3 where one takes 2)
        assert abs(int(r["it"]) - int(one["it"])) <= 1 and abs(int(r["it2"]) - int(one["it2"])) <= 1
        assert ew < PARTITION_TOL and eJ < PARTITION_TOL and eg < PARTITION_TOL
        assert int(r["ntop"]) > 0
    # the reason given for the +-1 above, tested: it is the threshold, not the partitioning.  One
    # application of the factor contracts the residual by ~1e-5, so at rtol 1e-12 the second iterate lands within a decade of the
    # threshold and rounding decides whether a third is taken; at rtol 1e-10 it clears it by two decades -- there every partitioning
    # must take exactly the iterations of the one-partition run
    loose = {world: H.run_threads(world, m, marker, fields, rtol=1e-10) for world in (1, 2, 8)}
    for world in (2, 8):
        assert int(loose[world]["it"]) == int(loose[1]["it"]) and int(loose[world]["it2"]) == int(loose[1]["it2"]), \
            (world, int(loose[world]["it"]), int(loose[world]["it2"]), int(loose[1]["it"]), int(loose[1]["it2"]))


def test_config4_two_processes_over_gloo_at_full_size():
    """The same 1 M-DOF skin in two partitions as two PROCESSES with torch.distributed (gloo, host-staged): the code path
    bench.py --gpus 2 takes, but for the collective backend."""
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "res.npz")
        mp.spawn(H.worker, args=(2, _free_port(), "wing1m", "hip", path), nprocs=2, join=True)
        r = {k: v for k, v in np.load(path).items()}
    assert int(r["it"]) <= 4 and int(r["it2"]) <= 4
    _check_against_config3_golden(r)


def test_bench_distributed_path_over_rccl_with_one_rank():
    """bench.py's N > 1 code path (process group over nccl = RCCL, the partitioned driver, max-over-ranks timing, the JSON line with both
    roofline objects) with ONE rank -- FEMO_BENCH_FORCE_DIST=1 -- on a shortened wing skin: what a one-GPU box can rehearse of the
    multi-GPU bench."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FEMO_BENCH_FORCE_DIST="1", FEMO_BENCH_NS="60", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["unit"] == "DOF/s"
    assert line["config"]["pcg_iterations_forward"] <= 4 and line["config"]["relres_forward"] < 1e-9
    # the rank-k updates of rank 0's factorisation: all launches against the MFMA peak (the same definition as the N = 1 line), the two
    # classes by binding roof inside
    rf = line["roofline"]
    assert (rf["bound"], rf["unit"]) == ("mfma", "TFLOP/s") and 0 < rf["frac"] < 1 and line["roofline_spmv"]["bound"] == "hbm"
    assert set(rf["by_binding_roof"]) <= {"mfma", "hbm"} and rf["by_binding_roof"]
    assert line["config"]["parallelism"].startswith("element partition over 1 GPUs")
    # the other scaling rides along (with one rank: the same skin again)
    w = line["weak"]
    assert w["scaling"] == "weak" and w["ndof"] == line["config"]["ndof"] and w["value"] > 0 and w["pcg_iterations_forward"] <= 4


def test_bench_with_four_ranks_at_full_size_on_one_card():
    """``python bench.py --gpus 4 --share-gpu`` at the FULL size of BASELINE config 4 (FEMO_BENCH_NS = 580: the 1 015 470-DOF skin in
    four element partitions): bench.py starts its own four ranks (torch.distributed.run --standalone), they share the one card of the
    box (a box admits six GPU processes; collectives over gloo), and the JSON line must carry what the single-GPU solve of the same
    skin gives -- 2 PCG iterations forward and adjoint at rtol 1e-10 (the config-3 golden test) -- both roofline objects, and the WEAK
    figure beside the strong one (four times the span, a 1 M-DOF partition per rank: 4.1 M DOF through the same four processes).  What is
    rehearsed: the launcher, the rendezvous, the partitioned driver of four processes with the HIP engine, the max-over-ranks timing.
    Not measured: anything about scaling (one card)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FEMO_BENCH_NS="580")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "FEMO_BENCH_FORCE_DIST"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--share-gpu", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 4 and line["scaling"] == "strong" and line["unit"] == "DOF/s" and line["value"] > 0
    cfg = line["config"]
    assert cfg["ndof"] == 1015470 and cfg["gauss_points_per_direction"] == 5
    # the single-GPU solve of this skin takes 2 + 2 iterations at rtol 1e-10; a partitioned run may sit on the threshold (+1)
    assert 2 <= cfg["pcg_iterations_forward"] <= 3 and 2 <= cfg["pcg_iterations_adjoint"] <= 3
    assert cfg["relres_forward"] < 1e-10 and cfg["relres_adjoint"] < 1e-10
    # the rank-k updates of rank 0's factorisation: all launches against the MFMA peak (the same definition as the N = 1 line), the two
    # classes by binding roof inside
    rf = line["roofline"]
    assert (rf["bound"], rf["unit"]) == ("mfma", "TFLOP/s") and 0 < rf["frac"] < 1 and line["roofline_spmv"]["bound"] == "hbm"
    assert set(rf["by_binding_roof"]) <= {"mfma", "hbm"} and rf["by_binding_roof"]
    assert cfg["parallelism"].startswith("element partition over 4 GPUs")
    # the other scaling rides along: four times the span, one 1 M-DOF partition per rank
    w = line["weak"]
    assert w["scaling"] == "weak" and w["ndof"] > 4_000_000 and 0.9e6 < w["ndof_per_gpu"] < 1.2e6 and w["value"] > 0
    assert w["pcg_iterations_forward"] <= 4 and w["pcg_iterations_adjoint"] <= 4
