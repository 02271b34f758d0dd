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


@pytest.mark.parametrize("world", [1, 2, 4])
def test_hip_partitioned_solve_matches_oracle(world):
    m, marker, fields = H.make_case("wing")
    w0, J0, dJ0, M0 = H.reference_solution(m, marker, fields)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "res.npz")
        mp.spawn(H.worker, args=(world, _free_port(), "wing", "hip", path), nprocs=world, join=True)
        r = np.load(path)
    assert int(r["it"]) <= 4 and float(r["rel"]) < 1e-12
    assert np.abs(r["w"] - w0).max() < 1e-8 * np.abs(w0).max()
    assert abs(float(r["J"]) - J0) < 1e-8 * abs(J0)
    assert abs(float(r["M"]) - M0) < 1e-12 * M0
    assert np.abs(r["g"] - dJ0).max() < 1e-7 * np.abs(dJ0).max()
