"""Host logic of bench.py that needs no GPU: the roofline bookkeeping of the rank-k updates (two classes by the roof that binds a
launch), the gate on the committed counter passes (HBM traffic is reported only for the kernel sources it was measured on), the
workload table."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _prof(mfma_ms, hbm_ms):
    mf = {"launches": 40, "ms": mfma_ms, "flops": 150e9, "bytes": 5.8e9}
    hb = {"launches": 18, "ms": hbm_ms, "flops": 36e9, "bytes": 6.5e9}
    return {"trailing": {"ms": mfma_ms + hbm_ms, "launches": 58}, "trailing_flops": 186e9, "trailing_bytes": 12.3e9,
            "trailing_mfma_bound": mf, "trailing_hbm_bound": hb}


def test_roofline_is_all_launches_with_the_two_classes_inside():
    allk, main, other = bench.trailing_roofline(_prof(4.0, 2.8), 3.5e8, {"mfma": 3.4e8, "hbm": 3.8e8})
    # the top-level object: every rank-k launch against the MFMA peak (the same definition every round)
    assert allk["bound"] == "mfma" and allk["unit"] == "TFLOP/s" and allk["peak"] == bench.FP64_PEAK_TFLOPS
    assert allk["frac"] == pytest.approx(186e9 / 6.8e-3 / 1e12 / 78.6) and allk["traffic"] == 3.5e8
    assert allk["launches_per_factorisation"] == 58 and allk["algorithmic_flops_per_launch"] == pytest.approx(186e9 / 58)
    assert set(allk["by_binding_roof"]) == {"mfma", "hbm"} and allk["by_binding_roof"]["mfma"] is main
    # the class that takes more time, then the other one
    assert main["bound"] == "mfma" and main["unit"] == "TFLOP/s" and main["peak"] == bench.FP64_PEAK_TFLOPS
    assert main["achieved"] == pytest.approx(150e9 / 4.0e-3 / 1e12) and main["frac"] == pytest.approx(main["achieved"] / 78.6)
    assert main["traffic"] == 3.4e8 and main["algorithmic_bytes_per_launch"] == pytest.approx(5.8e9 / 40)
    assert other["bound"] == "hbm" and other["unit"] == "GB/s" and other["peak"] == bench.HBM_PEAK_GBS and other["traffic"] == 3.8e8
    assert other["achieved"] == pytest.approx(6.5e9 / 2.8e-3 / 1e9)
    _, main2, other2 = bench.trailing_roofline(_prof(1.0, 2.8), None)          # a mesh of small fronts: HBM is the roof that binds
    assert main2["bound"] == "hbm" and other2["bound"] == "mfma" and main2["traffic"] is None
    # the ridge the launches are classified by (femo_hip.hip, profile class 7) is the ratio of the two peaks the line quotes
    assert bench.FP64_PEAK_TFLOPS * 1e12 / (bench.HBM_PEAK_GBS * 1e9) == pytest.approx(9.8, abs=0.05)


def test_counter_traffic_is_reported_only_for_the_sources_it_was_measured_on(tmp_path, monkeypatch):
    from femo_alpha_amd import _build
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.pmc_traffic("wing1m") == (None, None, {})                      # no file
    rec = {"source_digest": "0" * 64, "apply_hbm_bytes_per_launch": 1.0, "trailing_hbm_bytes_per_launch": 2.0,
           "trailing_mfma_bound_hbm_bytes_per_launch": 3.0, "trailing_hbm_bound_hbm_bytes_per_launch": 4.0}
    (prof / "pmc_wing1m.json").write_text(json.dumps(rec))
    assert bench.pmc_traffic("wing1m") == (None, None, {})                      # stale: other kernel sources
    rec["source_digest"] = _build.source_digest()
    (prof / "pmc_wing1m.json").write_text(json.dumps(rec))
    assert bench.pmc_traffic("wing1m") == (1.0, 2.0, {"mfma": 3.0, "hbm": 4.0})


def test_committed_counter_passes_belong_to_the_committed_sources():
    """profiles/pmc_wing1m.json must carry the digest of csrc/ as committed, or the driver's bench line has roofline.traffic = null."""
    from femo_alpha_amd import _build
    rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_wing1m.json")))
    assert rec["source_digest"] == _build.source_digest(), "kernel sources changed after the counter passes: re-run scripts/r4_rocprof.sh"


def test_workload_table():
    m, fields, marker, desc = bench.make_workload("plate8k")
    assert m.ndof == 8046 and "8046 DOF" in desc
    m, fields, marker, desc = bench.make_workload("plate250k", renumber=False)
    assert m.ndof == 255438 and fields["thickness"].shape == (m.nn,)
    m, fields, marker, desc = bench.make_workload("uquad1m", renumber=False)
    assert m.is_quad and m.ndof == 1016124 and m.recommended_nquad() == 6      # kites: strongly non-affine
    with pytest.raises(SystemExit):
        bench.make_workload("wing0m")
    with pytest.raises(SystemExit):
        bench.make_workload("nothing")
