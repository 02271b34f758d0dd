"""Device-side unit checks of two building blocks that the solver-level parity tests only see through many layers:
``wave_sum_cols<N>`` (column sums of a wave in one butterfly: permlane swaps + DPP) against exact integer sums, and the staged
element-matrix column loop (``stage_qpoints``: lane q computes quadrature point q once, LDS) against the plain loop on a warped
quadrilateral.  The programs live in scripts/micro/ and are compiled here with hipcc for gfx950."""
import os
import shutil
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MICRO = os.path.join(ROOT, "scripts", "micro")
CSRC = os.path.join(ROOT, "femo_alpha_amd", "csrc")


def _build_and_run(name, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = os.path.join(str(tmp_path), name)
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-I" + CSRC, os.path.join(MICRO, name + ".hip"), "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    sys.stdout.write(r.stdout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_wave_sum_cols_on_the_device(tmp_path):
    out = _build_and_run("wsum_micro", tmp_path)
    assert "PASSED" in out and "FAIL" not in out.replace("FAILED", "")


def test_staged_quadrature_points_match_the_plain_column_loop(tmp_path):
    out = _build_and_run("stage_qpoints_micro", tmp_path)
    assert "max err" in out
