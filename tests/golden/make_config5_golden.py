"""Generator of the full-size golden for BASELINE config 5: the transient march of bench.py's workload "plate500k_dynamic"
(82 x 410 plate, 508 734 DOF, 100 midpoint / Newmark steps, 1-cosine gust, strong clamp, strain quadrature degree 3).

    python tests/golden/make_config5_golden.py        (about two minutes of the host's cores)

The CPU restatement's march (oracle/cpu_baseline.py::dynamic_march: C++/OpenMP element kernels, LAPACK/BLAS multifrontal Cholesky)
with ONE change: every time step's solve is refined on the residual of the matrix-free step operator  (2/dt^2 M + K/2) w - rhs  until
the correction is below 1e-13 of the state -- dynamic_march itself does one direct solve per step like the reference, which leaves
~1e-7; a golden has to be sharper than the 1e-8 it is compared at.  Stored: the tip deflection at every time level, 4096 seeded
samples of the last state and velocity, the total strain energy  sum_i 1/2 w_i^T K w_i.

Like the other goldens this pins the HIP path to the CPU oracle; the transient forms themselves are restated from the reference's call
sites (DynamicElasticModel is not vendored): parity of config 5 stays unpinned in that sense.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]

from bench import dynamic_case                                   # noqa: E402
from femo_alpha_amd.solver.symbolic import build_plan            # noqa: E402
from oracle import cpu_baseline as cb                            # noqa: E402
from oracle.rm_shell_oracle import ShellOracle                   # noqa: E402


def main():
    t0 = time.time()
    nsteps = 100
    mesh, dt, F = dynamic_case(nsteps=nsteps)
    strong = mesh.locate_dofs_geometrical(lambda x: np.isclose(x[0], 0.0, atol=1e-6))
    o = ShellOracle(mesh, strong_dofs=strong, nred=2)
    o.set_fields(h=0.1, E=1e8, nu=0.3, rho=10.0)
    cs = cb.CpuShell(o)
    cores = cb.host_cores()
    mf = cb.CpuMultifrontal(cs, build_plan(mesh, mesh.recommended_leaf_size()), cores)
    a, b = 2.0 / dt ** 2, 2.0 / dt
    mf.operator = (0.5, a)
    mf.factorize()
    keep = np.ones(mesh.ndof); keep[o.strong_dofs] = 0.0
    tip = int(np.argmax(mesh.nodes[:, 0] + 1e-3 * mesh.nodes[:, 1]))           # the vertex at x = 2, y = 10
    w = np.zeros(mesh.ndof); wd = np.zeros(mesh.ndof)
    tips, energy, worst = [0.0], 0.0, 0.0
    for i in range(1, nsteps + 1):
        o.f = F[i].reshape(-1, 3)
        rhs = cs.load_vector(cores)
        y = cs.apply_op(w, -0.5, a, cores)
        cs.apply_op(wd, 0.0, b, cores, out=y)
        rhs += keep * y
        rhs[o.strong_dofs] = 0.0
        x = mf.solve(rhs)
        for _ in range(6):                                   # refinement on the step operator's own residual
            r = rhs - keep * cs.apply_op(x, 0.5, a, cores)
            r[o.strong_dofs] = -x[o.strong_dofs]
            dx = mf.solve(r)
            x += dx
            corr = np.abs(dx).max() / max(np.abs(x).max(), 1e-300)
            if corr < 1e-13:
                break
        worst = max(worst, corr)
        wd = b * (x - w) - wd
        w = x
        tips.append(float(w[3 * tip + 2]))
        energy += 0.5 * float(w @ cs.apply_op(w, 1.0, 0.0, cores))
        if i % 20 == 0:
            print(f"  level {i}: tip {tips[-1]:.12e}  last correction {corr:.1e}  ({time.time() - t0:.0f} s)", flush=True)
    sample = np.sort(np.random.default_rng(7).choice(mesh.ndof, size=4096, replace=False))
    print(f"ndof {mesh.ndof}, {nsteps} steps: tip deflection {tips[-1]:.15e}, total strain energy {energy:.15e}, worst last correction {worst:.1e}, "
          f"{time.time() - t0:.0f} s")
    np.savez_compressed(os.path.join(os.environ.get("FEMO_GOLDEN_OUT", HERE), "config5_plate500k_dynamic.npz"), ndof=mesh.ndof, nsteps=nsteps, dt=dt,
                        tip_vertex=tip, tip_history=np.array(tips), total_strain_energy=energy, sample_index=sample, w_last_sample=w[sample],
                        wdot_last_sample=wd[sample], w_last_maxabs=np.abs(w).max(), wdot_last_maxabs=np.abs(wd).max(), worst_correction=worst)


if __name__ == "__main__":
    main()
