"""BASELINE config 5: 2x10 plate, 82x410 quads (508,734 DOF), 100 midpoint/Newmark steps, 1 GPU -- timed both ways:
the operator factorised once per thickness (this build's default) and re-assembled + re-factorised before every step, as
the reference does (nonlinear_utils.py:210-233) and BASELINE.json configs[4] words it.
    python scripts/bench_dynamic.py [nx ny steps [out.json]]"""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from femo_alpha_amd.mesh import plate_mesh
from femo_alpha_amd.dynamic_rm_shell.plate_sim import PlateSim
nx, ny, N = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (82, 410, 100)
mesh = plate_mesh(2.0, 10.0, nx, ny)
T = 2.86; dt = T / N
t0 = time.time()
ps = PlateSim(mesh, 1e8, 0.3, 10.0, dt, N, quad_deg=3)
ps.update_t(np.full(mesh.nn, 0.1))
tt = np.arange(N + 1) * dt
fz = np.where((tt >= 0.02) & (tt <= 0.14), 0.1 * 50 * (1 - np.cos(2 * np.pi * (tt - 0.02) / 0.12)), 0.0)
F = np.zeros((N + 1, mesh.nn, 3)); F[:, :, 2] = fz[:, None]
ps.update_f_history(F.reshape(N + 1, -1))
print(f"setup {time.time()-t0:.1f}s ndof {mesh.ndof}", flush=True)
import torch
out = {}
for label, every in (("factor_once", False), ("reassemble_every_step", True)):
    for rep in range(2):
        ps.update_t(np.full(mesh.nn, 0.1))
        torch.cuda.synchronize(); t0 = time.time()
        W = ps.solve_dynamic_problem(reassemble_every_step=every)
        torch.cuda.synchronize(); dt_wall = time.time() - t0
    its = [i for i, r in ps.solve_info]
    out[label] = dict(ndof=mesh.ndof, steps=N, wall_s=dt_wall, steps_per_s=N / dt_wall, dof_steps_per_s=mesh.ndof * N / dt_wall,
                      pcg_iters_max=max(its), tip=float(np.abs(W[2:mesh.ndof_u:3, -1]).max()))
    print(label, json.dumps(out[label]), flush=True)
U, T, work = ps.energy_audit()
E = U + T
bal = np.abs(np.diff(E) - work[1:]).max() / max(E.max(), 1e-300)
out["energy_balance_defect"] = float(bal)
print("energy balance defect (max over steps, relative to the largest energy):", bal)
if len(sys.argv) > 4:
    json.dump(out, open(sys.argv[4], "w"), indent=1)
