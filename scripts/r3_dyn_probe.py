import sys, os, time
import numpy as np
sys.path.insert(0, "/root/repo")
from femo_alpha_amd.mesh import plate_mesh
from femo_alpha_amd.dynamic_rm_shell.plate_sim import PlateSim
import torch
nx, ny, N = 82, 410, 100
mesh = plate_mesh(2.0, 10.0, nx, ny)
T = 2.86; dt = T / N
ps = PlateSim(mesh, 1e8, 0.3, 10.0, dt, N, quad_deg=3)
ps.update_t(np.full(mesh.nn, 0.1))
tt = np.arange(N + 1) * dt
fz = np.where((tt >= 0.02) & (tt <= 0.14), 0.1 * 50 * (1 - np.cos(2 * np.pi * (tt - 0.02) / 0.12)), 0.0)
F = np.zeros((N + 1, mesh.nn, 3)); F[:, :, 2] = fz[:, None]
ps.update_f_history(F.reshape(N + 1, -1))
W0 = None
for rtol, maxit in ((1e-11, 50), (1e-9, 50), (1e-7, 50), (1e-11, 1)):
    ps.ctx.set_solver(preconditioner=2, rtol=rtol, maxit=maxit, check_every=1)
    if maxit == 1: ps.ctx.set_option("strict", 0)
    for rep in range(2):
        ps.update_t(np.full(mesh.nn, 0.1))
        torch.cuda.synchronize(); t0 = time.time()
        W = ps.solve_dynamic_problem()
        torch.cuda.synchronize(); wall = time.time() - t0
    if W0 is None: W0 = W
    print(f"rtol {rtol:g} maxit {maxit}: {N / wall:.0f} steps/s, iters {sorted(set(i for i, r in ps.solve_info))}, relres first 3 {[f'{r:.1e}' for i, r in ps.solve_info[:3]]} last {ps.solve_info[-1][1]:.1e}, history vs rtol 1e-11: {np.abs(W - W0).max() / np.abs(W0).max():.2e}")
sw = np.min([ps.ctx.sweep_profile() for _ in range(3)], axis=0)
print("precond apply ms:", sw.sum(), "apply:", ps.ctx.bench_kernel("apply", 50))
