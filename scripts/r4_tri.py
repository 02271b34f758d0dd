"""Robustness / speed datapoint: the config-3 skin cut into triangles (134 560 P2-P1 triangles, ~0.95 M DOF), forward solve and
adjoint gradient through the multifrontal Cholesky + PCG, true residual by the operator."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from femo_alpha_amd.backend import ShellContext
from femo_alpha_amd.mesh import quads_to_triangles, wing_skin_mesh

m = quads_to_triangles(wing_skin_mesh(116, 580)).renumbered()[0]
c = ShellContext(m)
for k, v in dict(thickness=[1.27e-3], E=[73.1e9], nu=[0.33], density=[2780.0]).items():
    c.set_field(k, v)
c.set_field("F_solid", np.tile([0.0, 0.0, -2780.0 * 1.27e-3 * 9.81], (m.nn, 1)))
c.set_penalty_facets(m.penalty_facets(lambda x: np.less(x[1], 1e-9)))
leaf = int(sys.argv[1]) if len(sys.argv) > 1 else 12
t0 = time.perf_counter(); plan = c.enable_frontal(leaf); t1 = time.perf_counter() - t0
c.set_solver(preconditioner=2, rtol=1e-10, maxit=50, check_every=1)
print(f"leaf {leaf}: {m.nel} triangles, {m.ndof} DOF; plan {plan.summary()}  ({t1:.2f} s)", flush=True)
h = c.get_field("thickness")
tf, ta = [], []
for _ in range(6):
    c.set_field("thickness", h)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    it, rr = c.solve_state(zero_guess=True)
    t1 = time.perf_counter()
    g, it2, rr2 = c.total_gradient("compliance", "thickness")
    t2 = time.perf_counter()
    tf.append(t1 - t0); ta.append(t2 - t1)
w = c.get_state()
r = c.load_vector() - c.apply_K(w)
print(f"forward {np.median(tf[1:]) * 1e3:.2f} ms = {m.ndof / np.median(tf[1:]) / 1e6:.1f} M DOF/s ({it} iterations, relres {rr:.1e}), "
      f"adjoint {np.median(ta[1:]) * 1e3:.2f} ms ({it2} iterations); true residual {np.linalg.norm(r) / np.linalg.norm(c.load_vector()):.1e}; "
      f"info {c.frontal_info()}", flush=True)
print("compliance", c.functional("compliance"), " |gradient| max", np.abs(g).max())
