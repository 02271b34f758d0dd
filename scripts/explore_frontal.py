"""Exploration: multifrontal-preconditioned PCG on the BASELINE configs."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from femo_alpha_amd.mesh import plate_mesh, wing_skin_mesh
from femo_alpha_amd.backend import ShellContext

def run(name, m, fields, marker, leaf=16, rtol=1e-10):
    t = time.time()
    c = ShellContext(m)
    for k, v in fields.items():
        c.set_field(k, v)
    c.set_penalty_facets(m.penalty_facets(marker))
    t1 = time.time()
    plan = c.enable_frontal(leaf)
    print(f"{name}: ndof={m.ndof} nel={m.nel} ctx {t1-t:.2f}s symbolic+upload {time.time()-t1:.2f}s  {plan.summary()}", flush=True)
    c.set_solver(preconditioner=2, rtol=rtol, maxit=30, check_every=1)
    for rep in range(3):
        t = time.time(); info = c.factorize(); dt = time.time() - t
        print(f"   factorize wall {dt*1e3:.1f} ms  {info}", flush=True)
    for rep in range(2):
        c.set_field("thickness", c.get_field("thickness"))      # marks the factorisation stale
        t = time.time(); it, rr = c.solve_state(True); dt = time.time() - t
        print(f"   solve_state (factor + PCG): iters={it} relres={rr:.2e} wall={dt*1e3:.1f} ms {c.last_timing()} -> {m.ndof/dt/1e6:.2f} MDOF/s", flush=True)
    t = time.time(); g, it2, rr2 = c.total_gradient("compliance", "thickness"); dt = time.time() - t
    print(f"   adjoint gradient: iters={it2} relres={rr2:.2e} wall={dt*1e3:.1f} ms", flush=True)
    w = c.get_state()
    print("   max|u| =", np.abs(w[:m.ndof_u]).max(), "compliance", c.functional("compliance"), "|g|max", np.abs(g).max(), flush=True)
    c.close()

which = sys.argv[1] if len(sys.argv) > 1 else "all"
leaf = int(sys.argv[2]) if len(sys.argv) > 2 else 16
if which in ("c1", "all"):
    m = plate_mesh(2.0, 10.0, 10, 50)
    run("config1 plate 10x50", m, dict(thickness=[0.1], E=[1e8], nu=[0.3], density=[10.0],
        F_solid=np.tile([0, 0, 5.0], (m.nn, 1))), lambda x: np.less(x[0], 3e-16), leaf)
if which in ("c2", "all"):
    m = plate_mesh(2.0, 10.0, 58, 290)
    rng = np.random.default_rng(0)
    run("config2 plate 58x290", m, dict(thickness=0.1 * (1 + 0.2 * rng.uniform(-1, 1, m.nn)), E=[1e8], nu=[0.3], density=[10.0],
        F_solid=np.tile([0, 0, 5.0], (m.nn, 1))), lambda x: np.less(x[0], 3e-16), leaf)
if which in ("c3", "all"):
    m = wing_skin_mesh(116, 580)
    run("config3 wing 116x580", m, dict(thickness=[1.27e-3], E=[73.1e9], nu=[0.33], density=[2780.0],
        F_solid=np.tile([0, 0, -2780.0 * 1.27e-3 * 9.81], (m.nn, 1))), lambda x: np.less(x[1], 1e-9), leaf)
