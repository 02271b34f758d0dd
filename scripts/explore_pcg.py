"""Exploration: Jacobi-PCG iteration counts and kernel times on the BASELINE configs."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from femo_alpha_amd.mesh import plate_mesh, wing_skin_mesh
from femo_alpha_amd.backend import ShellContext

def run(name, m, fields, marker, maxit, rtol=1e-8):
    t = time.time()
    c = ShellContext(m)
    for k, v in fields.items():
        c.set_field(k, v)
    c.set_penalty_facets(m.penalty_facets(marker))
    print(f"{name}: ndof={m.ndof} nel={m.nel} setup {time.time()-t:.2f}s", flush=True)
    for k in ("apply", "pcg_update", "pcg_direction", "diag"):
        print(f"   kernel {k}: {c.bench_kernel(k, 50)*1e3:.1f} us", flush=True)
    c.set_solver(rtol=rtol, maxit=maxit, check_every=500)
    t = time.time()
    it, rr = c.solve_state(True)
    dt = time.time() - t
    print(f"   PCG: iters={it} relres={rr:.3e} wall={dt:.2f}s  {c.last_timing()}  per-iter {dt/max(it,1)*1e6:.1f} us", flush=True)
    w = c.get_state()
    print("   max|u| =", np.abs(w[:m.ndof_u]).max(), "compliance", c.functional("compliance"), flush=True)
    c.close()

which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("c1", "all"):
    m = plate_mesh(2.0, 10.0, 10, 50)
    run("config1 plate 10x50", m, dict(thickness=[0.1], E=[1e8], nu=[0.3], density=[10.0],
        F_solid=np.tile([0, 0, 5.0], (m.nn, 1))), lambda x: np.less(x[0], 3e-16), 200000)
if which in ("c2", "all"):
    m = plate_mesh(2.0, 10.0, 58, 290)
    rng = np.random.default_rng(0)
    run("config2 plate 58x290", m, dict(thickness=0.1 * (1 + 0.2 * rng.uniform(-1, 1, m.nn)), E=[1e8], nu=[0.3], density=[10.0],
        F_solid=np.tile([0, 0, 5.0], (m.nn, 1))), lambda x: np.less(x[0], 3e-16), int(sys.argv[2]) if len(sys.argv) > 2 else 100000)
if which in ("c3", "all"):
    m = wing_skin_mesh(116, 580)
    run("config3 wing 116x580", m, dict(thickness=[1.27e-3], E=[73.1e9], nu=[0.33], density=[2780.0],
        F_solid=np.tile([0, 0, -2780.0 * 1.27e-3 * 9.81], (m.nn, 1))), lambda x: np.less(x[1], 1e-9), int(sys.argv[2]) if len(sys.argv) > 2 else 20000)
