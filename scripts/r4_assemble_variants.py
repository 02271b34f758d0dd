"""Where does the front assembly of the transient step operator spend its time?  assemble_ms of femo_factorize on the config-5 plate for
the static operator, the reduced strain rule, the inertia term, and both."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from femo_alpha_amd.backend import ShellContext
from femo_alpha_amd.mesh import plate_mesh
m = plate_mesh(2.0, 10.0, 82, 410)
c = ShellContext(m, nquad=4)
for k, v in dict(thickness=[0.1], E=[1e8], nu=[0.3], density=[10.0]).items():
    c.set_field(k, v)
c.set_strong_dofs(m.locate_dofs_geometrical(lambda x: np.less(x[0], 3e-16)))
c.enable_frontal()
c.set_solver(preconditioner=2, rtol=1e-8, maxit=30, check_every=1)
for name, nred, aK, aM in (("static, 4 x 4", 0, 1.0, 0.0), ("static, strain rule 2 x 2 + 4 x 4", 2, 1.0, 0.0), ("K/2 + 2/dt^2 M, 4 x 4", 0, 0.5, 2 / 0.0286 ** 2),
                           ("K/2 + 2/dt^2 M, strain rule 2 x 2 + 4 x 4", 2, 0.5, 2 / 0.0286 ** 2)):
    c.set_strain_quadrature(nred)
    c.set_operator(aK, aM)
    c.factorize()
    t = [c.factorize()["assemble_ms"] for _ in range(5)]
    print(f"{name:45s} assemble {np.median(t):.3f} ms", flush=True)
