"""Per-kernel-class timing of one multifrontal factorisation (HIP events around every launch)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from femo_alpha_amd.mesh import plate_mesh, wing_skin_mesh
from femo_alpha_amd.backend import ShellContext
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
leaf = int(sys.argv[2]) if len(sys.argv) > 2 else 16
if which == "c3":
    m = wing_skin_mesh(116, 580)
    fields = dict(thickness=[1.27e-3], E=[73.1e9], nu=[0.33], density=[2780.0], F_solid=np.tile([0, 0, -34.6], (m.nn, 1)))
    marker = lambda x: np.less(x[1], 1e-9)
else:
    m = plate_mesh(2.0, 10.0, 58, 290)
    fields = dict(thickness=[0.1], E=[1e8], nu=[0.3], density=[10.0], F_solid=np.tile([0, 0, 5.0], (m.nn, 1)))
    marker = lambda x: np.less(x[0], 3e-16)
c = ShellContext(m)
for k, v in fields.items():
    c.set_field(k, v)
c.set_penalty_facets(m.penalty_facets(marker))
c.enable_frontal(leaf)
c.set_solver(preconditioner=2, rtol=1e-10, maxit=30, check_every=1)
c.factorize()
p = c.factorize_profile()
tot = sum(v["ms"] for v in p.values() if isinstance(v, dict))
for k, v in p.items():
    if isinstance(v, dict):
        print(f"{k:16s} {v['ms']:8.2f} ms  {v['launches']:5d} launches  avg {v['ms']/max(v['launches'],1)*1e3:8.1f} us")
print("sum", tot, "trailing GFLOP", p["trailing_flops"] / 1e9)
print(c.factorize())
