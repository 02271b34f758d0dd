"""Target of scripts/r3_timeline.sh: three factorisations of one mesh under the given schedule options.
    python3 scripts/r3_timeline_target.py wing1m [key=value ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_workload
from femo_alpha_amd.backend import ShellContext

which = sys.argv[1] if len(sys.argv) > 1 else "wing1m"
opts = {a.split("=")[0]: float(a.split("=")[1]) for a in sys.argv[2:]}
m, fields, marker, desc = make_workload(which)
c = ShellContext(m)
for k, v in fields.items():
    c.set_field(k, v)
c.set_penalty_facets(m.penalty_facets(marker))
c.enable_frontal(12)
for k, v in opts.items():
    c.set_option(k, v)
c.set_solver(preconditioner=2, rtol=1e-10, maxit=30, check_every=1)
for _ in range(3):
    c.factorize()
c.sync()
