"""How well is the discrete solution of the 1M-DOF wing defined in double precision?  The CPU oracle's stiffness matrix is
perturbed by one unit in the last place on a random (symmetric) two thirds of its entries -- the kind of difference two
correct fp64 evaluations of the same operator have -- and the system is solved again (multifrontal Cholesky of the original
matrix as the preconditioner of an extended-precision refinement on the perturbed one).  The change of the displacement,
compliance and gradient is the floor below which two implementations cannot be expected to agree."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.argv = [sys.argv[0]]
from bench import make_workload
from femo_alpha_amd.solver.symbolic import build_plan
from oracle import cpu_baseline as cb
from oracle.rm_shell_oracle import ShellOracle
from _extended import operator_from_float64, refine                    # noqa: E402  (a float64 matrix behind the refinement's interface)

m, fields, marker, desc = make_workload("wing1m")
cores = cb.host_cores()
o = ShellOracle(m, penalty_facets=m.penalty_facets(marker))
o.set_fields(h=fields["thickness"], E=fields["E"], nu=fields["nu"], rho=fields["density"], f=fields["F_solid"])
cs = cb.CpuShell(o); cs.pattern()
K = cs.assemble_K(cores).tocsr(); K.sort_indices()
b = cs.load_vector(cores)
mf = cb.CpuMultifrontal(cs, build_plan(m, 12), cores); mf.factorize()
w, _ = refine(operator_from_float64(K, cs, cores), mf.solve, b, mf.solve(b), steps=6)
rows = np.repeat(np.arange(K.shape[0], dtype=np.int64), np.diff(K.indptr))
lo, hi = np.minimum(rows, K.indices), np.maximum(rows, K.indices)
key = (lo * 1000003 + hi) * 2654435761 % (2 ** 31)
s = (key >> 7) % 3 - 1                                    # -1, 0, +1 per unordered index pair: the perturbation is symmetric
Kp = K.copy(); Kp.data = K.data * (1.0 + s * 2.2204460492503131e-16)
print("entries changed:", float(np.mean(Kp.data != K.data)))
wp, _ = refine(operator_from_float64(Kp, cs, cores), mf.solve, b, w.copy(), steps=8)
J, Jp = o.compliance(w), o.compliance(wp)
print(f"displacement changes by {np.abs(wp - w).max() / np.abs(w).max():.2e} (max norm), compliance by {abs(Jp / J - 1):.2e}")
