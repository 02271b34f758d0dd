import numpy as np, sys
sys.path.insert(0, '/root/repo')
sys.path.insert(0, '.')
from tests.test_gpu_fullsize import _context, _solver
for kind in ("plate250k", "wing1m"):
    m, fields, marker, rng = _context(kind)
    c = _solver(m, fields, marker)
    it, rr = c.solve_state(zero_guess=True); w1 = c.get_state()
    c.set_solver(preconditioner=2, rtol=1e-30, maxit=6, check_every=1)
    it2, rr2 = c.solve_state(zero_guess=True); w2 = c.get_state()
    F = c.load_vector()
    print(kind, 'its', it, rr, '|', it2, rr2, 'rel change', np.linalg.norm(w1 - w2) / np.linalg.norm(w1), 'true res', np.linalg.norm(c.residual(w1)) / np.linalg.norm(F), np.linalg.norm(c.residual(w2)) / np.linalg.norm(F))
    c.close()
