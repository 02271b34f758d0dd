"""FD-versus-tangent distance of tests/test_gpu_dynamic.py::test_forward_mode_of_the_transient_operator for both front assemblies."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from femo_alpha_amd.mesh import plate_mesh
from femo_alpha_amd.dynamic_rm_shell.plate_sim import PlateSim
from test_gpu_dynamic import _gust
for ewt, sw in [(False, False), (True, True), (False, True)]:
    for eps in (1e-5, 1e-4, 1e-3):
        mesh = plate_mesh(2.0, 10.0, 4, 12)
        E, nu, rho, dt, N = 1e8, 0.3, 10.0, 0.01, 8
        ps = PlateSim(mesh, E, nu, rho, dt, N, element_wise_thickness=ewt, add_self_weight=sw, quad_deg=3, leaf_size=8, rtol=1e-12)
        n_t = mesh.nel if ewt else mesh.nn
        rng = np.random.default_rng(3)
        t0 = 0.1 * (1 + 0.2 * rng.uniform(-1, 1, n_t))
        F = _gust(N + 1, mesh.nn, dt)
        def march(t, Fh=F):
            ps.update_f_history(Fh); ps.update_t(t)
            return ps.solve_dynamic_problem()
        d = rng.uniform(-1, 1, n_t) * t0
        fd = (march(t0 + eps * d) - march(t0 - eps * d)) / (2 * eps)
        march(t0)
        dW = -ps.tangent_history(ps.jacobian_products_fwd(dthickness=d))
        print(os.environ.get("FEMO_OPTIONS"), ewt, sw, eps, np.abs(dW - fd).max() / np.abs(fd).max(), flush=True)
