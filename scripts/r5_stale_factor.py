"""Option "stale_factor" measured: (a) PCG iterations and time of a forward solve with the PREVIOUS design's factor
against the relative thickness change, config 2 (plate 58 x 290, 255 438 DOF); (b) the SLSQP loop of
tests/test_gpu_operators.py::test_thickness_optimisation_loop at config 2 size with and without the option: wall time per optimiser
iteration, number of factorisations, final compliance.      python scripts/r5_stale_factor.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
from femo_alpha_amd import csdl
from femo_alpha_amd.backend import ShellContext
from femo_alpha_amd.mesh import plate_mesh
from femo_alpha_amd.rm_shell.rm_shell_model import RMShellModel
from optimize import slsqp

CLAMP = lambda x: np.less(x[0], 3e-16)
m = plate_mesh(2.0, 10.0, 58, 290)
rng = np.random.default_rng(0)
h0 = 0.1 * (1 + 0.2 * rng.uniform(-1, 1, m.nn))
c = ShellContext(m)
for k, v in dict(thickness=h0, E=[1e8], nu=[0.3], density=[10.0], F_solid=np.tile([0.0, 0.0, 5.0], (m.nn, 1))).items():
    c.set_field(k, v)
c.set_penalty_facets(m.penalty_facets(CLAMP))
c.enable_frontal()
c.set_solver(preconditioner=2, rtol=1e-10, maxit=200, check_every=1)
print(f"(a) config 2, {m.ndof} DOF, rtol 1e-10: forward solve of design h0 (1 + d U(-1, 1)) with the factor of h0 kept")
print("    d        PCG iterations   solve ms   |  always re-factorise: iterations   ms")
c.set_option("stale_rel", 1.0)                         # the gate off: this table is what the gate is set by
for d in (1e-4, 1e-3, 3e-3, 1e-2, 3e-2, 0.1, 0.2, 0.3, 0.5):
    h = h0 * (1 + d * np.random.default_rng(1).uniform(-1, 1, m.nn))
    row = []
    for stale in (1000, 0):
        c.set_option("stale_factor", stale)
        c.set_field("thickness", h0); c.factorize(); c.set_field("E", [1e8]); c.solve_state(zero_guess=True)      # the factor (and its snapshot) of h0
        ts = []
        for _ in range(3):
            c.set_field("thickness", h)
            t0 = time.perf_counter(); it, rr = c.solve_state(zero_guess=True); ts.append(time.perf_counter() - t0)
            assert c.last_timing()["factor_state"] == (1 if stale else 0)
        row += [it, np.median(ts) * 1e3]
    print(f"    {d:7.0e}   {row[0]:8d}        {row[1]:8.2f}   |  {row[2]:8d}                        {row[3]:6.2f}", flush=True)
c.close()

print("\n(a') the same at config 3 (wing1m, 1 015 470 DOF): forward solve + adjoint gradient of design h0 (1 + d U) with the factor of h0 kept (stale_rel 1: gate off)")
print("    d        forward its / ms   adjoint its / ms   step ms  |  always re-factorise: forward its / ms   adjoint its / ms   step ms", flush=True)
from bench import make_workload
sys_argv = sys.argv; sys.argv = [sys.argv[0]]
mw, fw, mkw, _ = make_workload("wing1m")
cw = ShellContext(mw)
for k, v in fw.items():
    cw.set_field(k, v)
cw.set_penalty_facets(mw.penalty_facets(mkw))
cw.enable_frontal()
cw.set_solver(preconditioner=2, rtol=1e-10, maxit=200, check_every=1)
hw = np.full(mw.nn, 1.27e-3)
cw.set_option("stale_rel", 1.0)
for d in (1e-4, 1e-3, 3e-3, 1e-2, 3e-2, 0.1):
    h = hw * (1 + d * np.random.default_rng(1).uniform(-1, 1, mw.nn))
    row = []
    for stale in (1000, 0):
        cw.set_option("stale_factor", stale)
        cw.set_field("thickness", hw); cw.factorize(); cw.set_field("E", [73.1e9]); cw.solve_state(zero_guess=True)   # the factor (and its snapshot) of hw
        tf, ta = [], []
        for _ in range(3):
            cw.set_field("thickness", h)
            t0 = time.perf_counter(); it, _ = cw.solve_state(zero_guess=True); t1 = time.perf_counter()
            g, it2, _ = cw.total_gradient("compliance", "thickness"); t2 = time.perf_counter()
            tf.append(t1 - t0); ta.append(t2 - t1)
        row += [it, np.median(tf) * 1e3, it2, np.median(ta) * 1e3]
    print(f"    {d:7.0e}   {row[0]:4d} / {row[1]:6.2f}      {row[2]:4d} / {row[3]:6.2f}     {row[1] + row[3]:6.2f}   |  {row[4]:4d} / {row[5]:6.2f}                         {row[6]:4d} / {row[7]:6.2f}     {row[5] + row[7]:6.2f}", flush=True)
cw.close()

print("\n(b) SLSQP (scipy) on 30 spanwise thickness stations (piecewise linear, bounds 0.02..0.2, mass held at its initial value), compliance objective, config 2 size:")
print("    every function evaluation a forward solve, every gradient an adjoint solve (the loop of tests/test_gpu_operators.py::test_thickness_optimisation_loop;")
print("    scipy's SLSQP is dense in the design variables, so the 17 169 nodal thicknesses are driven through 30 stations)", flush=True)
from scipy.optimize import minimize
xs = np.linspace(0.0, 10.0, 30)
B = np.zeros((m.nn, xs.size))                      # nodal thickness = B @ stations (hat functions along the span)
pos = np.clip(np.searchsorted(xs, m.nodes[:, 0]) - 1, 0, xs.size - 2)
tloc = (m.nodes[:, 0] - xs[pos]) / (xs[pos + 1] - xs[pos])
B[np.arange(m.nn), pos] = 1 - tloc; B[np.arange(m.nn), pos + 1] = tloc
for stale in (0, 8):
    c = ShellContext(m)
    for k, v in dict(thickness=np.full(m.nn, 0.1), E=[1e8], nu=[0.3], density=[10.0], F_solid=np.tile([0.0, 0.0, 5.0], (m.nn, 1))).items():
        c.set_field(k, v)
    c.set_penalty_facets(m.penalty_facets(CLAMP))
    c.enable_frontal()
    c.set_solver(preconditioner=2, rtol=1e-10, maxit=200, check_every=1)
    c.set_option("stale_factor", stale)
    log = []
    state = {"x": None}

    def put(x):
        if state["x"] is not None and np.array_equal(state["x"], x):
            return
        c.set_field("thickness", B @ x)
        t0 = time.perf_counter()
        it, _ = c.solve_state(zero_guess=True)
        log.append((it, c.last_timing()["factor_state"], (time.perf_counter() - t0) * 1e3, 0.0 if state["x"] is None else float(np.abs(x - state["x"]).max() / np.abs(state["x"]).max())))
        state["x"] = x.copy()

    x0 = np.full(xs.size, 0.1)
    put(x0)
    J0, m0 = c.functional("compliance"), c.functional("mass")
    fun = lambda x: (put(x), c.functional("compliance") / J0)[1]
    jac = lambda x: (put(x), B.T @ c.total_gradient("compliance", "thickness")[0] / J0)[1]
    con = {"type": "eq", "fun": lambda x: (put(x), c.functional("mass") / m0 - 1.0)[1],
           "jac": lambda x: (put(x), B.T @ c.dfunctional("mass", "thickness") / m0)[1]}
    t0 = time.perf_counter()
    res = minimize(fun, x0, jac=jac, bounds=[(0.02, 0.2)] * xs.size, constraints=[con], method="SLSQP", options=dict(maxiter=25, ftol=1e-10))
    dt = time.perf_counter() - t0
    its = np.array([l[0] for l in log[1:]]); st = np.array([l[1] for l in log[1:]]); ms = np.array([l[2] for l in log[1:]]); dx = np.array([l[3] for l in log[1:]])
    print(f"    stale_factor {stale}: {res.nit} optimiser iterations, {its.size} forward solves of which {int((st != 1).sum())} factorised; PCG iterations median "
          f"{int(np.median(its))} max {its.max()}; forward solve median {np.median(ms):.2f} ms, sum {ms.sum():.0f} ms; wall {dt:.2f} s = {dt / max(res.nit, 1) * 1e3:.0f} ms "
          f"per optimiser iteration; compliance ratio {res.fun:.9f}, mass error {abs(c.functional('mass') / m0 - 1):.1e}", flush=True)
    print("      relative design step | PCG iterations | factor state, per forward solve: " + " ".join(f"{d:.0e}|{i}|{s_}" for d, i, s_ in zip(dx, its, st)), flush=True)
    c.close()
