"""A small SLSQP driver over the csdl stand-in's recorder -- the loop the reference hands to modopt's ``PySLSQP``
(examples/advanced_examples/simple_shell_opt/ex_simple_shell_opt.py:114-131): design variables, constraints and the
objective are the variables tagged with ``set_as_design_variable`` / ``set_as_constraint`` / ``set_as_objective``;
every function evaluation re-runs the recorded operations (forward solve on the GPU) and every gradient is one
reverse sweep (adjoint solve on the GPU).  Only for the in-tree stand-in: with ``csdl_alpha`` + ``modopt`` installed
the reference's own driver applies unchanged.
"""
from __future__ import annotations

import numpy as np


def slsqp(recorder, maxiter=50, ftol=1e-9, callback=None):
    """Minimise the recorder's objective with scipy's SLSQP.  Returns scipy's ``OptimizeResult``; the design
    variables hold the optimum afterwards and the recorded outputs are consistent with it."""
    from scipy.optimize import minimize
    if len(recorder.objectives) != 1:
        raise ValueError("exactly one variable must be tagged with set_as_objective()")
    if not recorder.design_variables:
        raise ValueError("no design variables: tag at least one variable with set_as_design_variable()")
    obj = recorder.objectives[0]
    dvs = list(recorder.design_variables)
    sizes = [int(np.prod(v.shape)) for v in dvs]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    oscale = obj.optimization_role[1].get("scaler", 1.0)
    state = {"x": None}

    def put(x):
        if state["x"] is not None and np.array_equal(state["x"], x):
            return
        for v, a, b in zip(dvs, offs[:-1], offs[1:]):
            v.set_value(x[a:b])
        recorder.run()
        state["x"] = x.copy()

    def total(of):
        return np.concatenate([np.ravel(recorder.compute_totals(of, v)) for v in dvs])

    def fun(x):
        put(x)
        return float(np.ravel(obj.value)[0]) * oscale

    def jac(x):
        put(x)
        return total(obj) * oscale

    cons = []
    for cv in recorder.constraints:
        if int(np.prod(cv.shape)) != 1:
            raise ValueError("only scalar constraints are supported by this driver")
        role = cv.optimization_role[1]
        sc = role.get("scaler", 1.0)
        lo, up, eq = role.get("lower"), role.get("upper"), role.get("equals")
        if eq is None and lo is not None and up is not None and lo == up:
            eq = lo

        def make(cv=cv, sc=sc, shift=0.0, sign=1.0, kind="eq"):
            return {"type": kind,
                    "fun": lambda x: (put(x), sign * (float(np.ravel(cv.value)[0]) - shift) * sc)[1],
                    "jac": lambda x: (put(x), sign * sc * total(cv))[1]}
        if eq is not None:
            cons.append(make(shift=eq, kind="eq"))
        else:
            if lo is not None:
                cons.append(make(shift=lo, sign=1.0, kind="ineq"))
            if up is not None:
                cons.append(make(shift=up, sign=-1.0, kind="ineq"))
    bounds = []
    for v, n in zip(dvs, sizes):
        role = v.optimization_role[1]
        lo = np.broadcast_to(np.asarray(role.get("lower", -np.inf), dtype=float), (n,)) if role.get("lower") is not None else [-np.inf] * n
        up = np.broadcast_to(np.asarray(role.get("upper", np.inf), dtype=float), (n,)) if role.get("upper") is not None else [np.inf] * n
        bounds.extend(zip(lo, up))
    x0 = np.concatenate([np.ravel(v.value) for v in dvs])
    res = minimize(fun, x0, jac=jac, bounds=bounds, constraints=cons, method="SLSQP",
                   options={"maxiter": maxiter, "ftol": ftol}, callback=callback)
    put(res.x)
    return res
