"""Stand-in for the slice of ``csdl_alpha`` the operator surface touches.

The reference's operators subclass ``csdl.CustomExplicitOperation`` /
``csdl.experimental.CustomImplicitOperation`` and are driven by a CSDL simulator
(reference femo_alpha/csdl_alpha_opt/state_operation.py:8, output_operation.py:6,72).
``csdl_alpha`` is not installable here; when it *is* importable the
package uses the real thing (see ``femo_alpha_amd.csdl``), otherwise this module -- scaffolding for the tests and examples,
not part of the product package -- provides the
same protocol -- ``declare_input / create_output / declare_derivative_parameters``, inline
execution as with ``csdl.Recorder(inline=True)`` (ex_simple_shell_opt.py:58), and a small
reverse-mode driver (``Recorder.compute_totals`` / ``check_totals``) that calls
``compute_derivatives``, ``apply_inverse_jacobian(mode='rev')`` and
``compute_jacvec_product(mode='rev')`` the way the CSDL simulator does.
"""
from __future__ import annotations

import numpy as np

__all__ = ["Variable", "VariableGroup", "CustomExplicitOperation", "CustomImplicitOperation",
           "Recorder", "check_parameter", "experimental", "reshape", "transpose"]

_active = []


class Variable:
    def __init__(self, value=None, shape=None, name=None):
        if value is None:
            value = np.zeros(shape if shape is not None else (1,))
        value = np.array(value, dtype=np.float64)
        if shape is not None and value.shape != tuple(shape):
            value = np.broadcast_to(value, shape).copy()
        self.value = value
        self.shape = value.shape
        self.names = [name] if name else []
        self._producer = None       # (op, key) for operation outputs; ('index', parent, idx, shape) for views

    @property
    def name(self):
        return self.names[0] if self.names else None

    def add_name(self, name):
        self.names.insert(0, name)
        return self

    def set_value(self, v):
        self.value[...] = np.asarray(v, dtype=np.float64).reshape(self.shape)

    # optimisation roles: recorded on the variable (and on the active recorder) for whoever drives the optimiser;
    # the stand-in has no optimiser of its own (the reference hands the recorder to modopt, ex_simple_shell_opt.py:114-131)
    def _role(self, kind, **kw):
        self.optimization_role = (kind, {k: v for k, v in kw.items() if v is not None})
        if _active:
            getattr(_active[-1], kind + "s").append(self)
        return self

    def set_as_design_variable(self, upper=None, lower=None, scaler=None):
        return self._role("design_variable", upper=upper, lower=lower, scaler=scaler)

    def set_as_constraint(self, upper=None, lower=None, equals=None, scaler=None):
        return self._role("constraint", upper=upper, lower=lower, equals=equals, scaler=scaler)

    def set_as_objective(self, scaler=None):
        return self._role("objective", scaler=scaler)

    # fancy-index / reshape views that stay differentiable (the reference permutes inputs this way,
    # rm_shell_model.py:398-438)
    def __getitem__(self, idx):
        out = Variable(self.value[idx])
        out._producer = ("index", self, idx)
        _ViewOp.register(out, lambda: self.value[idx])
        return out

    def reshape(self, shape):
        out = Variable(self.value.reshape(shape))
        out._producer = ("reshape", self)
        _ViewOp.register(out, lambda: self.value.reshape(shape))
        return out

    # scalar arithmetic used by the stress aggregation (rm_shell_model.py:501-503): c * v and v ** p
    def _unary(self, fn, dfn):
        out = Variable(fn(self.value))
        out._producer = ("unary", self, dfn)
        _ViewOp.register(out, lambda: fn(self.value))
        return out

    def __mul__(self, c):
        c = float(c)
        return self._unary(lambda v: c * v, lambda v: c * np.ones_like(v))

    __rmul__ = __mul__

    def __truediv__(self, c):
        return self.__mul__(1.0 / float(c))

    def __pow__(self, p):
        p = float(p)
        return self._unary(lambda v: v ** p, lambda v: p * v ** (p - 1.0))

    def __repr__(self):
        return f"Variable({self.name}, shape={self.shape})"


def reshape(v, shape):
    return v.reshape(shape)


def transpose(v):
    out = Variable(v.value.T.copy())
    out._producer = ("transpose", v)
    _ViewOp.register(out, lambda: v.value.T)
    return out


class _ViewOp:
    """Re-evaluates an index / reshape / transpose view when the graph is re-run."""
    _outputs = {}

    def __init__(self, var, fn):
        self.var, self.fn = var, fn

    @staticmethod
    def register(var, fn):
        if _active:
            _active[-1].ops.append(_ViewOp(var, fn))

    def _run(self):
        self.var.value[...] = self.fn()


class VariableGroup:
    """Attribute bag; missing attributes read as None like ``getattr(inputs, name)`` checks expect."""

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return None


def check_parameter(value, name, types=None):
    if types is not None and not isinstance(value, types):
        raise TypeError(f"'{name}' must be {types}, got {type(value).__name__}")


class _Operation:
    def __init__(self):
        self._inputs = {}
        self._outputs = {}
        self._declared = []

    def declare_input(self, key, variable):
        if not isinstance(variable, Variable):
            raise TypeError(f"input '{key}' must be a Variable")
        self._inputs[key] = variable

    def create_output(self, key, shape):
        shape = tuple(shape) if not np.isscalar(shape) else (int(shape),)
        v = Variable(shape=shape)
        v._producer = (self, key)
        self._outputs[key] = v
        rec = _active[-1] if _active else None
        if rec is not None and self not in rec.ops:
            rec.ops.append(self)
        return v

    def declare_derivative_parameters(self, of, wrt, dependent=True):
        self._declared.append((of, wrt, dependent))

    def _input_vals(self):
        return {k: v.value.reshape(-1) if v.value.ndim > 1 else v.value for k, v in self._inputs.items()}

    def _finish_evaluate(self):
        rec = _active[-1] if _active else None
        if rec is None or rec.inline:
            self._run()


class CustomExplicitOperation(_Operation):
    def _run(self):
        out = {}
        self.compute(self._input_vals(), out)
        for k, v in out.items():
            self._outputs[k].value[...] = np.asarray(v, dtype=np.float64).reshape(self._outputs[k].shape)


class CustomImplicitOperation(_Operation):
    def _run(self):
        out = {}
        self.solve_residual_equations(self._input_vals(), out)
        for k, v in out.items():
            self._outputs[k].value[...] = np.asarray(v, dtype=np.float64).reshape(self._outputs[k].shape)


class Recorder:
    """Records operations in creation order; ``inline=True`` runs them as they are created."""

    def __init__(self, inline=True):
        self.inline = inline
        self.ops = []
        self.design_variables, self.constraints, self.objectives = [], [], []

    def start(self):
        _active.append(self)
        return self

    def stop(self):
        if _active and _active[-1] is self:
            _active.pop()

    # ------------------------------------------------------------------ execution
    def run(self):
        for op in self.ops:
            op._run()

    # ------------------------------------------------------------------ reverse mode
    @staticmethod
    def _push_to_source(var, bar, sink):
        """Propagate an adjoint through index/reshape/transpose views down to a root Variable
        or an operation output; accumulates into ``sink[id(root)]``."""
        while True:
            p = var._producer
            if isinstance(p, tuple) and p and p[0] == "index":
                parent, idx = p[1], p[2]
                full = np.zeros(parent.shape)
                np.add.at(full, idx, bar.reshape(var.shape))
                var, bar = parent, full
            elif isinstance(p, tuple) and p and p[0] == "reshape":
                var, bar = p[1], bar.reshape(p[1].shape)
            elif isinstance(p, tuple) and p and p[0] == "transpose":
                var, bar = p[1], bar.reshape(var.shape).T
            elif isinstance(p, tuple) and p and p[0] == "unary":
                var, bar = p[1], bar.reshape(var.shape) * p[2](p[1].value)
            else:
                break
        key = id(var)
        if key in sink:
            sink[key][1] += bar.reshape(var.shape)
        else:
            sink[key] = [var, bar.reshape(var.shape).copy()]

    def compute_totals(self, of, wrt):
        """d of / d wrt for a scalar ``of`` (shape (1,)) by one reverse sweep."""
        if int(np.prod(of.shape)) != 1:
            raise ValueError("compute_totals: 'of' must be a scalar output")
        sink = {}
        self._push_to_source(of, np.ones(of.shape), sink)
        for op in reversed(self.ops):
            for key, outvar in op._outputs.items():
                ent = sink.pop(id(outvar), None)
                if ent is None:
                    continue
                bar = ent[1].reshape(-1)
                ivals = op._input_vals()
                ovals = {k: v.value.reshape(-1) for k, v in op._outputs.items()}
                if isinstance(op, CustomImplicitOperation):
                    d_res = {}
                    op.apply_inverse_jacobian(ivals, ovals, {key: bar.copy()}, d_res, "rev")
                    d_in = {k: np.zeros(v.value.size) for k, v in op._inputs.items()}
                    op.compute_jacvec_product(ivals, ovals, d_in, {}, {key: d_res[key]}, "rev")
                    for k, v in op._inputs.items():
                        self._push_to_source(v, -d_in[k], sink)        # dy/dx = -(dR/dy)^-1 dR/dx
                elif hasattr(op, "vjp"):
                    (k, v), = op._inputs.items()
                    self._push_to_source(v, np.asarray(op.vjp(bar), dtype=np.float64), sink)
                else:
                    derivs = {}
                    op.compute_derivatives(ivals, ovals, derivs)
                    for k, v in op._inputs.items():
                        if (key, k) in derivs:
                            J = np.asarray(derivs[(key, k)], dtype=np.float64).reshape(-1)
                            self._push_to_source(v, bar[0] * J, sink)
        ent = sink.get(id(wrt))
        return np.zeros(wrt.shape) if ent is None else ent[1]

    def check_totals(self, of, wrt, step=1e-6, indices=None, raise_on_error=False, rtol=1e-5):
        """Adjoint against central finite differences on selected entries of ``wrt`` -- the
        reference's own verification method (ex_simple_shell_opt.py:109-111)."""
        ana = self.compute_totals(of, wrt).reshape(-1)
        base = wrt.value.copy()
        flat = wrt.value.reshape(-1)
        idx = range(flat.size) if indices is None else indices
        rows = []
        for i in idx:
            h = step * (abs(flat[i]) if flat[i] != 0.0 else 1.0)      # relative step
            flat[i] += h; self.run(); fp = float(of.value.reshape(-1)[0])
            flat[i] -= 2 * h; self.run(); fm = float(of.value.reshape(-1)[0])
            flat[i] += h
            fd = (fp - fm) / (2 * h)
            rows.append((i, ana[i], fd, abs(ana[i] - fd) / max(abs(fd), 1e-300)))
        wrt.value[...] = base
        self.run()
        if raise_on_error and any(r[3] > rtol for r in rows):
            raise AssertionError(f"check_totals failed: {rows}")
        return rows


class _Experimental:
    CustomImplicitOperation = CustomImplicitOperation

    class PySimulator:
        def __init__(self, recorder):
            self.recorder = recorder

        def run(self):
            self.recorder.run()

        def compute_totals(self, ofs, wrts):
            return {(o.name, w.name): self.recorder.compute_totals(o, w) for o in ofs for w in wrts}

        def check_totals(self, ofs, wrts, step=1e-6, **kw):
            return {(o.name, w.name): self.recorder.check_totals(o, w, step=step, **kw) for o in ofs for w in wrts}


experimental = _Experimental()
