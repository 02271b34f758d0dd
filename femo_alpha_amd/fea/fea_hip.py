"""FEA registry + the free functions the operators import, over the HIP backend.

Mirror of the slice of ``femo_alpha/fea/fea_dolfinx.py`` and ``fea/utils_dolfinx.py`` that
``csdl_alpha_opt/state_operation.py:1-4`` and ``output_operation.py:1`` import -- same names,
argument meaning and error behaviour -- with dolfinx Functions / UFL forms / PETSc matrices
replaced by light handles onto a ``ShellContext`` (all arithmetic runs in libfemo_hip.so):

  dolfinx.fem.Function      -> Function        (a named buffer resident in HBM)
  UFL form                  -> Form / ResidualForm (a name the HIP library knows)
  derivative(form, f)       -> PartialForm
  PETSc Mat from assembleMatrix(dR/dx) -> JacobianOperator (matrix-free, applied on the device)
  PETSc KSP + MUMPS LU      -> LinearSolver (device PCG)
"""
from __future__ import annotations

import numpy as np

from ..backend import ShellContext
from ..mesh_io import readFEAMesh, reconstructFEAMesh            # noqa: F401  (same names as fea/utils_dolfinx.py)


class FunctionSpace:
    """kind: 'W' (CG2xCG1 state), 'VT' (thickness-like scalars), 'VF' (pressure), 'VU' (mesh motion)."""

    def __init__(self, ctx: ShellContext, kind: str):
        self.ctx, self.kind = ctx, kind
        m = ctx.mesh
        self.dim = {"W": m.ndof, "VT": m.nel if ctx.element_wise_material else m.nn,
                    "VF": 3 * (m.nel if ctx.elementwise_pressure else m.nn), "VU": 3 * m.nn}[kind]


class Function:
    """A vector of ``function_space.dim`` doubles.  Once bound to a role of the PDE ('thickness',
    'F_solid', 'E', 'nu', 'density', 'uhat' or 'state') its values live on the device; unbound
    functions (d_state, d_residual work vectors) are host arrays."""

    def __init__(self, function_space: FunctionSpace, role: str | None = None):
        self.function_space = function_space
        self.role = role
        self._host = np.zeros(function_space.dim)

    @property
    def ctx(self):
        return self.function_space.ctx

    def bind(self, role):
        if self.role not in (None, role):
            raise ValueError(f"Function already plays the role '{self.role}', cannot also be '{role}'")
        self.role = role
        return self

    def get(self):
        if self.role == "state":
            return self.ctx.get_state()
        if self.role is not None:
            return self.ctx.get_field(self.role)
        return self._host.copy()

    def set(self, values):
        values = np.asarray(values, dtype=np.float64).ravel()
        if values.size == 1 and self.function_space.dim != 1:
            values = np.full(self.function_space.dim, values[0])     # utils_dolfinx.py:327-330
        if values.size != self.function_space.dim:
            raise ValueError(f"expected {self.function_space.dim} values, got {values.size}")
        if self.role == "state":
            self.ctx.set_state(values)
        elif self.role is not None:
            self.ctx.set_field(self.role, values)
        else:
            self._host[:] = values


    # ---- read-only stand-ins for the dolfinx post-processing idiom of the reference's examples
    #      (``w.sub(0).collapse().x.array``, ex_simple_shell_opt.py:142-144)
    @property
    def x(self):
        return _ArrayView(self.get())

    def sub(self, i):
        """Sub-function of the mixed state: 0 = mid-surface displacement (P2 nodes, xyz), 1 = rotation (vertices, xyz)."""
        if self.function_space.kind != "W":
            raise ValueError("only the mixed state function has sub-functions")
        nu = self.ctx.mesh.ndof_u
        w = self.get()
        if i == 0:
            return _SubFunction(w[:nu])
        if i == 1:
            return _SubFunction(w[nu:])
        raise IndexError("the shell state has two sub-functions: displacement (0) and rotation (1)")


class _ArrayView:
    def __init__(self, a):
        self.array = a


class _SubFunction:
    def __init__(self, values):
        self._values = values

    def collapse(self):
        return self

    @property
    def x(self):
        return _ArrayView(self._values)


class Form:
    """Scalar output known to the backend: 'compliance', 'mass', 'elastic_energy', 'pnorm_stress', 'volume'.
    ``subdomain`` restricts the stress aggregate to one tagged set of cells (the reference's ``dxx(i)`` measure)."""

    def __init__(self, ctx, name, subdomain=-1, stress_params=None):
        self.ctx, self.name, self.subdomain = ctx, name, subdomain
        # (m, rho, regularisation coefficient) of a p-norm stress form: every form keeps its own, as every UFL expression of the
        # reference does (rm_shell_pde.py:112-128), and hands them to the context right before it is evaluated
        self.stress_params = stress_params

    def _select(self):
        if self.subdomain >= 0 or getattr(self.ctx, "_subdomain", -1) >= 0:
            self.ctx.select_subdomain(self.subdomain)
            self.ctx._subdomain = self.subdomain
        if self.stress_params is not None and getattr(self.ctx, "_stress_params", None) != self.stress_params:
            m, rho, regc = self.stress_params
            self.ctx.set_option("stress_regularization", regc)
            self.ctx.set_stress_params(m, rho)
            self.ctx._stress_params = self.stress_params


class FieldForm:
    """Field output known to the backend: 'stress' (top-surface von Mises, projected onto DG1)."""

    def __init__(self, ctx, name):
        self.ctx, self.name = ctx, name


class ResidualForm:
    """R(w; thickness, F_solid, E, nu, uhat) of the shell, Dirichlet treatment included."""

    def __init__(self, ctx):
        self.ctx = ctx


class PartialForm:
    def __init__(self, form, wrt: Function):
        if wrt.role is None:
            raise ValueError("cannot differentiate with respect to a Function that is not part of the PDE")
        self.form, self.wrt = form, wrt


class JacobianOperator:
    """dR/d(wrt) as a matrix-free operator (what ``assembleMatrix`` hands back)."""

    def __init__(self, partial: PartialForm, bcs=None):
        if not isinstance(partial.form, ResidualForm):
            raise TypeError("assembleMatrix expects a derivative of the residual form")
        self.ctx = partial.form.ctx
        self.wrt = partial.wrt.role
        self.shape = (self.ctx.ndof, partial.wrt.function_space.dim)

    def mult(self, x):
        if self.wrt != "state":
            raise NotImplementedError("forward products with dR/d(input) are not provided "
                                      "(the reference's fwd mode raises KeyError, state_operation.py:167-171)")
        return self.ctx.apply_K(x)

    def multTranspose(self, lam):
        if self.wrt == "state":
            return self.ctx.apply_K(lam)                   # symmetric
        return self.ctx.dRdarg_T(self.wrt, lam)


class LinearSolver:
    """Stands where the reference keeps a PETSc KSP with a MUMPS LU of A (utils_dolfinx.py:514-531)."""

    def __init__(self, A: JacobianOperator):
        self.A = A
        self.iterations, self.relres = 0, 0.0

    def solve(self, rhs):
        rhs = np.asarray(rhs, dtype=np.float64)
        if rhs.ndim == 2:
            # several right-hand sides (rows): ONE grouped solve -- the factor is streamed once per group of up to four vectors
            # (femo_solve_linear_multi) instead of once per vector
            x, its, rrs = self.A.ctx.solve_linear_multi(rhs)
            self.iterations, self.relres = int(its.max()), float(rrs.max())
            return x
        x, self.iterations, self.relres = self.A.ctx.solve_linear(rhs)
        return x


# ------------------------------------------------------------------------------- free functions
def getFuncArray(v: Function):
    return v.get()


def setFuncArray(v: Function, v_array):
    v.set(v_array)


def update(v: Function, v_values):
    """Copy ``v_values`` into ``v``; a length-1 array broadcasts (utils_dolfinx.py:319-330)."""
    v.set(v_values)


def computePartials(form, function):
    return PartialForm(form, function)


def createFunction(function: Function):
    return Function(function.function_space)


def assembleScalar(c: Form):
    c._select()
    return c.ctx.functional(c.name)


def assembleVector(v):
    if isinstance(v, ResidualForm):
        return v.ctx.residual()
    if isinstance(v, PartialForm) and isinstance(v.form, Form):
        wrt = "disp_solid" if v.wrt.role == "state" else v.wrt.role
        v.form._select()
        return v.form.ctx.dfunctional(v.form.name, wrt)
    raise TypeError("assembleVector: unsupported form")


def assembleMatrix(M, bcs=()):
    return JacobianOperator(M, bcs)


def assembleSystem(J, F, bcs=()):
    """(A, b): the operator with the Dirichlet treatment applied, and the residual vector."""
    return JacobianOperator(J, bcs), F.ctx.residual()


def assemble(f, dim=0, bcs=()):
    if dim == 0:
        return assembleScalar(f)
    if dim == 1:
        return assembleVector(f)
    raise TypeError("Invalid type for assembly.")


def computeMatVecProductFwd(A: JacobianOperator, x):
    return A.mult(x.get() if isinstance(x, Function) else x)


def computeMatVecProductBwd(A: JacobianOperator, R):
    return A.multTranspose(R.get() if isinstance(R, Function) else R)


def setUpKSP_MUMPS(A: JacobianOperator):
    return LinearSolver(A)


def solveNonlinear(res: ResidualForm, func: Function, bc, solver="Newton", report=False, initialize=False):
    """The reference runs 3 Newton iterations with unattainable tolerances on a residual that is
    linear in w (utils_dolfinx.py:438-468) == one linear solve; here one device PCG solve, warm-started
    from the stored state unless ``initialize``."""
    it, rr = res.ctx.solve_state(zero_guess=bool(initialize))
    if report:
        print(f"PCG iterations: {it}, relative residual: {rr:.3e}")
    return it, rr


class FEA:
    """Registry of inputs / states / outputs read by the operators -- the dictionaries and attributes
    of the reference class (fea_dolfinx.py:28-136), minus the XDMF recorders."""

    def __init__(self, mesh):
        self.mesh = mesh
        self.inputs_dict = dict()
        self.states_dict = dict()
        self.outputs_dict = dict()
        self.outputs_field_dict = dict()
        self.bc = []
        self.PDE_SOLVER = "Newton"
        self.REPORT = False
        self.ubc = None
        self.custom_solve = None
        self.opt_iter = 0
        self.initial_solve = True
        self.initialize = False
        self.record = False
        self.recorder_path = "records"
        self.linear_problem = False
        self.nel = mesh.nel
        self.nn = mesh.nn
        self.last_solve = (0, 0.0)

    def add_input(self, name, function: Function, init_val=1.0, record=False):
        if name in self.inputs_dict:
            raise ValueError("name has already been used for an input")
        function.set(np.array([init_val]))
        self.inputs_dict[name] = dict(function=function, function_space=function.function_space,
                                      shape=function.function_space.dim, recorder=None, record=False)

    def add_state(self, name, function: Function, residual_form, arguments, dR_du=None, dR_df_list=None, record=False):
        if dR_du is None:
            dR_du = PartialForm(residual_form, function)
        self.states_dict[name] = dict(function=function, residual_form=residual_form,
                                      function_space=function.function_space, shape=function.function_space.dim,
                                      d_residual=Function(function.function_space),
                                      d_state=Function(function.function_space), dR_du=dR_du,
                                      dR_df_list=dR_df_list, arguments=arguments, recorder=None, record=False)

    def add_output(self, name, form, arguments):
        partials = []
        for argument in arguments:
            if argument in self.inputs_dict:
                partials.append(PartialForm(form, self.inputs_dict[argument]["function"]))
            elif argument in self.states_dict:
                partials.append(PartialForm(form, self.states_dict[argument]["function"]))
        self.outputs_dict[name] = dict(form=form, shape=1, arguments=arguments, partials=partials)

    def add_field_output(self, name, form, arguments, function_space=("CG", 1), record=False, vtk=False):
        if tuple(function_space) != ("DG", 1):
            raise NotImplementedError("field outputs are projected onto ('DG', 1), the space rm_shell_model.py:237 asks for")
        space = FunctionSpace.__new__(FunctionSpace)
        space.ctx, space.kind, space.dim = form.ctx, "DG1", form.ctx.mesh.nvc * form.ctx.mesh.nel
        self.outputs_field_dict[name] = dict(form=form, function=Function(space), shape=space.dim, arguments=arguments,
                                             partials=[], recorder=None, record=False)

    def projectFieldOutput(self, form, func):
        """L2 projection of the field form onto its DG1 function (fea_dolfinx.py:205-206)."""
        func.set(form.ctx.field_output(form.name))

    def add_strong_bc(self, ubc, locate_BC_list, function_space=None):
        for dofs in locate_BC_list:
            self.bc.append(DirichletBC(np.asarray(dofs, dtype=np.int32)))

    def solve(self, res, func, bc):
        if self.custom_solve is not None and self.initial_solve:
            self.custom_solve(res, func, bc, self.REPORT)
        else:
            self.last_solve = solveNonlinear(res, func, bc, self.PDE_SOLVER, self.REPORT, self.initialize)

    def solveLinearFwd(self, du, A, dR, dR_array, ksp=None):
        """du = A^-1 dR.  ``dR_array`` of shape (k, ndof): k right-hand sides in one grouped solve (an extension of the reference's
        one-vector call, fea_dolfinx.py:172-187: the seeds of several outputs or several tangent columns are known together)."""
        ksp = ksp or LinearSolver(A)
        if np.ndim(dR_array) == 2:
            return ksp.solve(dR_array)
        setFuncArray(dR, dR_array)
        du.set(ksp.solve(dR.get()))
        return du.get()

    def solveLinearBwd(self, dR, A, du, du_array, ksp=None):
        """dR = A^-T du; A is symmetric so the same solve serves (reference quirk Q3).  ``du_array`` of shape (k, ndof): k adjoint
        right-hand sides -- one per registered output of the state, rm_shell_model.py:221-253 -- in one grouped solve."""
        ksp = ksp or LinearSolver(A)
        if np.ndim(du_array) == 2:
            return ksp.solve(du_array)
        setFuncArray(du, du_array)
        dR.set(ksp.solve(du.get()))
        return dR.get()


class DirichletBC:
    def __init__(self, dofs):
        self._dofs = np.unique(dofs)

    def dof_indices(self):
        return self._dofs, self._dofs.size
