"""``RMShellPDE``: the shell's spaces, residual and output forms, resident on one MI355X.

Interface of femo_alpha/rm_shell/rm_shell_pde.py:21-110,173-293.  Where the reference builds UFL
expressions from dolfinx Functions, this class binds ``Function`` handles to the roles they play
in the residual ('thickness', 'F_solid', 'E', 'nu', 'uhat', state) and returns named forms that
libfemo_hip evaluates with hand-written kernels (femo_alpha_amd/csrc/shell_device.h).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from ..backend import PENALTY_BETA, ShellContext
from ..fea.fea_hip import FieldForm, Form, Function, FunctionSpace, ResidualForm


class FacetSet:
    """What a tagged ``ds(100)`` / ``dS(100)`` measure integrates over: (cell, local edge) pairs."""

    def __init__(self, pairs):
        self.pairs = np.asarray(pairs, dtype=np.int32).reshape(-1, 2)


class RMShellPDE:
    def __init__(self, mesh, element_wise_material=False, elementwise_pressure=False, nquad=None, device=0, solver="direct",
                 element_type=None):
        # element_type: 'CG2CG1' (what the reference's RMShellPDE hard-codes, rm_shell_pde.py:27) or 'CG1CG1' (the other quadrilateral /
        # triangle choice of ShellElement.setUpFunctionSpace, linear_shell_model.py:74-79); None: the element the mesh object carries
        if element_type is not None and element_type != mesh.element:
            from ..mesh import ShellMesh
            mesh = ShellMesh(mesh.nodes, mesh.cells, element_type)
        self.mesh = mesh
        self.element_wise_material = element_wise_material
        self.elementwise_pressure = elementwise_pressure
        self.ctx = ShellContext(mesh, element_wise_material, elementwise_pressure, nquad=nquad, device=device)
        # the reference always solves with a sparse direct LU (MUMPS through PETSc, fea/utils_dolfinx.py:466,514-531):
        # the drop-in surface does the same unless the caller opts out ('jacobi': matrix-free Jacobi-PCG, thick plates only)
        if solver == "direct":
            self.ctx.use_direct_solver()
        elif solver != "jacobi":
            raise ValueError("solver must be 'direct' or 'jacobi'")
        self.W = FunctionSpace(self.ctx, "W")          # [CG2]^3 x [CG1]^3, linear_shell_model.py:60-65
        self.VT = FunctionSpace(self.ctx, "VT")        # rm_shell_pde.py:37-40
        self.VF = FunctionSpace(self.ctx, "VF")        # :41-44
        self.VU = FunctionSpace(self.ctx, "VU")        # :45

    # ------------------------------------------------------------------ residual
    def pdeRes(self, h, w, uhat, f, E, nu, penalty=False, dss=None, dSS=None, g=None):
        """Residual of the elastic energy + penalty - load (rm_shell_pde.py:50-58,
        linear_shell_model.py:308-333).  ``g``: the prescribed state of the penalty term (a Function on W; the reference's
        model passes zeros, rm_shell_model.py:183-185); values on strongly imposed DOFs are not used (those are homogeneous,
        as in the reference, rm_shell_model.py:168-180)."""
        h.bind("thickness"); w.bind("state"); uhat.bind("uhat"); f.bind("F_solid"); E.bind("E"); nu.bind("nu")
        if g is not None:
            if not penalty and np.any(g.get()):
                raise NotImplementedError("prescribed values need the penalty treatment (PENALTY_BC=True)")
            g0 = g.get()
            g.bind("dirichlet")
            g.set(g0)
        if penalty:
            pairs = [s.pairs for s in (dss, dSS) if s is not None]
            self.ctx.set_penalty_facets(np.vstack(pairs) if pairs else np.zeros((0, 2), np.int32), PENALTY_BETA)
        return ResidualForm(self.ctx)

    # ------------------------------------------------------------------ outputs
    def compliance(self, u_mid, uhat, h, f):
        """int u.u J dx + regularisation(h) (rm_shell_pde.py:85-89)."""
        return Form(self.ctx, "compliance")

    def regularization(self, h, type=None):
        """Thickness regularisation added to the compliance (rm_shell_pde.py:64-83): 'H1' for nodal thickness,
        'L2' for element-wise thickness -- the two the reference's compliance uses; None means no regularisation."""
        if type is None:
            return 0.0
        expected = "L2" if self.element_wise_material else "H1"
        if type != expected:
            raise NotImplementedError(f"regularization type '{type}': the backend provides '{expected}' for this material layout "
                                      "(the one the reference's compliance selects)")
        return Form(self.ctx, "regularization")

    def tip_disp(self, u_mid, uhat, dxx):
        """0.5 int u.u J over the tagged sub-domain ``dxx`` (its index; rm_shell_pde.py:95-96)."""
        return Form(self.ctx, "tip_disp", subdomain=int(dxx))

    def area_subdomain(self, uhat, dxx):
        """int J over the tagged sub-domain ``dxx`` (rm_shell_pde.py:104-105)."""
        return Form(self.ctx, "area", subdomain=int(dxx))

    def sum_stress_subdomain(self, w, uhat, h, E, nu, dxx):
        """The six integrals of the top-surface in-plane stress components over the tagged sub-domain ``dxx``
        (rm_shell_pde.py:130-150): (sum_x, sum_y, sum_z, sum_xy, sum_xz, sum_yz), restated as ShellStressRM.inplaneStress
        writes them (linear_shell_model.py:446-458).  Values only: the reference registers these outputs in commented-out
        code (rm_shell_model.py:255-262), their partials are not implemented."""
        return tuple(Form(self.ctx, "sum_stress_" + c, subdomain=int(dxx)) for c in ("x", "y", "z", "xy", "xz", "yz"))

    def volume(self, uhat, h):
        """int h J dx (rm_shell_pde.py:98-99)."""
        return Form(self.ctx, "volume")

    def mass(self, uhat, h, rho):
        rho.bind("density")
        return Form(self.ctx, "mass")

    def elastic_energy(self, w, uhat, h, E):
        return Form(self.ctx, "elastic_energy")

    def pnorm_stress(self, w, uhat, h, E, nu, dx=None, m=1e-6, rho=100, alpha=None, regularization=False):
        """1/alpha int (m vm_top)^rho J dx with the degree-4 measure (rm_shell_pde.py:112-128); alpha is the
        reference area, evaluated by the backend on first use."""
        # regularization=True adds 0.5 * 1e3 int h^rho J dx inside the 1/alpha (rm_shell_pde.py:120-122); (m, rho, that coefficient)
        # travel with the form and reach the context when the form is evaluated: two forms with different parameters coexist
        # dx: None for the whole mesh, or the index i of a tagged sub-domain (the reference passes dxx(i))
        sel = -1 if dx is None else int(dx)
        self.ctx.set_stress_alpha(alpha, sel)          # None: the reference area, evaluated by the backend at first use (:123-127)
        return Form(self.ctx, "pnorm_stress", subdomain=sel, stress_params=(float(m), float(rho), 0.5e3 if regularization else 0.0))

    def von_Mises_stress(self, w, uhat, h, E, nu, surface="Top"):
        """von Mises stress at xi2 = h/2 ('Top'), 0 ('Mid') or -h/2 ('Bot') (rm_shell_pde.py:153-165), as a field the
        backend projects onto DG1.  The reference falls through on any other value (it builds a TypeError without raising
        it, :164) and then fails on the undefined name; here the bad argument is refused."""
        names = {"Top": "stress", "Mid": "stress_mid", "Bot": "stress_bot"}
        if surface not in names:
            raise TypeError("Unsupported surface type for stress computation.")
        return FieldForm(self.ctx, names[surface])

    # ------------------------------------------------------------------ maps
    def construct_force_to_pressure_map(self):
        """Consistent mass matrix of the pressure space VF = [CG1]^3, a = int Pv . w dx
        (rm_shell_pde.py:194-209), node-major xyz ordering like ``F_solid``."""
        if self.elementwise_pressure:
            raise NotImplementedError("force -> pressure conversion is defined for nodal pressures")
        return force_to_pressure_map(self.mesh)

    def construct_nodal_disp_map(self):
        """Sparse (3 nn x ndof) map state -> [ux; uy; uz] at the mesh vertices (rm_shell_pde.py:212-221)."""
        return nodal_disp_map(self.mesh)

    def compute_nodal_disp(self, func: Function):
        u = func.get()[: 3 * self.mesh.nn].reshape(-1, 3)
        return u[:, 0].copy(), u[:, 1].copy(), u[:, 2].copy()


def force_to_pressure_map(mesh):
    """Consistent mass matrix of [CG1]^3 on ``mesh``, node-major xyz ordering (rm_shell_pde.py:194-209;
    dynamic_rm_shell/plate_sim.py:452-468).  Host-side set-up map (3x3 Gauss on quads, 3-point rule on triangles; exact
    for the bilinear / linear basis on affine cells)."""
    m = mesh
    X = m.nodes[m.cells]
    if m.is_quad:
        g = np.array([-np.sqrt(0.6), 0.0, np.sqrt(0.6)]); w = np.array([5.0, 8.0, 5.0]) / 9.0
        P = np.array([(a, b) for a in g for b in g]); Wq = np.array([wa * wb for wa in w for wb in w])
        sx, sy = np.array([-1, 1, 1, -1.0]), np.array([-1, -1, 1, 1.0])
        N = 0.25 * (1 + sx[None] * P[:, :1]) * (1 + sy[None] * P[:, 1:])
        dN = np.stack([0.25 * sx[None] * (1 + sy[None] * P[:, 1:]), 0.25 * sy[None] * (1 + sx[None] * P[:, :1])], axis=-1)
    else:
        P = np.array([[1 / 6, 1 / 6], [2 / 3, 1 / 6], [1 / 6, 2 / 3]]); Wq = np.full(3, 1 / 6)
        N = np.stack([1 - P[:, 0] - P[:, 1], P[:, 0], P[:, 1]], axis=1)
        dN = np.broadcast_to(np.array([[-1.0, -1.0], [1.0, 0.0], [0.0, 1.0]]), (3, 3, 2))
    J = np.einsum("ebi,qbk->eqik", X, dN)
    det = np.linalg.norm(np.cross(J[..., 0], J[..., 1]), axis=-1)
    Me = np.einsum("q,eq,qa,qb->eab", Wq, det, N, N)
    nv = m.nvc
    rows = np.repeat(m.cells, nv, axis=1).ravel()
    cols = np.tile(m.cells, (1, nv)).ravel()
    Ms = sp.coo_matrix((Me.ravel(), (rows, cols)), shape=(m.nn, m.nn)).tocsr()
    return sp.kron(Ms, sp.identity(3), format="csr")


def nodal_disp_map(mesh):
    """Sparse (3 nn x ndof) map state -> [ux; uy; uz] at the mesh vertices, the stacking the reference builds from P2 basis
    evaluations (rm_shell_pde.py:212-221; dynamic_rm_shell/plate_sim.py:470-480); vertices are P2 nodes here, so every
    row holds a single 1."""
    nn, ndof = mesh.nn, mesh.ndof
    rows = np.arange(3 * nn)
    cols = (3 * np.arange(nn)[None, :] + np.arange(3)[:, None]).ravel()
    return sp.csr_matrix((np.ones(3 * nn), (rows, cols)), shape=(3 * nn, ndof))
