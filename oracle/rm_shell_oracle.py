"""CPU ORACLE -- test infrastructure only, never the product path.

A float64 numpy/scipy restatement of the reference's Reissner-Mindlin shell forward +
adjoint algorithm (femo_alpha's ``RMShellPDE`` / ``ElasticModelShapeOpt`` / ``FEA``
path), written from the formulas, element by element, with explicit strain-displacement
matrices, dense 39x39 element matrices, scipy CSR assembly and SuperLU (``splu``) as the
stand-in for PETSc + MUMPS.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.

PARITY STATUS: **parity unpinned by the reference's own tests** -- femo_alpha's
``tests/`` directory is an unmodified project template (reference
tests/test_pytest.py:5-86) and FEniCSx/PETSc are not installable here, so no reference
output exists to compare with.  The restatement is pinned instead by (tests/test_oracle*.py):
analytic Euler-Bernoulli cantilever limit printed by the reference example
(examples/advanced_examples/simple_shell_opt/ex_simple_shell_opt.py:100-105), rigid-body
/ symmetry / patch identities, the reference's own finite-difference-vs-adjoint method
(ex_simple_shell_opt.py:109-111), and an independent sympy derivation of the element
energy committed as golden vectors (tests/golden/).

What each piece follows (all paths relative to /root/reference/femo_alpha):
  * function space CG2 x CG1 ............ rm_shell/linear_shell_fenicsx/linear_shell_model.py:60-65
  * single-layer CLT A, B=0, D, A_s ...... linear_shell_model.py:136-157  (k = 0.833)
  * local basis E0,E1,E2, T .............. linear_shell_fenicsx/kinematics.py:54-91
  * F = I + grad(uhat), gradx, J ......... kinematics.py:12-44
  * strains eps, kappa, gamma ............ linear_shell_model.py:232-258
  * energies (J only on shear+drilling) .. linear_shell_model.py:275-306
  * residual, load, penalty .............. linear_shell_model.py:308-333
  * compliance, regularisation, mass ..... rm_shell/rm_shell_pde.py:64-110
  * Newton + LU solve semantics .......... fea/utils_dolfinx.py:438-468
  * adjoint operator protocol ............ csdl_alpha_opt/state_operation.py:174-220
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

SHEAR_CORRECTION = 0.833     # linear_shell_model.py:146
PENALTY_BETA = 1.0e15        # linear_shell_model.py:324
REG_ALPHA1 = 1.0e-2          # rm_shell_pde.py:67


# ----------------------------------------------------------------------------- tables
# Gauss-Legendre rules up to 7 points from decimal literals of the exact nodes and weights (computed with mpmath at 50 digits):
# the correctly rounded doubles.  numpy's leggauss is up to 4e-16 away from them, and at BASELINE config 3 (1 M DOF) a change of
# that size in the 5-point weights moves displacement / compliance / gradient by 3.5e-7 / 2.8e-7 / 3.5e-7
# (tests/golden/make_config3_golden.py with either table): the tables are part of the definition of the discrete problem at that
# size, so the checker and the HIP library each carry the exact ones.
_GL = {
    1: ([0.0], [2.0]),
    2: ([-0.5773502691896257645091488, 0.5773502691896257645091488], [1.0, 1.0]),
    3: ([-0.7745966692414833770358531, 0.0, 0.7745966692414833770358531],
        [0.5555555555555555555555556, 0.8888888888888888888888889, 0.5555555555555555555555556]),
    4: ([-0.8611363115940525752239465, -0.3399810435848562648026658, 0.3399810435848562648026658, 0.8611363115940525752239465],
        [0.3478548451374538573730639, 0.6521451548625461426269361, 0.6521451548625461426269361, 0.3478548451374538573730639]),
    5: ([-0.9061798459386639927976269, -0.5384693101056830910363144, 0.0, 0.5384693101056830910363144, 0.9061798459386639927976269],
        [0.236926885056189087514264, 0.4786286704993664680412915, 0.5688888888888888888888889, 0.4786286704993664680412915,
         0.236926885056189087514264]),
    # 6 and 7 points: only the quadrature-convergence study of the unstructured quadrilateral skin uses them (is n = 5 within 1e-8 of the
    # limit on kites?  profiles/r5_quadrature_uquad1m.txt); numpy's tables are up to 9e-16 away from these
    6: ([-0.9324695142031520278123015545, -0.661209386466264513661399595, -0.2386191860831969086305017217, 0.2386191860831969086305017217,
         0.661209386466264513661399595, 0.9324695142031520278123015545],
        [0.1713244923791703450402961422, 0.3607615730481386075698335138, 0.467913934572691047389870344, 0.467913934572691047389870344,
         0.3607615730481386075698335138, 0.1713244923791703450402961422]),
    7: ([-0.949107912342758524526189684, -0.7415311855993944398638647733, -0.4058451513773971669066064121, 0.0, 0.4058451513773971669066064121,
         0.7415311855993944398638647733, 0.949107912342758524526189684],
        [0.1294849661688696932706114327, 0.2797053914892766679014677714, 0.3818300505051189449503697755, 0.4179591836734693877551020408,
         0.3818300505051189449503697755, 0.2797053914892766679014677714, 0.1294849661688696932706114327]),
}


def gauss_legendre(n):
    if n in _GL:
        return np.array(_GL[n][0]), np.array(_GL[n][1])
    x, w = np.polynomial.legendre.leggauss(n)
    return x, w


def _lag2(t):
    """1-D quadratic Lagrange on nodes (-1, 0, 1): values and derivatives."""
    v = np.stack([0.5 * t * (t - 1.0), 1.0 - t * t, 0.5 * t * (t + 1.0)], axis=-1)
    d = np.stack([t - 0.5, -2.0 * t, t + 0.5], axis=-1)
    return v, d


def _lag1(t):
    v = np.stack([0.5 * (1.0 - t), 0.5 * (1.0 + t)], axis=-1)
    d = np.stack([-0.5 * np.ones_like(t), 0.5 * np.ones_like(t)], axis=-1)
    return v, d


# local node -> (i_xi, i_eta) index into the 1-D node sets; CCW vertices, then edge
# midpoints (edge k joins vertex k and k+1), then the centre
_Q2_IJ = [(0, 0), (2, 0), (2, 2), (0, 2), (1, 0), (2, 1), (1, 2), (0, 1), (1, 1)]
_Q1_IJ = [(0, 0), (1, 0), (1, 1), (0, 1)]


def quad_tables(pts):
    """Shape tables of the biquadratic (9) and bilinear (4) Lagrange bases at reference
    points ``pts`` (npts,2) in [-1,1]^2.  Returns N2,dN2,N1,dN1."""
    xi, et = pts[:, 0], pts[:, 1]
    a, da = _lag2(xi); b, db = _lag2(et)
    N2 = np.stack([a[:, i] * b[:, j] for i, j in _Q2_IJ], axis=1)
    dN2 = np.stack([np.stack([da[:, i] * b[:, j], a[:, i] * db[:, j]], axis=-1) for i, j in _Q2_IJ], axis=1)
    c, dc = _lag1(xi); d, dd = _lag1(et)
    N1 = np.stack([c[:, i] * d[:, j] for i, j in _Q1_IJ], axis=1)
    dN1 = np.stack([np.stack([dc[:, i] * d[:, j], c[:, i] * dd[:, j]], axis=-1) for i, j in _Q1_IJ], axis=1)
    return N2, dN2, N1, dN1


def quad_rule(n):
    x, w = gauss_legendre(n)
    X, Y = np.meshgrid(x, x, indexing="ij")
    W = np.outer(w, w)
    return np.stack([X.ravel(), Y.ravel()], axis=1), W.ravel()


# Fully symmetric rules on the unit triangle (area 1/2), orbit by orbit -- the same literals as femo_hip.hip::triangle_rule, derived
# in 60-digit arithmetic by scripts/derive_triangle_rules.py (the table is part of the discrete problem: DESIGN.md section 2).
# S21 rows (a, b = 1 - 2a, w) give the points (a,a) (b,a) (a,b); S111 rows (a, b, c = 1 - a - b, w) give (a,b) (b,a) (a,c) (c,a) (b,c) (c,b);
# the weights of a rule sum to one.
#   degree  4 ( 6 points): the reference's p-norm stress measure, quadrature_degree 4 (rm_shell_model.py:200-205)
#   degree  6 (12 points): exact for the static forms on affine cells with cell-wise polynomial data (integrand of degree <= 6)
#   degree  9 (19 points): what UFL estimates for the static forms on triangles (plain dx, linear_shell_model.py:88-103)
#   degree 12 (33 points): the convergence check beyond it
_TRI = {
    4: (None,
        [(0.4459484909159648863183293, 0.1081030181680702273633415, 0.2233815896780114656950070),
         (0.09157621350977074345957146, 0.8168475729804585130808571, 0.1099517436553218676383263)], []),
    6: (None,
        [(0.06308901449150222834033160, 0.8738219710169955433193368, 0.05084490637020681692093681),
         (0.2492867451709104212916386, 0.5014265096581791574167229, 0.1167862757263793660252896)],
        [(0.05314504984481694735324967, 0.3103524510337844054166077, 0.6365024991213986472301426, 0.08285107561837357519355346)]),
    9: (0.09713579628279883381924198,
        [(0.4896825191987376277837069, 0.02063496160252474443258615, 0.03133470022713907053685483),
         (0.4370895914929366372699304, 0.1258208170141267254601393, 0.07782754100477427931673936),
         (0.1882035356190327302409613, 0.6235929287619345395180774, 0.07964773892721025303289177),
         (0.04472951339445270986510659, 0.9105409732110945802697868, 0.02557767565869803126167880)],
        [(0.03683841205473628363481760, 0.2219629891607656956751025, 0.7411985987844980206900799, 0.04328353937728937728937729)]),
    12: (None,
         [(0.4882173897738048825646621, 0.02356522045239023487067587, 0.02573106644045533541779092),
          (0.4397243922944602729797366, 0.1205512154110794540405268, 0.04369254453803840213545726),
          (0.2712103850121159223459513, 0.4575792299757681553080973, 0.06285822421788510035427051),
          (0.1275761455415859246738963, 0.7448477089168281506522073, 0.03479611293070894298932840),
          (0.02131735045321037024685698, 0.9573652990935792595062860, 0.006166261051559017233866484)],
         [(0.1153434945346979991690112, 0.2757132696855141939747963, 0.6089432357797878068561924, 0.04037155776638092951782870),
          (0.02283833222225702961023378, 0.2813255809899395482481307, 0.6958360867878034221416355, 0.02235677320230344571183908),
          (0.02573405054833022816810924, 0.1162519159075971412413541, 0.8580140335440726305905366, 0.01731623110865889237164210)]),
}
TRI_DEGREES = tuple(sorted(_TRI))


def tri_rule(degree=6):
    """Points (npts, 2) and weights (summing to the area 1/2) of the symmetric rule of that degree: 4, 6, 9 or 12."""
    if degree not in _TRI:
        raise ValueError(f"triangles: the rule is named by its degree, one of {TRI_DEGREES} (got {degree})")
    s3, s21, s111 = _TRI[degree]
    pts, wts = [], []
    if s3 is not None:
        pts.append((1.0 / 3.0, 1.0 / 3.0)); wts.append(s3)
    for a, b, w in s21:
        pts += [(a, a), (b, a), (a, b)]; wts += [w] * 3
    for a, b, c, w in s111:
        pts += [(a, b), (b, a), (a, c), (c, a), (b, c), (c, b)]; wts += [w] * 6
    return np.array(pts), 0.5 * np.array(wts)


def degree4_rule(mesh):
    """``nquad`` of the reference's degree-4 measures (p-norm stress: rm_shell_model.py:200-205) on this mesh: 3 x 3 Gauss points on
    quadrilaterals, the 6-point rule of degree 4 on triangles."""
    return 3 if mesh.is_quad else 4


def tri_tables(pts):
    """P2 (6: 3 vertices, midpoints of edges 0-1, 1-2, 2-0) and P1 tables on the unit triangle."""
    x, y = pts[:, 0], pts[:, 1]
    L = np.stack([1 - x - y, x, y], axis=1)
    dL = np.array([[-1.0, -1.0], [1.0, 0.0], [0.0, 1.0]])
    N1 = L
    dN1 = np.broadcast_to(dL, (pts.shape[0], 3, 2)).copy()
    N2 = np.zeros((pts.shape[0], 6)); dN2 = np.zeros((pts.shape[0], 6, 2))
    for i in range(3):
        N2[:, i] = L[:, i] * (2 * L[:, i] - 1)
        dN2[:, i] = (4 * L[:, i] - 1)[:, None] * dL[i]
    for k, (i, j) in enumerate([(0, 1), (1, 2), (2, 0)]):
        N2[:, 3 + k] = 4 * L[:, i] * L[:, j]
        dN2[:, 3 + k] = 4 * (L[:, i][:, None] * dL[j] + L[:, j][:, None] * dL[i])
    return N2, dN2, N1, dN1


def _cross(a, b):
    return np.cross(a, b)


def _unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


class ShellOracle:
    """Float64 CPU restatement of the reference shell path on a ``ShellMesh``-like object
    (attributes nodes, cells, cell_p2, nV, nP2, ndof, ndof_u, is_quad)."""

    def __init__(self, mesh, element_wise_material=False, elementwise_pressure=False,
                 nquad=None, penalty_facets=None, strong_dofs=None, beta=PENALTY_BETA, rule=None, nred=0):
        self.mesh = mesh
        auto_rule = nquad is None
        if nquad is None:
            # the rule the mesh asks for: 4 x 4 Gauss on affine cells (exact there), 5 x 5 on warped quadrilaterals -- the
            # reference integrates its static forms (nearly) exactly (plain dx, linear_shell_model.py:88-103); on triangles the
            # DEGREE of the symmetric rule: 6 (exact for cell-wise polynomial data), raised to UFL's 9 by set_fields when a nodal
            # Poisson ratio that varies over the cells arrives (the same policy as ShellContext.set_field)
            nquad = mesh.recommended_nquad() if hasattr(mesh, "recommended_nquad") else (4 if mesh.is_quad else 6)
        self.nquad = int(nquad)
        self.ewm = bool(element_wise_material)
        self.ewp = bool(elementwise_pressure)
        self.beta = float(beta)
        self.penalty_facets = (np.zeros((0, 2), np.int32) if penalty_facets is None
                               else np.asarray(penalty_facets, np.int32).reshape(-1, 2))
        self.strong_dofs = (np.zeros(0, np.int32) if strong_dofs is None
                            else np.unique(np.asarray(strong_dofs, np.int32)))
        self._rule_auto = auto_rule and rule is None
        self._set_rule(self.nquad, rule, nred)
        self.npc = mesh.cell_p2.shape[1]
        self.nvc = mesh.cells.shape[1]
        self.ldof = 3 * self.npc + 3 * self.nvc
        self.hK = mesh.cell_diameters()
        self.dofs = mesh.cell_dofs()
        nT = mesh.nel if self.ewm else mesh.nn
        nF = mesh.nel if self.ewp else mesh.nn
        # defaults follow FEA.add_input init values, rm_shell_model.py:209-214
        self.h = np.full(nT, 1e-3); self.E = np.ones(nT); self.nu = np.ones(nT) * 0.3
        self.rho = np.ones(nT); self.f = np.ones((nF, 3)); self.uhat = np.zeros((mesh.nn, 3))

    def _set_rule(self, nquad, rule=None, nred=0):
        """Quadrature rule and shape tables: ``nquad`` x ``nquad`` Gauss points on quadrilaterals, the symmetric rule of DEGREE ``nquad``
        (4, 6, 9, 12) on triangles, or an explicit ``rule`` = (points, weights)."""
        mesh = self.mesh
        self.nquad, self.nred = int(nquad), int(nred)
        if mesh.is_quad:
            self.pts, self.wts = quad_rule(nquad) if rule is None else rule
            self.wts_strain = self.wts
            if nred:
                # membrane / bending / shear energies on an nred x nred rule (the dynamic path's quadrature degree 3,
                # dynamic_rm_shell/plate_sim.py:65-67,82-91), everything else on the full rule: one point list, two weights
                pr, wr = quad_rule(nred)
                self.wts_strain = np.concatenate([wr, 0 * self.wts])
                self.wts = np.concatenate([0 * wr, self.wts])
                self.pts = np.vstack([pr, self.pts])
            self.N2, self.dN2, self.N1, self.dN1 = quad_tables(self.pts)
        else:
            self.pts, self.wts = tri_rule(self.nquad) if rule is None else rule
            self.wts_strain = self.wts
            self.N2, self.dN2, self.N1, self.dN1 = tri_tables(self.pts)
        if getattr(mesh, "element", "CG2CG1") == "CG1CG1":
            # ShellElement 'CG1CG1' (linear_shell_model.py:74-79): the displacement is interpolated on the vertices as well
            self.N2, self.dN2 = self.N1, self.dN1
        # shape functions of the ROTATION: the vertex functions, or -- ShellElement 'CG2CR1' (linear_shell_model.py:68-73, triangles) --
        # the Crouzeix-Raviart functions of the edge midpoints: with barycentric coordinates L, the function of edge k (vertex k -> k + 1,
        # opposite vertex k + 2) is 1 - 2 L_(k+2): one at its own midpoint, zero at the other two, NOT continuous across edges
        self.cr = getattr(mesh, "element", "CG2CG1") == "CG2CR1"
        self.NR, self.dNR = self.N1, self.dN1
        if self.cr:
            opp = [2, 0, 1]
            self.NR = 1.0 - 2.0 * self.N1[:, opp]
            self.dNR = -2.0 * self.dN1[:, opp, :]
        self.nq = self.pts.shape[0]

    # ------------------------------------------------------------------ fields
    def set_fields(self, h=None, E=None, nu=None, rho=None, f=None, uhat=None):
        def bc(v, n):
            v = np.asarray(v, dtype=np.float64).ravel()
            return np.full(n, v[0]) if v.size == 1 else v.copy()   # utils_dolfinx.py:327-330
        nT = self.h.size
        if h is not None: self.h = bc(h, nT)
        if E is not None: self.E = bc(E, nT)
        if nu is not None:
            self.nu = bc(nu, nT)
            if self._rule_auto and not self.mesh.is_quad:
                varies = (not self.ewm) and bool(np.any(self.nu != self.nu[0]))
                if (9 if varies else 6) != self.nquad:
                    self._set_rule(9 if varies else 6)
        if rho is not None: self.rho = bc(rho, nT)
        if f is not None: self.f = np.asarray(f, np.float64).reshape(-1, 3).copy()
        if uhat is not None: self.uhat = np.asarray(uhat, np.float64).reshape(-1, 3).copy()

    def _at_qp(self, field, sl):
        """VT-field (CG1 nodal or DG0) at the quadrature points of cells ``sl``: (ne,nq)."""
        if self.ewm:
            return np.repeat(field[sl][:, None], self.nq, axis=1)
        return np.einsum("qb,eb->eq", self.N1, field[self.mesh.cells[sl]])

    # ------------------------------------------------------------------ geometry at points
    def _geometry(self, sl, N1, dN1):
        """Geometry of cells ``sl`` at reference points with tables N1,dN1 (np,nvc[,2])."""
        X = self.mesh.nodes[self.mesh.cells[sl]]                       # (ne,nvc,3)
        Jg = np.einsum("ebi,qbk->eqik", X, dN1)                        # (ne,nq,3,2)
        a = _cross(Jg[..., 0], Jg[..., 1])
        det = np.linalg.norm(a, axis=-1)
        E2 = a / det[..., None]
        E0 = _unit(Jg[..., 0])
        E1 = _cross(E2, E0)
        G = np.einsum("eqik,eqil->eqkl", Jg, Jg)
        Kinv = np.einsum("eqkl,eqil->eqki", np.linalg.inv(G), Jg)     # (ne,nq,2,3) pseudo-inverse
        # derivative of the unit normal along x_j (non-zero only on warped quads)
        W = np.zeros(Jg.shape[:2] + (3, 3))
        if self.mesh.is_quad:
            c = 0.25 * (X[:, 0] - X[:, 1] + X[:, 2] - X[:, 3])         # d2x / dxi deta
            c = np.broadcast_to(c[:, None, :], a.shape)
            da = np.stack([_cross(Jg[..., 0], c), _cross(c, Jg[..., 1])], axis=-1)   # (ne,nq,3,2)
            dn = (da - E2[..., None] * np.einsum("eqi,eqik->eqk", E2, da)[..., None, :]) / det[..., None, None]
            W = np.einsum("eqik,eqkj->eqij", dn, Kinv)
        # mesh-motion map
        U = self.uhat[self.mesh.cells[sl]]
        gradM = np.einsum("eqki,qbk->eqbi", Kinv, dN1)                 # (ne,nq,nvc,3) surface gradient of M_b
        Gu = np.einsum("ebi,eqbj->eqij", U, gradM)
        F = np.eye(3) + Gu
        Finv = np.linalg.inv(F)
        Ju = np.linalg.det(F)
        return dict(X=X, Jg=Jg, det=det, E0=E0, E1=E1, E2=E2, Kinv=Kinv, W=W, Finv=Finv, Ju=Ju,
                    gradM=gradM, F=F)

    def _B(self, sl):
        """Strain-displacement matrices (ne,nq,9,ldof) and geometry at the cell quadrature points.
        Rows: eps00, eps11, 2eps01, kap00, kap11, 2kap01, gam0, gam1, omega."""
        g = self._geometry(sl, self.N1, self.dN1)
        ne, nq, npc, nvc = g["det"].shape[0], self.nq, self.npc, self.nvc
        E0, E1, E2, Finv = g["E0"], g["E1"], g["E2"], g["Finv"]
        gradN = np.einsum("eqki,qak->eqai", g["Kinv"], self.dN2)        # (ne,nq,npc,3)
        gxN = np.einsum("eqak,eqkj->eqaj", gradN, Finv)
        gradR = np.einsum("eqki,qbk->eqbi", g["Kinv"], self.dNR)        # surface gradient of the rotation's shape functions
        gxM = np.einsum("eqbk,eqkj->eqbj", gradR, Finv)
        d = np.stack([np.einsum("eqj,eqaj->eqa", E0, gxN), np.einsum("eqj,eqaj->eqa", E1, gxN)], axis=-1)
        m = np.stack([np.einsum("eqj,eqbj->eqb", E0, gxM), np.einsum("eqj,eqbj->eqb", E1, gxM)], axis=-1)
        Wp = np.einsum("eqik,eqkj->eqij", g["W"], Finv)
        wl = [np.einsum("eqj,eqij->eqi", E0, Wp), np.einsum("eqj,eqij->eqi", E1, Wp)]   # w_J
        B = np.zeros((ne, nq, 9, self.ldof))
        Bu = B[..., : 3 * npc].reshape(ne, nq, 9, npc, 3)
        Bt = B[..., 3 * npc:].reshape(ne, nq, 9, nvc, 3)
        e0, e1, e2 = E0[:, :, None, :], E1[:, :, None, :], E2[:, :, None, :]
        d0, d1 = d[..., 0][..., None], d[..., 1][..., None]
        Bu[:, :, 0] = e0 * d0
        Bu[:, :, 1] = e1 * d1
        Bu[:, :, 2] = e0 * d1 + e1 * d0
        Bu[:, :, 6] = e2 * d0
        Bu[:, :, 7] = e2 * d1
        Bu[:, :, 8] = 0.5 * (e0 * d1 - e1 * d0)
        M = self.NR[None, :, :, None]                                   # (1,nq,nvc,1)
        m0, m1 = m[..., 0][..., None], m[..., 1][..., None]
        c00 = -e1 * m0 + M * _cross(E0, wl[0])[:, :, None, :]
        c01 = -e1 * m1 + M * _cross(E0, wl[1])[:, :, None, :]
        c10 = e0 * m0 + M * _cross(E1, wl[0])[:, :, None, :]
        c11 = e0 * m1 + M * _cross(E1, wl[1])[:, :, None, :]
        Bt[:, :, 3] = c00
        Bt[:, :, 4] = c11
        Bt[:, :, 5] = c01 + c10
        Bt[:, :, 6] = M * e1
        Bt[:, :, 7] = -M * e0
        Bt[:, :, 8] = M * e2
        return B, g

    def _C(self, sl, g, deriv=None):
        """Constitutive blocks x measure weights, (ne,nq,9,9); ``deriv`` in {None,'h','E','nu'}
        returns the derivative with respect to the (point value of the) field."""
        h, E, nu = self._at_qp(self.h, sl), self._at_qp(self.E, sl), self._at_qp(self.nu, sl)
        wdet = self.wts[None, :] * g["det"]
        wdetS = self.wts_strain[None, :] * g["det"]
        Ju = g["Ju"]
        hK2 = (self.hK[sl] ** 2)[:, None]
        k = SHEAR_CORRECTION
        one, zero = np.ones_like(nu), np.zeros_like(nu)
        P = np.stack([np.stack([one, nu, zero], -1), np.stack([nu, one, zero], -1),
                      np.stack([zero, zero, 0.5 * (1 - nu)], -1)], -2)
        dP = np.stack([np.stack([zero, one, zero], -1), np.stack([one, zero, zero], -1),
                       np.stack([zero, zero, -0.5 * one], -1)], -2)
        c = E / (1 - nu ** 2)
        G = E / 2 / (1 + nu)
        if deriv is None:
            Cm = (c * h)[..., None, None] * P
            Cb = (c * h ** 3 / 12)[..., None, None] * P
            cs = k * G * h
            cd = E * h ** 3 / hK2
        elif deriv == "h":
            Cm = c[..., None, None] * P
            Cb = (c * h ** 2 / 4)[..., None, None] * P
            cs = k * G
            cd = 3 * E * h ** 2 / hK2
        elif deriv == "E":
            Cm = (h / (1 - nu ** 2))[..., None, None] * P
            Cb = (h ** 3 / 12 / (1 - nu ** 2))[..., None, None] * P
            cs = k * h / 2 / (1 + nu)
            cd = h ** 3 / hK2
        elif deriv == "nu":
            dc = E * 2 * nu / (1 - nu ** 2) ** 2
            Cm = h[..., None, None] * (dc[..., None, None] * P + c[..., None, None] * dP)
            Cb = (h ** 3 / 12)[..., None, None] * (dc[..., None, None] * P + c[..., None, None] * dP)
            cs = -k * h * E / 2 / (1 + nu) ** 2
            cd = zero
        else:
            raise ValueError(deriv)
        C = np.zeros(wdet.shape + (9, 9))
        C[..., 0:3, 0:3] = Cm * wdetS[..., None, None]         # membrane: no J(uhat)  (:278-279)
        C[..., 3:6, 3:6] = Cb * wdetS[..., None, None]         # bending : no J(uhat)  (:281-282)
        sw = cs * Ju * wdetS
        C[..., 6, 6] = sw; C[..., 7, 7] = sw                   # shear   : J(uhat)     (:275-276)
        C[..., 8, 8] = cd * Ju * wdet                          # drilling: J(uhat)     (:284-296)
        return C

    def _chunks(self, size=2048):
        for s in range(0, self.mesh.nel, size):
            yield slice(s, min(s + size, self.mesh.nel))

    # ------------------------------------------------------------------ element & global operators
    def element_matrices(self, sl=slice(None)):
        """K_e (ne,ldof,ldof) of the elastic energy (no penalty)."""
        B, g = self._B(sl)
        C = self._C(sl, g)
        return np.einsum("eqik,eqij,eqjl->ekl", B, C, B, optimize=True)

    def _penalty_blocks(self):
        """Edge penalty matrices: list of (dofs (n,), block (n,n)) -- linear_shell_model.py:323-333."""
        out = []
        if self.penalty_facets.shape[0] == 0:
            return out
        cg1 = getattr(self.mesh, "element", "CG2CG1") == "CG1CG1"
        x, w = gauss_legendre(3)                                # degree 4 measure, utils_dolfinx.py:556
        L2, _ = _lag2(x); L1, _ = _lag1(x)
        mesh = self.mesh
        for cell, k in self.penalty_facets:
            nv = self.nvc
            va, vb = mesh.cells[cell, k], mesh.cells[cell, (k + 1) % nv]
            pa, pb = mesh.cell_p2[cell, k], mesh.cell_p2[cell, (k + 1) % nv]
            pm = None if cg1 else mesh.cell_p2[cell, nv + k]
            length = np.linalg.norm(mesh.nodes[vb] - mesh.nodes[va])
            nanson = self._nanson(cell, k, x)
            wq = w * 0.5 * length * nanson * self.beta / self.hK[cell]
            M2 = np.einsum("q,qi,qj->ij", wq, L2, L2)           # nodes (a, mid, b)
            M1 = np.einsum("q,qi,qj->ij", wq, L1, L1)           # nodes (a, b)
            for c in range(3):
                if cg1:                                   # displacement on the vertices: the linear edge functions
                    out.append((np.array([3 * pa + c, 3 * pb + c]), M1))
                else:
                    out.append((np.array([3 * pa + c, 3 * pm + c, 3 * pb + c]), M2))
                if self.cr:
                    # the trace of a Crouzeix-Raviart rotation on edge k involves all three functions of the cell: along the edge
                    # (s in [-1, 1] from vertex k to k + 1) R_k = 1, R_(k+1) = 1 - 2 L_k = s, R_(k+2) = 1 - 2 L_(k+1) = -s
                    R = np.zeros((x.size, 3))
                    R[:, k % 3], R[:, (k + 1) % 3], R[:, (k + 2) % 3] = 1.0, x, -x
                    out.append((mesh.ndof_u + 3 * mesh.cell_edges[cell].astype(np.int64) + c, np.einsum("q,qi,qj->ij", wq, R, R)))
                else:
                    out.append((mesh.ndof_u + np.array([3 * va + c, 3 * vb + c]), M1))
        return out

    def _nanson(self, cell, k, s):
        """|| J F^-T N || at edge parameter s in [-1,1] (1 when uhat = 0)."""
        if not np.any(self.uhat[self.mesh.cells[cell]]):
            return np.ones_like(s)
        if self.mesh.is_quad:
            pts = [np.stack([s, -np.ones_like(s)], 1), np.stack([np.ones_like(s), s], 1),
                   np.stack([-s, np.ones_like(s)], 1), np.stack([-np.ones_like(s), -s], 1)][k]
            _, _, N1, dN1 = quad_tables(pts)
        else:
            t = 0.5 * (s + 1)
            pts = [np.stack([t, 0 * t], 1), np.stack([1 - t, t], 1), np.stack([0 * t, 1 - t], 1)][k]
            _, _, N1, dN1 = tri_tables(pts)
        g = self._geometry(slice(cell, cell + 1), N1, dN1)
        nv = self.nvc
        tvec = _unit(self.mesh.nodes[self.mesh.cells[cell, (k + 1) % nv]] - self.mesh.nodes[self.mesh.cells[cell, k]])
        Nf = _cross(np.broadcast_to(tvec, g["E2"][0].shape), g["E2"][0])
        v = g["Ju"][0][:, None] * np.einsum("qji,qj->qi", g["Finv"][0], Nf)
        return np.linalg.norm(v, axis=-1)

    def assemble_K(self, with_penalty=True, with_strong=True):
        n = self.mesh.ndof
        rows, cols, vals = [], [], []
        for sl in self._chunks():
            Ke = self.element_matrices(sl)
            d = self.dofs[sl]
            rows.append(np.repeat(d, self.ldof, axis=1).ravel())
            cols.append(np.tile(d, (1, self.ldof)).ravel())
            vals.append(Ke.ravel())
        if with_penalty:
            for d, blk in self._penalty_blocks():
                rows.append(np.repeat(d, d.size)); cols.append(np.tile(d, d.size)); vals.append(blk.ravel())
        K = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)).tocsr()
        if with_strong and self.strong_dofs.size:
            K = self._apply_strong(K)
        return K

    def assemble_M(self):
        """Inertia of the dynamic shell: rho h (u.v + h_K^2 theta.eta) J dx (linear_shell_model.py:335-348)."""
        n = self.mesh.ndof
        rows, cols, vals = [], [], []
        for sl in self._chunks():
            g = self._geometry(sl, self.N1, self.dN1)
            c = self.wts[None, :] * g["det"] * g["Ju"] * self._at_qp(self.rho, sl) * self._at_qp(self.h, sl)
            Mu = np.einsum("eq,qa,qb->eab", c, self.N2, self.N2)
            Mt = np.einsum("eq,qa,qb->eab", c * (self.hK[sl] ** 2)[:, None], self.NR, self.NR)
            d = self.dofs[sl]
            for comp in range(3):
                du = d[:, comp:3 * self.npc:3]
                dt = d[:, 3 * self.npc + comp::3]
                rows.append(np.repeat(du, self.npc, axis=1).ravel()); cols.append(np.tile(du, (1, self.npc)).ravel()); vals.append(Mu.ravel())
                rows.append(np.repeat(dt, self.nvc, axis=1).ravel()); cols.append(np.tile(dt, (1, self.nvc)).ravel()); vals.append(Mt.ravel())
        return sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)).tocsr()

    def dynamic_history(self, f_history, dt, nsteps):
        """Midpoint / Newmark march of the reference's PlateSim (dynamic_rm_shell/plate_sim.py:131-140, 281-361):
        (2/dt^2 M + K/2) w_i = F_i + M (2/dt^2 w_{i-1} + 2/dt wdot_{i-1}) - K/2 w_{i-1}; strong BCs; zero initial state.
        Returns the (ndof, nsteps+1) history."""
        K = self.assemble_K(with_strong=False)
        M = self.assemble_M()
        a, b = 2.0 / dt ** 2, 2.0 / dt
        A = (a * M + 0.5 * K).tocsr()
        keep = np.ones(self.mesh.ndof); keep[self.strong_dofs] = 0.0
        D = sp.diags(keep)
        lu = spla.splu((D @ A @ D + sp.diags(1.0 - keep)).tocsc())
        W = np.zeros((self.mesh.ndof, nsteps + 1))
        wd = np.zeros(self.mesh.ndof)
        f0 = self.f.copy()
        for i in range(1, nsteps + 1):
            self.f = np.asarray(f_history[min(i, len(f_history) - 1)], dtype=np.float64).reshape(-1, 3)
            rhs = self.load_vector() + keep * (M @ (a * W[:, i - 1] + b * wd) - 0.5 * (K @ W[:, i - 1]))
            rhs[self.strong_dofs] = 0.0
            W[:, i] = lu.solve(rhs)
            wd = b * (W[:, i] - W[:, i - 1]) - wd
        self.f = f0
        return W

    def _apply_strong(self, K):
        """Zero BC rows/columns, unit diagonal -- dolfinx assemble_matrix(bcs) semantics
        reached from assembleSystem (utils_dolfinx.py:208-221)."""
        keep = np.ones(K.shape[0]); keep[self.strong_dofs] = 0.0
        D = sp.diags(keep)
        return (D @ K @ D + sp.diags(1.0 - keep)).tocsr()

    def set_dirichlet_values(self, g=None):
        """Prescribed state g of the penalty term beta/h_E |..| (w - g).v (linear_shell_model.py:323-333); None = zero."""
        self.g_dirichlet = None if g is None else np.asarray(g, dtype=np.float64).copy()

    def load_vector(self):
        """F_a = int f . N_a J dx (linear_shell_model.py:320), plus P g when the penalty term carries prescribed values."""
        Fv = np.zeros(self.mesh.ndof)
        if getattr(self, "g_dirichlet", None) is not None:
            for d, blk in self._penalty_blocks():
                Fv[d] += blk @ self.g_dirichlet[d]
        for sl in self._chunks():
            g = self._geometry(sl, self.N1, self.dN1)
            wj = self.wts[None, :] * g["det"] * g["Ju"]
            if self.ewp:
                fq = np.repeat(self.f[sl][:, None, :], self.nq, axis=1)
            else:
                fq = np.einsum("qb,ebc->eqc", self.N1, self.f[self.mesh.cells[sl]])
            Fe = np.einsum("eq,qa,eqc->eac", wj, self.N2, fq).reshape(fq.shape[0], -1)
            np.add.at(Fv, self.dofs[sl][:, : 3 * self.npc].ravel(), Fe.ravel())
        if self.strong_dofs.size:
            Fv[self.strong_dofs] = 0.0
        return Fv

    def apply_K(self, x, with_penalty=True):
        """y = K x element by element (no global matrix)."""
        y = np.zeros(self.mesh.ndof)
        for sl in self._chunks():
            Ke = self.element_matrices(sl)
            ye = np.einsum("eij,ej->ei", Ke, x[self.dofs[sl]])
            np.add.at(y, self.dofs[sl].ravel(), ye.ravel())
        if with_penalty:
            for d, blk in self._penalty_blocks():
                y[d] += blk @ x[d]
        return y

    def residual(self, w):
        """R(w) = K w + penalty(w) - F   (linear_shell_model.py:317-321)."""
        return self.apply_K(w) - self.load_vector()

    def factorize(self):
        self._K = self.assemble_K()
        # symmetric-mode SuperLU with minimum-degree ordering on A+A^T: 3x faster than the COLAMD
        # default on this matrix -- the 'best CPU effort' baseline of BASELINE.md section 3
        self._lu = spla.splu(self._K.tocsc(), permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0,
                             options=dict(SymmetricMode=True))
        return self._lu

    def solve(self):
        """Forward solve: Newton on a linear residual == one LU solve (the reference's
        remaining two Newton iterations are refinement no-ops, utils_dolfinx.py:438-468)."""
        lu = self.factorize()
        b = self.load_vector()
        w = lu.solve(b)
        w += lu.solve(b - self._K @ w)          # what Newton iteration 2 does
        return w

    def solve_adjoint(self, rhs):
        """lambda = A^-1 rhs with the stored factorisation, BC rows zeroed
        (state_operation.py:211-218; A symmetric so A == A^T, quirk Q3)."""
        rhs = np.array(rhs, dtype=np.float64)
        lam = self._lu.solve(rhs)
        lam += self._lu.solve(rhs - self._K @ lam)
        if self.strong_dofs.size:
            lam[self.strong_dofs] = 0.0
        return lam

    # ------------------------------------------------------------------ outputs
    def _u_at_qp(self, w, sl):
        ue = w[self.dofs[sl][:, : 3 * self.npc]].reshape(-1, self.npc, 3)
        return np.einsum("qa,eac->eqc", self.N2, ue)

    def regularization(self):
        val = 0.0
        for sl in self._chunks():
            g = self._geometry(sl, self.N1, self.dN1)
            wd = self.wts[None, :] * g["det"]
            if self.ewm:
                val += 0.5 * REG_ALPHA1 * np.sum(self.h[sl] ** 2 * wd.sum(axis=1))
            else:
                gh = np.einsum("eqbi,eb->eqi", g["gradM"], self.h[self.mesh.cells[sl]])
                val += 0.5 * REG_ALPHA1 * np.sum(wd * np.einsum("eqi,eqi->eq", gh, gh))
        return val

    def compliance(self, w):
        """int u.u J dx + regularisation(h)  (rm_shell_pde.py:85-89)."""
        val = 0.0
        for sl in self._chunks():
            g = self._geometry(sl, self.N1, self.dN1)
            u = self._u_at_qp(w, sl)
            val += np.sum(self.wts[None, :] * g["det"] * g["Ju"] * np.einsum("eqc,eqc->eq", u, u))
        return val + self.regularization()

    def mass(self):
        """int rho h J dx (rm_shell_pde.py:101-102)."""
        val = 0.0
        for sl in self._chunks():
            g = self._geometry(sl, self.N1, self.dN1)
            val += np.sum(self.wts[None, :] * g["det"] * g["Ju"] * self._at_qp(self.rho, sl) * self._at_qp(self.h, sl))
        return val

    def elastic_energy(self, w):
        return 0.5 * float(w @ self.apply_K(w, with_penalty=False))

    def dcompliance_du(self, w):
        out = np.zeros(self.mesh.ndof)
        for sl in self._chunks():
            g = self._geometry(sl, self.N1, self.dN1)
            u = self._u_at_qp(w, sl)
            ge = 2.0 * np.einsum("eq,qa,eqc->eac", self.wts[None, :] * g["det"] * g["Ju"], self.N2, u)
            np.add.at(out, self.dofs[sl][:, : 3 * self.npc].ravel(), ge.ravel())
        return out

    def dcompliance_dh(self, w=None):
        out = np.zeros(self.h.size)
        for sl in self._chunks():
            g = self._geometry(sl, self.N1, self.dN1)
            wd = self.wts[None, :] * g["det"]
            if self.ewm:
                out[sl] += REG_ALPHA1 * self.h[sl] * wd.sum(axis=1)
            else:
                gh = np.einsum("eqbi,eb->eqi", g["gradM"], self.h[self.mesh.cells[sl]])
                ge = REG_ALPHA1 * np.einsum("eq,eqbi,eqi->eb", wd, g["gradM"], gh)
                np.add.at(out, self.mesh.cells[sl].ravel(), ge.ravel())
        return out

    def dmass_dh(self):
        out = np.zeros(self.h.size)
        for sl in self._chunks():
            g = self._geometry(sl, self.N1, self.dN1)
            wj = self.wts[None, :] * g["det"] * g["Ju"] * self._at_qp(self.rho, sl)
            if self.ewm:
                out[sl] += wj.sum(axis=1)
            else:
                np.add.at(out, self.mesh.cells[sl].ravel(), np.einsum("eq,qb->eb", wj, self.N1).ravel())
        return out

    # ------------------------------------------------------------------ stress outputs
    def von_mises_top(self, w, sl=slice(None), zf=0.5, components=False):
        """von Mises stress at xi2 = zf h (zf = 1/2 top, 0 mid, -1/2 bottom surface; rm_shell_pde.py:153-165) at the
        quadrature points of this oracle's rule, (ne,nq), following ShellStressRM (linear_shell_model.py:393-467) with
        xi2 a *field* (rm_shell_pde.py:117-119): eps = eps_m - zf h kappa - zf sym((E2 x theta)_loc (x) gradx(h)_loc).
        ``components``: also return the local stresses (s0, s1, s2)."""
        B, g = self._B(sl)
        we = w[self.dofs[sl]]
        s = np.einsum("eqij,ej->eqi", B, we)
        h, E, nu = self._at_qp(self.h, sl), self._at_qp(self.E, sl), self._at_qp(self.nu, sl)
        th = np.einsum("qb,ebc->eqc", self.NR, we[:, 3 * self.npc:].reshape(-1, self.nvc, 3))
        b0 = -np.einsum("eqc,eqc->eq", th, g["E1"]); b1 = np.einsum("eqc,eqc->eq", th, g["E0"])
        if self.ewm:
            gh0 = gh1 = np.zeros_like(h)
        else:
            gxM = np.einsum("eqbk,eqkj->eqbj", g["gradM"], g["Finv"])
            hn = self.h[self.mesh.cells[sl]]
            gh = np.einsum("eb,eqbj->eqj", hn, gxM)
            gh0 = np.einsum("eqj,eqj->eq", g["E0"], gh); gh1 = np.einsum("eqj,eqj->eq", g["E1"], gh)
        e0 = s[..., 0] - zf * h * s[..., 3] - zf * b0 * gh0
        e1 = s[..., 1] - zf * h * s[..., 4] - zf * b1 * gh1
        gg = s[..., 2] - zf * h * s[..., 5] - zf * (b0 * gh1 + b1 * gh0)
        c = E / (1 - nu ** 2)
        s0, s1, s2 = c * (e0 + nu * e1), c * (nu * e0 + e1), c * 0.5 * (1 - nu) * gg
        vm = np.sqrt(s0 ** 2 - s0 * s1 + s1 ** 2 + 3 * s2 ** 2)
        return (vm, g, (s0, s1, s2)) if components else (vm, g)

    def pnorm_stress(self, w, m=1e-6, rho=100, alpha=None, cells=None, regularization=False):
        """1/alpha int (m vm)^rho J dx (rm_shell_pde.py:112-128); use an oracle built with nquad=3 for the
        reference's degree-4 measure (rm_shell_model.py:200-205).  alpha defaults to the reference area.
        ``cells``: restrict the measure to a sub-domain (the reference's dxx(i), rm_shell_model.py:242-253).
        ``regularization``: + 0.5 * 1e3 int h^rho J dx inside the 1/alpha (:120-122)."""
        val, area = 0.0, 0.0
        sel = None if cells is None else np.isin(np.arange(self.mesh.nel), np.asarray(cells))
        for sl in self._chunks():
            vm, g = self.von_mises_top(w, sl)
            wd = self.wts[None, :] * g["det"]
            if sel is not None:
                wd = wd * sel[sl][:, None]
            val += np.sum(wd * g["Ju"] * (m * vm) ** rho)
            if regularization:
                val += np.sum(wd * g["Ju"] * 0.5e3 * self._at_qp(self.h, sl) ** rho)
            area += np.sum(wd)
        return val / (area if alpha is None else alpha)

    def sum_stress_subdomain(self, w, cells=None):
        """(sum_x, sum_y, sum_z, sum_xy, sum_xz, sum_yz) = int sigma_ij J dx over ``cells`` of the top-surface in-plane
        stress (rm_shell_pde.py:130-150), with sigma exactly as ShellStressRM.inplaneStress writes it
        (linear_shell_model.py:446-458): sigma_ij = sum_kl E012[i,k] s3d[k,l] E012[j,l], E012[i,k] = component k of the
        i-th local basis vector, s3d = [[s0, s2, 0], [s2, s1, 0], [0, 0, 0]] -- only the x and y components of the basis
        vectors enter (restated as written, not "corrected")."""
        out = np.zeros(6)
        sel = None if cells is None else np.isin(np.arange(self.mesh.nel), np.asarray(cells))
        for sl in self._chunks():
            vm, g, (s0, s1, s2) = self.von_mises_top(w, sl, components=True)
            wj = self.wts[None, :] * g["det"] * g["Ju"]
            if sel is not None:
                wj = wj * sel[sl][:, None]
            Eb = [g["E0"], g["E1"], g["E2"]]
            comp = lambda i, j: (Eb[i][..., 0] * (s0 * Eb[j][..., 0] + s2 * Eb[j][..., 1])
                                 + Eb[i][..., 1] * (s2 * Eb[j][..., 0] + s1 * Eb[j][..., 1]))
            for k, (i, j) in enumerate(((0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2))):
                out[k] += np.sum(wj * comp(i, j))
        return out

    def stress_dg1(self, w, surface="Top"):
        """L2 projection of the von Mises stress on the 'Top' | 'Mid' | 'Bot' surface onto DG1, (nel, nvc)
        (rm_shell_pde.py:153-165, utils_dolfinx.py:568-602)."""
        zf = {"Top": 0.5, "Mid": 0.0, "Bot": -0.5}[surface]
        out = np.zeros((self.mesh.nel, self.nvc))
        for sl in self._chunks():
            vm, g = self.von_mises_top(w, sl, zf=zf)
            wd = self.wts[None, :] * g["det"]
            M = np.einsum("eq,qi,qj->eij", wd, self.N1, self.N1)
            b = np.einsum("eq,qi,eq->ei", wd, self.N1, vm)
            out[sl] = np.linalg.solve(M, b[..., None])[..., 0]
        return out

    # ------------------------------------------------------------------ (dR/d arg)^T lambda
    def dRdfield_T(self, name, w, lam):
        """(dR/d field)^T lam for field in {'h','E','nu'} -- what
        ``computeMatVecProductBwd(dRdf, lambda)`` returns (state_operation.py:180-184)."""
        out = np.zeros(self.h.size)
        for sl in self._chunks():
            B, g = self._B(sl)
            dC = self._C(sl, g, deriv=name)
            sw = np.einsum("eqij,ej->eqi", B, w[self.dofs[sl]])
            sl_ = np.einsum("eqij,ej->eqi", B, lam[self.dofs[sl]])
            dens = np.einsum("eqi,eqij,eqj->eq", sl_, dC, sw)
            if self.ewm:
                out[sl] += dens.sum(axis=1)
            else:
                np.add.at(out, self.mesh.cells[sl].ravel(), np.einsum("eq,qb->eb", dens, self.N1).ravel())
        return out

    def dRdf_T(self, lam):
        """(dR/df)^T lam = - int M_b (lam_u) J dx, node-major xyz like ``F_solid``."""
        out = np.zeros_like(self.f)
        for sl in self._chunks():
            g = self._geometry(sl, self.N1, self.dN1)
            wj = self.wts[None, :] * g["det"] * g["Ju"]
            lu = self._u_at_qp(lam, sl)
            if self.ewp:
                out[sl] -= np.einsum("eq,eqc->ec", wj, lu)
            else:
                ge = -np.einsum("eq,qb,eqc->ebc", wj, self.N1, lu)
                np.add.at(out, self.mesh.cells[sl].ravel(), ge.reshape(-1, 3))
        return out.ravel()

    # ------------------------------------------------------------------ the parity triple
    def forward_adjoint(self):
        """displacement, compliance, d compliance / d thickness (total derivative)."""
        w = self.solve()
        J = self.compliance(w)
        lam = self.solve_adjoint(self.dcompliance_du(w))
        dJdh = self.dcompliance_dh(w) - self.dRdfield_T("h", w, lam)
        return w, J, dJdh
