"""Generate tests/golden/sympy_triangle.npz -- the independent symbolic derivation of tests/golden/make_sympy_golden.py
for the element variants it does not cover: the TRIANGLES with 'CG2CG1' (P2 displacement, P1 rotation), 'CG2CR1'
(P2 displacement, Crouzeix-Raviart rotation on the edge midpoints) and 'CG1CG1' (P1 / P1), and the 'CG1CG1' QUADRILATERAL
(reference femo_alpha/rm_shell/linear_shell_fenicsx/linear_shell_model.py:60-80).
It pins those branches of the CPU oracle (oracle/rm_shell_oracle.py), which until round 5 were held by structural
identities only.

As in the quadrilateral script the reference's UFL text is followed operation by operation with sympy (surface gradient =
d/dxi * pseudo-inverse Jacobian, CellNormal, cross products, gradx = grad . inv(F), local frame, Voigt strains, energy
densities; kinematics.py:12-106, linear_shell_model.py:136-157, 199-306) -- no B matrix is formed by hand and no code is
shared with the oracle.

Case T: an affine triangle tilted in 3-D, nodal thickness (affine over the cell), uniform E / nu, uhat = 0: exact (rational)
integration of the element stiffness for the three spaces (27 x 27, 27 x 27, 18 x 18), and of the load vector.
Case Q: the affine quadrilateral of the other script's case A with the CG1CG1 space: 24 x 24.
Case S: the von Mises stress of ShellStressRM at the top, middle and bottom surface (thickness a field) on a warped quadrilateral
        with uhat != 0, at three points, for a given state.
Case W: the warped quadrilateral with uhat != 0 and nodal h / E / nu integrated with the 5 x 5 Gauss rule from the symbolic point values
        (element matrix and load vector): what a kernel using that rule must reproduce.
Case P: the penalty blocks of that quadrilateral's four facets with uhat != 0 (three-point facet rule).
Case PC: the penalty blocks of the affine triangle for CG2CR1 (the rotation's trace involves all three Crouzeix-Raviart functions).
Case W2: the shape sensitivity D . (dR/duhat)^T lam on that cell for three directions D, as a 50-digit central difference of the symbolic
        point values (nothing differentiated by hand).
Case N: the facet factor || J F^-T N || of the penalty term (Nanson's formula, linear_shell_model.py:323-333) with uhat != 0, at
        three points of every facet of a warped quadrilateral and of a triangle.

Local numbering (the oracle's and the library's): displacement nodes = vertices 0, 1, 2, then the midpoints of the edges
0-1, 1-2, 2-0; rotation nodes = the vertices (CG1) or those three midpoints (CR).  DOF 3 a + c, then 18 + 3 b + c.

Run:  python tests/golden/make_sympy_golden_tri.py   (about ten minutes; --keep-W keeps the committed values of case W, the longest)
"""
import itertools
import os

import numpy as np
import sympy as sm

xi, eta = sm.symbols("xi eta")
K_SHEAR = sm.Rational(833, 1000)
LAM = [1 - xi - eta, xi, eta]                                    # barycentric coordinates of the unit triangle
EDGES = [(0, 1), (1, 2), (2, 0)]
N2 = [l * (2 * l - 1) for l in LAM] + [4 * LAM[i] * LAM[j] for i, j in EDGES]
N1 = list(LAM)
NCR = [1 - 2 * LAM[(k + 2) % 3] for k in range(3)]               # one on its own edge midpoint, zero on the other two
Q1 = [(1 - xi) * (1 - eta) / 4, (1 + xi) * (1 - eta) / 4, (1 + xi) * (1 + eta) / 4, (1 - xi) * (1 + eta) / 4]   # on [-1, 1]^2


def vec(fn, coefs):
    return sm.Matrix([sum(fn[b] * coefs[b][c] for b in range(len(fn))) for c in range(3)])


def build(X, hn, E, nu, NGeo, NDisp, NRot):
    """NGeo: vertex functions of the cell (geometry, thickness); NDisp / NRot: the displacement / rotation spaces."""
    x = vec(NGeo, X)
    Jg = x.jacobian([xi, eta])
    a = Jg[:, 0].cross(Jg[:, 1])
    detg = sm.sqrt(a.dot(a))
    n = a / detg                                                 # CellNormal
    Kinv = (Jg.T * Jg).inv() * Jg.T                              # pseudo-inverse (2 x 3)

    def grad(v):                                                 # UFL grad on the manifold; uhat = 0: gradx == grad
        return v.jacobian([xi, eta]) * Kinv

    A0 = Jg[:, 0]
    E0 = A0 / sm.sqrt(A0.dot(A0))                                # kinematics.py:66-67
    E1 = n.cross(E0)                                             # kinematics.py:68
    T = sm.Matrix([E0.T, E1.T])                                  # kinematics.py:79-80
    h = sum(NGeo[b] * hn[b] for b in range(len(NGeo)))
    hK = max(sm.sqrt(sum((X[i][c] - X[j][c]) ** 2 for c in range(3))) for i, j in itertools.combinations(range(len(NGeo)), 2))

    def strains(U, TH):
        u = vec(NDisp, U)
        th = vec(NRot, TH)
        gradu = grad(u)                                          # linear_shell_model.py:220
        t_gu = T * gradu * T.T                                   # :222, kinematics.py:90-91
        eps = (t_gu + t_gu.T) / 2                                # :238
        gb = T * grad(n.cross(th)) * T.T                         # :242
        kap = (gb + gb.T) / 2
        gam = T * (-(n.cross(th))) + T * (gradu.T * n)           # :252-257
        om = (t_gu[0, 1] - t_gu[1, 0]) / 2 + th.dot(n)           # :288-289
        return [eps[0, 0], eps[1, 1], 2 * eps[0, 1], kap[0, 0], kap[1, 1], 2 * kap[0, 1], gam[0], gam[1], om]

    Cp = (E / (1 - nu * nu)) * sm.Matrix([[1, nu, 0], [nu, 1, 0], [0, 0, (1 - nu) / 2]])
    G = E / 2 / (1 + nu)
    C = sm.zeros(9, 9)
    C[0:3, 0:3] = h * Cp                                         # linear_shell_model.py:136-157
    C[3:6, 3:6] = h ** 3 / 12 * Cp
    C[6, 6] = C[7, 7] = K_SHEAR * G * h
    C[8, 8] = E * h ** 3 / hK ** 2                               # drilling, :284-296
    return strains, sm.simplify(C), sm.simplify(detg)


def tri_int(p):
    """Exact integral over the unit triangle of a polynomial in xi, eta:  int xi^i eta^j = i! j! / (i + j + 2)!"""
    P = sm.Poly(sm.expand(p), xi, eta)
    return sum(c * sm.factorial(i) * sm.factorial(j) / sm.factorial(i + j + 2) for (i, j), c in P.terms())


def quad_int(p):
    """Exact integral over [-1, 1]^2 of a polynomial in xi, eta."""
    P = sm.Poly(sm.expand(p), xi, eta)
    return sum(c * sm.Rational(2, i + 1) * sm.Rational(2, j + 1) for (i, j), c in P.terms() if i % 2 == 0 and j % 2 == 0)


def unit_dofs(nd, nr):
    for i in range(3 * (nd + nr)):
        U = [[0] * 3 for _ in range(nd)]
        TH = [[0] * 3 for _ in range(nr)]
        if i < 3 * nd:
            U[i // 3][i % 3] = 1
        else:
            TH[(i - 3 * nd) // 3][(i - 3 * nd) % 3] = 1
        yield U, TH


def stiffness(X, hn, E, nu, NGeo, NDisp, NRot, integrate):
    strains, C, detg = build(X, hn, E, nu, NGeo, NDisp, NRot)
    n = 3 * (len(NDisp) + len(NRot))
    Bs = [[sm.expand(sm.simplify(e)) for e in strains(U, TH)] for U, TH in unit_dofs(len(NDisp), len(NRot))]
    CB = [[sm.expand(sum(C[k, l] * Bs[j][l] for l in range(9))) for k in range(9)] for j in range(n)]
    Ke = np.zeros((n, n))
    for i in range(n):
        for j in range(i, n):
            Ke[i, j] = Ke[j, i] = float(integrate(sum(Bs[i][k] * CB[j][k] for k in range(9)) * detg))
    return Ke, detg


def case_T():
    R = sm.Rational
    x0 = [R(1, 10), R(-1, 5), R(3, 10)]
    d0 = [R(6, 5), 0, R(8, 5)]                                   # x1 - x0: length 2 (the cell diameter)
    d1 = [R(3, 5), R(4, 5), R(4, 5)]                             # x2 - x0; |d0 x d1| = 8 / 5
    X = [x0, [x0[c] + d0[c] for c in range(3)], [x0[c] + d1[c] for c in range(3)]]
    hn = [R(1, 10), R(3, 25), R(2, 25)]
    E, nu = R(7, 2), R(3, 10)
    fn = [[R(1), R(-2), R(1, 2)], [R(1, 3), R(0), R(2)], [R(-1), R(1), R(1)]]
    Ke_cg, detg = stiffness(X, hn, E, nu, N1, N2, N1, tri_int)
    print("CG2CG1 triangle done")
    Ke_cr, _ = stiffness(X, hn, E, nu, N1, N2, NCR, tri_int)
    print("CG2CR1 triangle done")
    Ke_11, _ = stiffness(X, hn, E, nu, N1, N1, N1, tri_int)
    print("CG1CG1 triangle done")
    f = vec(N1, fn)
    Fe = np.zeros(18)
    for a in range(6):
        for c in range(3):
            Fe[3 * a + c] = float(tri_int(N2[a] * f[c] * detg))
    return dict(T_X=np.array(X, float), T_h=np.array(hn, float), T_E=np.array([float(E)]), T_nu=np.array([float(nu)]),
                T_f=np.array(fn, float), T_Ke_cg2cg1=Ke_cg, T_Ke_cg2cr1=Ke_cr, T_Ke_cg1cg1=Ke_11, T_Fe=Fe)


def case_Q():
    R = sm.Rational                                              # the parallelogram of make_sympy_golden.py, case A
    x0 = [R(1, 10), R(-1, 5), R(3, 10)]
    d0 = [R(6, 5), 0, R(8, 5)]
    d1 = [R(3, 5), R(4, 5), R(4, 5)]
    X = [x0, [x0[c] + d0[c] for c in range(3)], [x0[c] + d0[c] + d1[c] for c in range(3)], [x0[c] + d1[c] for c in range(3)]]
    hn = [R(1, 10), R(3, 25), R(2, 25), R(11, 100)]
    E, nu = R(7, 2), R(3, 10)
    Ke, _ = stiffness(X, hn, E, nu, Q1, Q1, Q1, quad_int)
    print("CG1CG1 quadrilateral done")
    return dict(Q_X=np.array(X, float), Q_h=np.array(hn, float), Q_E=np.array([float(E)]), Q_nu=np.array([float(nu)]), Q_Ke_cg1cg1=Ke)


def nanson(X, Uhat, NGeo, edges_ref, params):
    """|| J(uhat) F(uhat)^-T N || on the facets (linear_shell_model.py:323-333: Nanson's formula for the facet measure), with
    N = FacetNormal of a manifold cell: the reference facet normal pushed forward by the pseudo-inverse Jacobian and normalised.
    ``edges_ref``: per local facet (point(s) -> (xi, eta), reference outward normal).  Returns values [facet][parameter]."""
    x = vec(NGeo, X)
    Jg = x.jacobian([xi, eta])
    Kinv = (Jg.T * Jg).inv() * Jg.T

    def grad(v):
        return v.jacobian([xi, eta]) * Kinv

    F = sm.eye(3) + grad(vec(NGeo, Uhat))                        # kinematics.py:42-44
    out = []
    for point, nref in edges_ref:
        N = Kinv.T * sm.Matrix(nref)
        row = []
        for sv in params:
            px, py = point(sv)
            sub = {xi: px, eta: py}
            Fn = F.subs(sub)
            Nn = N.subs(sub)
            Nn = Nn / sm.sqrt(Nn.dot(Nn))
            v = Fn.det() * (Fn.inv().T * Nn)
            row.append(float(sm.sqrt(v.dot(v)).evalf(30)))
        out.append(row)
    return np.array(out)


def case_N():
    R = sm.Rational
    params = [R(-3, 5), R(1, 10), R(4, 5)]                       # edge parameter s in [-1, 1], from local vertex k to k + 1
    Xq = [[R(0), R(0), R(0)], [R(1), R(1, 10), R(1, 5)], [R(6, 5), R(9, 10), R(-1, 10)], [R(-1, 10), R(1), R(3, 10)]]   # case B's warped quad
    Uq = [[R(1, 50), R(-1, 100), R(3, 100)], [R(-1, 50), R(1, 40), R(0)], [R(1, 100), R(1, 100), R(-1, 50)], [R(0), R(-3, 100), R(1, 100)]]
    quad_edges = [(lambda s: (s, -1), (0, -1)), (lambda s: (1, s), (1, 0)), (lambda s: (-s, 1), (0, 1)), (lambda s: (-1, -s), (-1, 0))]
    Xt = [[R(1, 10), R(-1, 5), R(3, 10)], [R(13, 10), R(-1, 5), R(19, 10)], [R(7, 10), R(3, 5), R(11, 10)]]              # case T's triangle
    Ut = [[R(1, 50), R(-1, 100), R(3, 100)], [R(-1, 50), R(1, 40), R(0)], [R(1, 100), R(1, 100), R(-1, 50)]]
    half = lambda s: (s + 1) / 2
    tri_edges = [(lambda s: (half(s), 0), (0, -1)), (lambda s: (1 - half(s), half(s)), (1, 1)), (lambda s: (0, 1 - half(s)), (-1, 0))]
    out = dict(N_s=np.array(params, float), N_quad_X=np.array(Xq, float), N_quad_uhat=np.array(Uq, float),
               N_quad=nanson(Xq, Uq, Q1, quad_edges, params),
               N_tri_X=np.array(Xt, float), N_tri_uhat=np.array(Ut, float), N_tri=nanson(Xt, Ut, N1, tri_edges, params))
    print("Nanson factors done")
    return out


Q2_IJ = [(0, 0), (2, 0), (2, 2), (0, 2), (1, 0), (2, 1), (1, 2), (0, 1), (1, 1)]   # local P2 nodes of a quadrilateral: vertices, edge midpoints, centre


def _lag(nodes, t):
    out = []
    for i, a in enumerate(nodes):
        p = sm.Integer(1)
        for j, b in enumerate(nodes):
            if i != j:
                p *= (t - b) / (a - b)
        out.append(sm.expand(p))
    return out


def case_S():
    """von Mises stress of ShellStressRM (linear_shell_model.py:350-467) at xi2 = zf h(x) -- the through-thickness coordinate is a FIELD
    (rm_shell_pde.py:117-119, 153-165), so gradx differentiates it too -- on the warped quadrilateral with uhat != 0, nodal h / E / nu and
    a given state, at three points and zf = 1/2, 0, -1/2."""
    R = sm.Rational
    X = [[R(0), R(0), R(0)], [R(1), R(1, 10), R(1, 5)], [R(6, 5), R(9, 10), R(-1, 10)], [R(-1, 10), R(1), R(3, 10)]]
    Uhat = [[R(1, 50), R(-1, 100), R(3, 100)], [R(-1, 50), R(1, 40), R(0)], [R(1, 100), R(1, 100), R(-1, 50)], [R(0), R(-3, 100), R(1, 100)]]
    hn = [R(1, 20), R(3, 50), R(1, 25), R(11, 200)]
    En = [R(2), R(5, 2), R(9, 4), R(3)]
    nun = [R(3, 10), R(1, 4), R(7, 20), R(1, 5)]
    rs = np.random.default_rng(7)
    U = [[R(int(v), 1000) for v in row] for row in rs.integers(-40, 40, (9, 3))]
    TH = [[R(int(v), 1000) for v in row] for row in rs.integers(-60, 60, (4, 3))]
    pts = [(R(-3, 5), R(1, 4)), (R(1, 3), R(-7, 10)), (R(4, 5), R(9, 10))]
    L2x, L2y = _lag([-1, 0, 1], xi), _lag([-1, 0, 1], eta)
    NQ2 = [L2x[i] * L2y[j] for i, j in Q2_IJ]
    x = vec(Q1, X)
    Jg = x.jacobian([xi, eta])
    a = Jg[:, 0].cross(Jg[:, 1])
    E2 = a / sm.sqrt(a.dot(a))                                   # CellNormal
    Kinv = (Jg.T * Jg).inv() * Jg.T
    grad = lambda v: v.jacobian([xi, eta]) * Kinv
    Finv = (sm.eye(3) + grad(vec(Q1, Uhat))).inv()
    gradx = lambda v: grad(v) * Finv                             # kinematics.py:21
    A0 = Jg[:, 0]
    E0 = A0 / sm.sqrt(A0.dot(A0))
    E1 = E2.cross(E0)
    E01 = sm.Matrix([E0.T, E1.T])
    h = sum(Q1[b] * hn[b] for b in range(4)); E = sum(Q1[b] * En[b] for b in range(4)); nu = sum(Q1[b] * nun[b] for b in range(4))
    u_mid, theta = vec(NQ2, U), vec(Q1, TH)
    out = np.zeros((3, len(pts)))
    for iz, zf in enumerate((R(1, 2), R(0), R(-1, 2))):
        u = u_mid - (zf * h) * E2.cross(theta)                   # ShellStressRM.u, :393-398, with xi2 = zf h
        gl = E01 * gradx(u) * E01.T                              # gradu_local, :401-409
        em = (gl + gl.T) / 2
        eps = sm.Matrix([em[0, 0], em[1, 1], 2 * em[0, 1]])      # :412-420
        D = (E / (1 - nu * nu)) * sm.Matrix([[1, nu, 0], [nu, 1, 0], [0, 0, (1 - nu) / 2]])
        sg = D * eps                                             # cauchyStresses, :433-442
        vm = sm.sqrt(sg[0] ** 2 - sg[0] * sg[1] + sg[1] ** 2 + 3 * sg[2] ** 2)      # :459-467
        for ip, (px, py) in enumerate(pts):
            out[iz, ip] = float(vm.subs({xi: px, eta: py}).evalf(30))
        if iz == 0:
            # the p-norm aggregate of the top surface, 1 / alpha int (m vm)^rho J dx (rm_shell_pde.py:112-128), with the 3 x 3 Gauss rule of
            # the reference's degree-4 measure (rm_shell_model.py:200-201), m = 2, rho = 4 and alpha = 1 given by the caller
            detg_ = sm.sqrt(a.dot(a))
            Ju_ = (sm.eye(3) + grad(vec(Q1, Uhat))).det()
            import mpmath as mp
            mp.mp.dps = 40
            g3 = [-mp.sqrt(mp.mpf(3) / 5), mp.mpf(0), mp.sqrt(mp.mpf(3) / 5)]; w3 = [mp.mpf(5) / 9, mp.mpf(8) / 9, mp.mpf(5) / 9]
            fdens = sm.lambdify((xi, eta), (2 * vm) ** 4 * Ju_ * detg_, "mpmath")       # (numeric points: substituting sqrt(3/5) symbolically does not end)
            pnorm = float(sum(w3[i] * w3[j] * fdens(g3[i], g3[j]) for i in range(3) for j in range(3)))
            # int sigma_ij J dx of the top-surface in-plane stress in GLOBAL coordinates (ShellStressRM.inplaneStress, :444-457;
            # sum_stress_subdomain, rm_shell_pde.py:130-150), components xx, yy, zz, xy, xz, yz, with the 5 x 5 Gauss rule
            E012 = sm.Matrix([E0.T, E1.T, E2.T])
            s3 = sm.Matrix([[sg[0], sg[2], 0], [sg[2], sg[1], 0], [0, 0, 0]])
            sglob = E012 * s3 * E012.T                            # as written: sigma[i, j] = E012[i, k] sigma_hat_3d[k, l] E012[j, l]
            comps = sm.Matrix([sglob[0, 0], sglob[1, 1], sglob[2, 2], sglob[0, 1], sglob[0, 2], sglob[1, 2]]) * Ju_ * detg_
            fs = sm.lambdify((xi, eta), comps, "mpmath")
            x5 = sorted(sm.Poly(sm.legendre(5, xi), xi).nroots(n=40))
            dP5 = sm.diff(sm.legendre(5, xi), xi)
            w5 = [mp.mpf(str(2 / ((1 - xv * xv) * dP5.subs(xi, xv) ** 2))) for xv in x5]
            x5 = [mp.mpf(str(xv)) for xv in x5]
            acc = mp.zeros(6, 1)
            for i in range(5):
                for j in range(5):
                    acc += w5[i] * w5[j] * fs(x5[i], x5[j])
            sum_stress = np.array([float(v) for v in acc])
    # compliance density u_mid . u_mid J(uhat) + 1/2 1e-2 grad(h) . grad(h)  (rm_shell_pde.py:64-89, nodal thickness: 'H1'), and the mass
    # density rho h J(uhat) (:101-102), per unit of the reference measure (x detg)
    detg = sm.sqrt(a.dot(a))
    Ju = (sm.eye(3) + grad(vec(Q1, Uhat))).det()
    gh = sm.Matrix([[h]]).jacobian([xi, eta]) * Kinv
    comp = (u_mid.dot(u_mid) * Ju + sm.Rational(1, 2) * sm.Rational(1, 100) * (gh * gh.T)[0, 0]) * detg
    rho = R(27, 10)
    mdens = rho * h * Ju * detg
    fun = np.array([[float(comp.subs({xi: px, eta: py}).evalf(30)), float(mdens.subs({xi: px, eta: py}).evalf(30))] for px, py in pts])
    print("von Mises stresses and functional densities done")
    return dict(S_fun=fun, S_rho=np.array([float(rho)]), S_pnorm=np.array([pnorm]), S_sum_stress=sum_stress, S_X=np.array(X, float), S_uhat=np.array(Uhat, float), S_h=np.array(hn, float), S_E=np.array(En, float),
                S_nu=np.array(nun, float), S_U=np.array(U, float), S_TH=np.array(TH, float), S_pts=np.array(pts, float), S_vm=out)


def case_P():
    """The penalty blocks of every facet of the warped quadrilateral with uhat != 0 (linear_shell_model.py:323-333, per unit of beta):
    1 / h_K  int_facet || J F^-T N ||  L_i L_j ds  with the three-point Gauss rule of the reference's degree-4 facet measure
    (utils_dolfinx.py:556), for the P2 trace (vertex a, midpoint, vertex b) and the P1 trace (a, b)."""
    R = sm.Rational
    X = [[R(0), R(0), R(0)], [R(1), R(1, 10), R(1, 5)], [R(6, 5), R(9, 10), R(-1, 10)], [R(-1, 10), R(1), R(3, 10)]]
    Uq = [[R(1, 50), R(-1, 100), R(3, 100)], [R(-1, 50), R(1, 40), R(0)], [R(1, 100), R(1, 100), R(-1, 50)], [R(0), R(-3, 100), R(1, 100)]]
    quad_edges = [(lambda s: (s, -1), (0, -1)), (lambda s: (1, s), (1, 0)), (lambda s: (-s, 1), (0, 1)), (lambda s: (-1, -s), (-1, 0))]
    g3 = [-sm.sqrt(R(3, 5)), R(0), sm.sqrt(R(3, 5))]
    w3 = [R(5, 9), R(8, 9), R(5, 9)]
    nf = nanson(X, Uq, Q1, quad_edges, g3)                      # [facet][point]
    hK = max(float(sm.sqrt(sum((X[i][c] - X[j][c]) ** 2 for c in range(3)))) for i, j in itertools.combinations(range(4), 2))
    M2 = np.zeros((4, 3, 3)); M1 = np.zeros((4, 2, 2))
    for k in range(4):
        a, b = X[k], X[(k + 1) % 4]
        length = float(sm.sqrt(sum((a[c] - b[c]) ** 2 for c in range(3))))
        for iq in range(3):
            sq = float(g3[iq])
            L2 = np.array([sq * (sq - 1) / 2, 1 - sq * sq, sq * (sq + 1) / 2])
            L1 = np.array([(1 - sq) / 2, (1 + sq) / 2])
            wq = float(w3[iq]) * 0.5 * length * nf[k, iq] / hK
            M2[k] += wq * np.outer(L2, L2); M1[k] += wq * np.outer(L1, L1)
    print("penalty blocks done")
    return dict(P_X=np.array(X, float), P_uhat=np.array(Uq, float), P_M2=M2, P_M1=M1)


def case_PC():
    """Penalty blocks of the three facets of the affine triangle T for the CG2CR1 element (uhat = 0; per unit of beta, with 1 / h_K):
    the P2 trace of the displacement (vertex a, midpoint, vertex b) and the trace of the Crouzeix-Raviart rotation, in which ALL THREE
    functions of the cell take part (they are not zero on the other edges).  Exact integrals along the facet."""
    R = sm.Rational
    s_ = sm.Symbol("s")
    x0 = [R(1, 10), R(-1, 5), R(3, 10)]
    d0 = [R(6, 5), 0, R(8, 5)]
    d1 = [R(3, 5), R(4, 5), R(4, 5)]
    X = [x0, [x0[c] + d0[c] for c in range(3)], [x0[c] + d1[c] for c in range(3)]]
    hK = max(sm.sqrt(sum((X[i][c] - X[j][c]) ** 2 for c in range(3))) for i, j in itertools.combinations(range(3), 2))
    t = (s_ + 1) / 2
    pts = [(t, 0), (1 - t, t), (0, 1 - t)]                      # facet k from vertex k to k + 1, s in [-1, 1]
    M2 = np.zeros((3, 3, 3)); MR = np.zeros((3, 3, 3))
    for k in range(3):
        a, b = X[k], X[(k + 1) % 3]
        length = sm.sqrt(sum((a[c] - b[c]) ** 2 for c in range(3)))
        sub = {xi: pts[k][0], eta: pts[k][1]}
        tr2 = [N2[k].subs(sub), N2[3 + k].subs(sub), N2[(k + 1) % 3].subs(sub)]
        trR = [NCR[i].subs(sub) for i in range(3)]
        for i in range(3):
            for j in range(3):
                M2[k, i, j] = float(sm.integrate(sm.expand(tr2[i] * tr2[j]), (s_, -1, 1)) * length / 2 / hK)
                MR[k, i, j] = float(sm.integrate(sm.expand(trR[i] * trR[j]), (s_, -1, 1)) * length / 2 / hK)
    print("CG2CR1 penalty blocks done")
    return dict(PC_X=np.array(X, float), PC_M2=M2, PC_MR=MR)


def case_W(n=5):
    """The warped, non-planar quadrilateral of make_sympy_golden.py's case B (uhat != 0, nodal h / E / nu) INTEGRATED with the n x n
    Gauss-Legendre rule: K_e = sum_q w_q B(q)^T C(q) B(q) detg(q) and the load vector, every point evaluated from the symbolic
    expressions at 40 digits.  What a kernel that uses the same rule must reproduce -- no oracle in between."""
    import importlib.util
    import mpmath as mp
    spec = importlib.util.spec_from_file_location("make_sympy_golden", os.path.join(os.path.dirname(os.path.abspath(__file__)), "make_sympy_golden.py"))
    q = importlib.util.module_from_spec(spec); spec.loader.exec_module(q)
    R = sm.Rational
    X = [[R(0), R(0), R(0)], [R(1), R(1, 10), R(1, 5)], [R(6, 5), R(9, 10), R(-1, 10)], [R(-1, 10), R(1), R(3, 10)]]
    hn = [R(1, 20), R(3, 50), R(1, 25), R(11, 200)]
    En = [R(2), R(5, 2), R(9, 4), R(3)]
    nun = [R(3, 10), R(1, 4), R(7, 20), R(1, 5)]
    Uhat = [[R(1, 50), R(-1, 100), R(3, 100)], [R(-1, 50), R(1, 40), R(0)], [R(1, 100), R(1, 100), R(-1, 50)], [R(0), R(-3, 100), R(1, 100)]]
    fn = [[R(1), R(-2), R(1, 2)], [R(1, 3), R(0), R(2)], [R(-1), R(1), R(1)], [R(1, 4), R(1, 5), R(-3)]]
    mp.mp.dps = 40
    strains, geo = q.build(X, Uhat, hn, En, nun)
    C = q.cmat(geo)
    Bsym = sm.Matrix([strains(U, TH)[0] for _, U, TH in q.unit_dofs()]).T                # 9 x 39
    fB = sm.lambdify((xi, eta), Bsym, "mpmath")
    fC = sm.lambdify((xi, eta), C, "mpmath")
    fd = sm.lambdify((xi, eta), sm.Matrix([geo["detg"], geo["Ju"]]), "mpmath")
    f = q.vec(q.N1, fn)
    fF = sm.lambdify((xi, eta), sm.Matrix([q.N2[a] * f[c] for a in range(9) for c in range(3)]), "mpmath")
    xs = sorted(sm.Poly(sm.legendre(n, xi), xi).nroots(n=40))
    dP = sm.diff(sm.legendre(n, xi), xi)
    ws = [2 / ((1 - x * x) * dP.subs(xi, x) ** 2) for x in xs]
    Ke = mp.zeros(39, 39); Fe = mp.zeros(27, 1)
    for x1, w1 in zip(xs, ws):
        for x2, w2 in zip(xs, ws):
            a, b = mp.mpf(str(x1)), mp.mpf(str(x2))
            B = fB(a, b); Cn = fC(a, b); dj = fd(a, b)
            wq = mp.mpf(str(w1)) * mp.mpf(str(w2)) * dj[0]
            Ke += wq * (B.T * Cn * B)
            Fe += wq * dj[1] * fF(a, b)
    # the functionals on a given state with the same rule: compliance = int u.u J dx + 1/2 1e-2 int grad(h).grad(h) dx, mass = int rho h J dx
    # (rm_shell_pde.py:64-102); the elastic energy is 1/2 w^T K_e w
    rs = np.random.default_rng(11)
    U = [[R(int(v), 1000) for v in row] for row in rs.integers(-40, 40, (9, 3))]
    TH = [[R(int(v), 1000) for v in row] for row in rs.integers(-60, 60, (4, 3))]
    u_mid = q.vec(q.N2, U)
    x = q.vec(q.N1, X)
    Jg = x.jacobian([xi, eta])
    Kinv = (Jg.T * Jg).inv() * Jg.T
    hf = sum(q.N1[b] * hn[b] for b in range(4))
    gh = sm.Matrix([[hf]]).jacobian([xi, eta]) * Kinv
    rho = R(27, 10)
    ffun = sm.lambdify((xi, eta), sm.Matrix([u_mid.dot(u_mid), (gh * gh.T)[0, 0], hf]), "mpmath")
    comp = mp.mpf(0); mass = mp.mpf(0)
    for x1, w1 in zip(xs, ws):
        for x2, w2 in zip(xs, ws):
            a, b = mp.mpf(str(x1)), mp.mpf(str(x2))
            dj = fd(a, b); v = ffun(a, b)
            wq = mp.mpf(str(w1)) * mp.mpf(str(w2)) * dj[0]
            comp += wq * (v[0] * dj[1] + mp.mpf(1) / 200 * v[1])
            mass += wq * mp.mpf(27) / 10 * v[2] * dj[1]
    # the thickness sensitivities on that state and a given multiplier lam (rm_shell_model.py:216-232, state_operation.py:242-258):
    #   ((dR/dh)^T lam)_b = int lam^T B^T (dC/dh) B w  N_b dx      (the load does not depend on h),
    #   (d compliance / dh)_b = 1e-2 int grad(h) . grad(N_b) dx      (the H1 regularisation; u.u J does not depend on h)
    Hs = sm.Symbol("Hs")
    geoH = dict(geo); geoH["h"] = Hs
    dC = sm.diff(q.cmat(geoH), Hs).subs(Hs, geo["h"])
    fdC = sm.lambdify((xi, eta), dC, "mpmath")
    LU = [[R(int(v), 1000) for v in row] for row in rs.integers(-50, 50, (9, 3))]
    LT = [[R(int(v), 1000) for v in row] for row in rs.integers(-50, 50, (4, 3))]
    wv = mp.matrix([mp.mpf(int(v.p)) / int(v.q) for row in U for v in row] + [mp.mpf(int(v.p)) / int(v.q) for row in TH for v in row])
    lv = mp.matrix([mp.mpf(int(v.p)) / int(v.q) for row in LU for v in row] + [mp.mpf(int(v.p)) / int(v.q) for row in LT for v in row])
    fgh = sm.lambdify((xi, eta), sm.Matrix([[(gh * (sm.Matrix([[q.N1[b_]]]).jacobian([xi, eta]) * Kinv).T)[0, 0] for b_ in range(4)]]), "mpmath")
    fN1 = sm.lambdify((xi, eta), sm.Matrix(list(q.N1)), "mpmath")
    dRdh = [mp.mpf(0)] * 4; dJdh = [mp.mpf(0)] * 4
    for x1, w1 in zip(xs, ws):
        for x2, w2 in zip(xs, ws):
            a, b = mp.mpf(str(x1)), mp.mpf(str(x2))
            dj = fd(a, b); B = fB(a, b)
            wq = mp.mpf(str(w1)) * mp.mpf(str(w2)) * dj[0]
            val = ((B * lv).T * fdC(a, b) * (B * wv))[0, 0]
            n1 = fN1(a, b); gg = fgh(a, b)
            for b_ in range(4):
                dRdh[b_] += wq * val * n1[b_]
                dJdh[b_] += wq * mp.mpf(1) / 100 * gg[b_]
    # ... and the same for the nodal E and nu fields (derivatives of the constitutive matrix), and for the load:
    #   ((dR/df)^T lam)_(b, c) = - int N_b lam_u[c] J dx     (rm_shell_model.py:216-232)
    Es, Ns = sm.Symbol("Es"), sm.Symbol("Ns")
    geoE = dict(geo); geoE["E"] = Es
    geoN = dict(geo); geoN["nu"] = Ns
    fdCE = sm.lambdify((xi, eta), sm.diff(q.cmat(geoE), Es).subs(Es, geo["E"]), "mpmath")
    fdCN = sm.lambdify((xi, eta), sm.diff(q.cmat(geoN), Ns).subs(Ns, geo["nu"]), "mpmath")
    lam_u = q.vec(q.N2, LU)
    flu = sm.lambdify((xi, eta), lam_u, "mpmath")
    dRdE = [mp.mpf(0)] * 4; dRdnu = [mp.mpf(0)] * 4; dRdf = [[mp.mpf(0)] * 3 for _ in range(4)]
    for x1, w1 in zip(xs, ws):
        for x2, w2 in zip(xs, ws):
            a, b = mp.mpf(str(x1)), mp.mpf(str(x2))
            dj = fd(a, b); B = fB(a, b)
            wq = mp.mpf(str(w1)) * mp.mpf(str(w2)) * dj[0]
            Bl, Bw = B * lv, B * wv
            vE = (Bl.T * fdCE(a, b) * Bw)[0, 0]; vN = (Bl.T * fdCN(a, b) * Bw)[0, 0]
            n1 = fN1(a, b); lu = flu(a, b)
            for b_ in range(4):
                dRdE[b_] += wq * vE * n1[b_]; dRdnu[b_] += wq * vN * n1[b_]
                for c_ in range(3):
                    dRdf[b_][c_] -= wq * dj[1] * n1[b_] * lu[c_]
    sens = dict(W_LU=np.array(LU, float), W_LT=np.array(LT, float), W_dRdh_T_lam=np.array([float(v) for v in dRdh]),
                W_dcompliance_dh=np.array([float(v) for v in dJdh]), W_dRdE_T_lam=np.array([float(v) for v in dRdE]),
                W_dRdnu_T_lam=np.array([float(v) for v in dRdnu]), W_dRdf_T_lam=np.array([[float(v) for v in row] for row in dRdf]))
    # the inertia operator of the dynamic shell, rho h (u.v + h_K^2 theta.eta) J dx (linear_shell_model.py:335-348), with the same rule:
    # per component a 9 x 9 block on the displacement nodes and a 4 x 4 block on the rotation nodes
    hK = max(float(sm.sqrt(sum((X[i][c] - X[j][c]) ** 2 for c in range(3)))) for i, j in itertools.combinations(range(4), 2))
    fN = sm.lambdify((xi, eta), sm.Matrix(list(q.N2) + list(q.N1) + [hf]), "mpmath")
    Mu = mp.zeros(9, 9); Mt = mp.zeros(4, 4)
    for x1, w1 in zip(xs, ws):
        for x2, w2 in zip(xs, ws):
            a, b = mp.mpf(str(x1)), mp.mpf(str(x2))
            dj = fd(a, b); v = fN(a, b)
            wq = mp.mpf(str(w1)) * mp.mpf(str(w2)) * dj[0] * dj[1] * mp.mpf(27) / 10 * v[13]
            n2 = mp.matrix([v[i] for i in range(9)]); n1 = mp.matrix([v[9 + i] for i in range(4)])
            Mu += wq * (n2 * n2.T); Mt += wq * (n1 * n1.T)
    Me = np.zeros((39, 39))
    Mu_, Mt_ = np.array(Mu.tolist(), dtype=float), np.array(Mt.tolist(), dtype=float) * hK ** 2
    for c in range(3):
        Me[np.ix_(range(c, 27, 3), range(c, 27, 3))] = Mu_
        Me[np.ix_(range(27 + c, 39, 3), range(27 + c, 39, 3))] = Mt_
    print("warped quadrilateral, integrated: done")
    return dict(**sens, W_Me=Me, W_U=np.array(U, float), W_TH=np.array(TH, float), W_rho=np.array([float(rho)]), W_compliance=np.array([float(comp)]),
                W_mass=np.array([float(mass)]), W_n=np.array([n]), W_X=np.array(X, float), W_h=np.array(hn, float), W_E=np.array(En, float), W_nu=np.array(nun, float),
                W_uhat=np.array(Uhat, float), W_f=np.array(fn, float),
                W_Ke=np.array(Ke.tolist(), dtype=float), W_Fe=np.array(Fe.tolist(), dtype=float).ravel())


def case_W2(n=5):
    """The shape sensitivity of the residual on the warped quadrilateral of case W, without differentiating anything by hand: with
    uhat = uhat0 + ep D the scalar lam^T R(w; uhat) = lam^T (K_e(uhat) w - F_e(uhat)) is evaluated from the symbolic point values (n x n
    Gauss rule) at ep = +-1e-20 in 50-digit arithmetic; the central difference is D . (dR/duhat)^T lam to ~40 digits.  Three directions D."""
    import importlib.util
    import mpmath as mp
    spec = importlib.util.spec_from_file_location("make_sympy_golden", os.path.join(os.path.dirname(os.path.abspath(__file__)), "make_sympy_golden.py"))
    q = importlib.util.module_from_spec(spec); spec.loader.exec_module(q)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sympy_triangle.npz")
    R = sm.Rational
    X = [[R(0), R(0), R(0)], [R(1), R(1, 10), R(1, 5)], [R(6, 5), R(9, 10), R(-1, 10)], [R(-1, 10), R(1), R(3, 10)]]
    hn = [R(1, 20), R(3, 50), R(1, 25), R(11, 200)]
    En = [R(2), R(5, 2), R(9, 4), R(3)]
    nun = [R(3, 10), R(1, 4), R(7, 20), R(1, 5)]
    U0 = [[R(1, 50), R(-1, 100), R(3, 100)], [R(-1, 50), R(1, 40), R(0)], [R(1, 100), R(1, 100), R(-1, 50)], [R(0), R(-3, 100), R(1, 100)]]
    fn = [[R(1), R(-2), R(1, 2)], [R(1, 3), R(0), R(2)], [R(-1), R(1), R(1)], [R(1, 4), R(1, 5), R(-3)]]
    rs = np.random.default_rng(11)                                # the state and the multiplier of case W
    U = [[R(int(v), 1000) for v in row] for row in rs.integers(-40, 40, (9, 3))]
    TH = [[R(int(v), 1000) for v in row] for row in rs.integers(-60, 60, (4, 3))]
    LU = [[R(int(v), 1000) for v in row] for row in rs.integers(-50, 50, (9, 3))]
    LT = [[R(int(v), 1000) for v in row] for row in rs.integers(-50, 50, (4, 3))]
    mp.mp.dps = 50
    tomp = lambda rows: [mp.mpf(int(v.p)) / int(v.q) for row in rows for v in row]
    wv, lv = mp.matrix(tomp(U) + tomp(TH)), mp.matrix(tomp(LU) + tomp(LT))
    xs = sorted(sm.Poly(sm.legendre(n, xi), xi).nroots(n=50))
    dP = sm.diff(sm.legendre(n, xi), xi)
    ws = [mp.mpf(str(2 / ((1 - x * x) * dP.subs(xi, x) ** 2))) for x in xs]
    xs = [mp.mpf(str(x)) for x in xs]
    f = q.vec(q.N1, fn)
    fF = sm.lambdify((xi, eta), sm.Matrix([q.N2[a] * f[c] for a in range(9) for c in range(3)]), "mpmath")
    ep = sm.Symbol("ep")
    rd = np.random.default_rng(5)
    dirs, vals = [], []
    for _ in range(3):
        D = [[R(int(v), 100) for v in row] for row in rd.integers(-9, 10, (4, 3))]
        Uh = [[U0[b][c] + ep * D[b][c] for c in range(3)] for b in range(4)]
        strains, geo = q.build(X, Uh, hn, En, nun)
        fB = sm.lambdify((xi, eta, ep), sm.Matrix([strains(Uu, Tt)[0] for _, Uu, Tt in q.unit_dofs()]).T, "mpmath")
        fC = sm.lambdify((xi, eta, ep), q.cmat(geo), "mpmath")
        fdj = sm.lambdify((xi, eta, ep), sm.Matrix([geo["detg"], geo["Ju"]]), "mpmath")

        def r(e):
            tot = mp.mpf(0)
            for i in range(n):
                for j in range(n):
                    a, b = xs[i], xs[j]
                    B = fB(a, b, e); dj = fdj(a, b, e)
                    wq = ws[i] * ws[j] * dj[0]
                    Fv = fF(a, b)
                    tot += wq * ((B * lv).T * fC(a, b, e) * (B * wv))[0, 0] - wq * dj[1] * sum(lv[k] * Fv[k] for k in range(27))
            return tot
        h = mp.mpf(10) ** -20
        vals.append(float((r(h) - r(-h)) / (2 * h)))
        dirs.append(np.array(D, float))
        print("shape direction done", vals[-1])
    return dict(W2_D=np.array(dirs), W2_val=np.array(vals))


if __name__ == "__main__":
    out = case_T()
    out.update(case_Q())
    out.update(case_N())
    out.update(case_S())
    out.update(case_P())
    out.update(case_PC())
    import sys
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sympy_triangle.npz")
    if "--keep-W" in sys.argv and os.path.exists(path):          # cases W and W2 take minutes each: keep the committed values
        old = np.load(path)
        out.update({k: old[k] for k in old.files if k.startswith("W_") or k.startswith("W2_")})
        if not any(k.startswith("W2_") for k in old.files):
            out.update(case_W2())
    else:
        out.update(case_W())
        out.update(case_W2())
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sympy_triangle.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)
