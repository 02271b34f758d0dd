"""Generate tests/golden/sympy_element.npz -- an INDEPENDENT symbolic derivation of the
reference's shell element energy, used to pin the CPU oracle (parity is otherwise unpinned,
see oracle/rm_shell_oracle.py header).

The script follows the reference's UFL text operation by operation with sympy (surface
gradient = d/dxi * pseudo-inverse Jacobian, CellNormal = normalised J0 x J1 and is
differentiated like UFL does on non-affine cells, cross products, ``gradx = grad . inv(F)``,
...) -- reference femo_alpha/rm_shell/linear_shell_fenicsx/kinematics.py:12-106 and
linear_shell_model.py:136-157,199-321.  It never forms a B matrix by hand and shares no
code with the oracle.

Case A: affine (parallelogram) quad tilted in 3-D, nodal h, uniform E/nu, uhat = 0:
        exact (rational) integration of the element stiffness and load vector.
Case B: warped, non-planar quad, nodal h/E/nu, uhat != 0: point-wise energy-density Hessians
        and load densities at three reference points (no quadrature involved).

Run:  python tests/golden/make_sympy_golden.py   (about a minute)
"""
import itertools
from fractions import Fraction

import numpy as np
import sympy as sm

xi, eta = sm.symbols("xi eta")
K_SHEAR = sm.Rational(833, 1000)

Q2_IJ = [(0, 0), (2, 0), (2, 2), (0, 2), (1, 0), (2, 1), (1, 2), (0, 1), (1, 1)]
Q1_IJ = [(0, 0), (1, 0), (1, 1), (0, 1)]


def lag(nodes, t):
    out = []
    for i, xi_ in enumerate(nodes):
        p = sm.Integer(1)
        for j, xj in enumerate(nodes):
            if i != j:
                p *= (t - xj) / (xi_ - xj)
        out.append(sm.expand(p))
    return out


L2x, L2y = lag([-1, 0, 1], xi), lag([-1, 0, 1], eta)
L1x, L1y = lag([-1, 1], xi), lag([-1, 1], eta)
N2 = [L2x[i] * L2y[j] for i, j in Q2_IJ]
N1 = [L1x[i] * L1y[j] for i, j in Q1_IJ]


def vec(fn, coefs):
    """sum_b fn_b * coefs[b,:] as a 3x1 Matrix."""
    return sm.Matrix([sum(fn[b] * coefs[b][c] for b in range(len(fn))) for c in range(3)])


def build(X, Uhat, hn, En, nun):
    """Return a function strains(U (9x3), TH (4x3)) -> dict of sympy expressions in xi,eta
    plus the geometric scalars, following the UFL text literally."""
    x = vec(N1, X)
    Jg = x.jacobian([xi, eta])
    a = Jg[:, 0].cross(Jg[:, 1])
    detg = sm.sqrt(a.dot(a))
    n = a / detg                                             # CellNormal
    Kinv = (Jg.T * Jg).inv() * Jg.T                          # pseudo-inverse (2x3)

    def grad(v):                                             # UFL grad on the manifold
        return v.jacobian([xi, eta]) * Kinv

    uhat = vec(N1, Uhat)
    F = sm.eye(3) + grad(uhat)                               # kinematics.py:42-44
    Finv = F.inv()
    Ju = F.det()                                             # kinematics.py:32

    def gradx(v):                                            # kinematics.py:21
        return grad(v) * Finv

    A0 = Jg[:, 0]
    E0 = A0 / sm.sqrt(A0.dot(A0))                            # kinematics.py:66-67
    E1 = n.cross(E0)                                         # kinematics.py:68
    T = sm.Matrix([E0.T, E1.T])                              # kinematics.py:79-80
    h = sum(N1[b] * hn[b] for b in range(4))
    E = sum(N1[b] * En[b] for b in range(4))
    nu = sum(N1[b] * nun[b] for b in range(4))
    hK = max(sm.sqrt(sum((X[i][c] - X[j][c]) ** 2 for c in range(3)))
             for i, j in itertools.combinations(range(4), 2))    # CellDiameter

    def strains(U, TH):
        u = vec(N2, U)
        th = vec(N1, TH)
        gradu = gradx(u)                                     # linear_shell_model.py:220
        t_gu = T * gradu * T.T                               # :222, kinematics.py:90-91
        eps = (t_gu + t_gu.T) / 2                            # :238
        gb = T * gradx(n.cross(th)) * T.T                    # :242
        kap = (gb + gb.T) / 2
        gam = T * (-(n.cross(th))) + T * (gradu.T * n)       # :252-257
        om = (t_gu[0, 1] - t_gu[1, 0]) / 2 + th.dot(n)       # :288-289
        return [eps[0, 0], eps[1, 1], 2 * eps[0, 1], kap[0, 0], kap[1, 1], 2 * kap[0, 1],
                gam[0], gam[1], om], u

    geo = dict(detg=detg, Ju=Ju, h=h, E=E, nu=nu, hK=hK)
    return strains, geo


def cmat(geo):
    """9x9 constitutive x (J factors), linear_shell_model.py:136-157, 275-296 (no measure weight)."""
    h, E, nu, Ju, hK = geo["h"], geo["E"], geo["nu"], geo["Ju"], geo["hK"]
    Cp = (E / (1 - nu * nu)) * sm.Matrix([[1, nu, 0], [nu, 1, 0], [0, 0, (1 - nu) / 2]])
    G = E / 2 / (1 + nu)
    C = sm.zeros(9, 9)
    C[0:3, 0:3] = h * Cp
    C[3:6, 3:6] = h ** 3 / 12 * Cp
    C[6, 6] = C[7, 7] = K_SHEAR * G * h * Ju
    C[8, 8] = E * h ** 3 / hK ** 2 * Ju
    return C


def unit_dofs():
    for i in range(39):
        U = [[0] * 3 for _ in range(9)]
        TH = [[0] * 3 for _ in range(4)]
        if i < 27:
            U[i // 3][i % 3] = 1
        else:
            TH[(i - 27) // 3][(i - 27) % 3] = 1
        yield i, U, TH


def poly_int(p):
    """Exact integral over [-1,1]^2 of a sympy polynomial in xi, eta."""
    P = sm.Poly(sm.expand(p), xi, eta)
    tot = sm.Integer(0)
    for (i, j), c in P.terms():
        if i % 2 == 0 and j % 2 == 0:
            tot += c * sm.Rational(2, i + 1) * sm.Rational(2, j + 1)
    return tot


def case_A():
    R = sm.Rational
    x0 = [R(1, 10), R(-1, 5), R(3, 10)]
    d0 = [R(6, 5), 0, R(8, 5)]          # x1 - x0  (|.|/2 = 1)
    d1 = [R(3, 5), R(4, 5), R(4, 5)]    # x3 - x0
    X = [x0, [x0[c] + d0[c] for c in range(3)], [x0[c] + d0[c] + d1[c] for c in range(3)],
         [x0[c] + d1[c] for c in range(3)]]
    hn = [R(1, 10), R(3, 25), R(2, 25), R(11, 100)]
    En = [R(7, 2)] * 4
    nun = [R(3, 10)] * 4
    fn = [[R(1), R(-2), R(1, 2)], [R(1, 3), R(0), R(2)], [R(-1), R(1), R(1)], [R(1, 4), R(1, 5), R(-3)]]
    Uhat = [[0, 0, 0]] * 4
    strains, geo = build(X, Uhat, hn, En, nun)
    C = sm.simplify(cmat(geo))
    detg = sm.simplify(geo["detg"])
    Bs = []
    for i, U, TH in unit_dofs():
        s, _ = strains(U, TH)
        Bs.append([sm.expand(sm.simplify(e)) for e in s])
    Ke = np.zeros((39, 39))
    CB = [[sm.expand(sum(C[k, l] * Bs[j][l] for l in range(9))) for k in range(9)] for j in range(39)]
    for i in range(39):
        for j in range(i, 39):
            integrand = sum(Bs[i][k] * CB[j][k] for k in range(9)) * detg
            Ke[i, j] = Ke[j, i] = float(poly_int(integrand))
    f = vec(N1, fn)
    Fe = np.zeros(27)
    for a in range(9):
        for c in range(3):
            Fe[3 * a + c] = float(poly_int(N2[a] * f[c] * detg))
    return dict(A_X=np.array(X, float), A_h=np.array(hn, float), A_E=np.array(En, float),
                A_nu=np.array(nun, float), A_f=np.array(fn, float), A_Ke=Ke, A_Fe=Fe)


def case_B():
    R = sm.Rational
    X = [[R(0), R(0), R(0)], [R(1), R(1, 10), R(1, 5)], [R(6, 5), R(9, 10), R(-1, 10)], [R(-1, 10), R(1), R(3, 10)]]
    hn = [R(1, 20), R(3, 50), R(1, 25), R(11, 200)]
    En = [R(2), R(5, 2), R(9, 4), R(3)]
    nun = [R(3, 10), R(1, 4), R(7, 20), R(1, 5)]
    Uhat = [[R(1, 50), R(-1, 100), R(3, 100)], [R(-1, 50), R(1, 40), R(0)],
            [R(1, 100), R(1, 100), R(-1, 50)], [R(0), R(-3, 100), R(1, 100)]]
    fn = [[R(1), R(-2), R(1, 2)], [R(1, 3), R(0), R(2)], [R(-1), R(1), R(1)], [R(1, 4), R(1, 5), R(-3)]]
    pts = [(R(-3, 5), R(1, 4)), (R(1, 3), R(-7, 10)), (R(4, 5), R(9, 10))]
    strains, geo = build(X, Uhat, hn, En, nun)
    C = cmat(geo)
    f = vec(N1, fn)
    Kq = np.zeros((len(pts), 39, 39))
    Fq = np.zeros((len(pts), 27))
    Bq = np.zeros((len(pts), 9, 39))
    scal = np.zeros((len(pts), 2))
    for ip, (px, py) in enumerate(pts):
        sub = {xi: px, eta: py}
        Cn = np.array(C.subs(sub).evalf(40), dtype=float)
        detg = float(geo["detg"].subs(sub).evalf(40))
        Ju = float(geo["Ju"].subs(sub).evalf(40))
        scal[ip] = detg, Ju
        for i, U, TH in unit_dofs():
            s, _ = strains(U, TH)
            Bq[ip, :, i] = [float(e.subs(sub).evalf(40)) for e in s]
        Kq[ip] = Bq[ip].T @ Cn @ Bq[ip] * detg               # energy-density Hessian x dx/dxi
        for a in range(9):
            for c in range(3):
                Fq[ip, 3 * a + c] = float((N2[a] * f[c]).subs(sub)) * detg * Ju
    return dict(B_X=np.array(X, float), B_h=np.array(hn, float), B_E=np.array(En, float),
                B_nu=np.array(nun, float), B_uhat=np.array(Uhat, float), B_f=np.array(fn, float),
                B_pts=np.array(pts, float), B_Kq=Kq, B_Fq=Fq, B_Bq=Bq, B_detJu=scal)


if __name__ == "__main__":
    import os
    out = {}
    out.update(case_A())
    print("case A done")
    out.update(case_B())
    print("case B done")
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sympy_element.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)
