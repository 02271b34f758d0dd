"""Generate tests/golden/sympy_triangle_rules.npz -- the symbolic pin of the TRIANGLE quadrature rules (round 6).

The reference integrates its p-norm stress measure with ``quadrature_degree`` 4 (femo_alpha/rm_shell/rm_shell_model.py:200-205) and leaves
the static forms to UFL's estimate (plain dx, linear_shell_fenicsx/linear_shell_model.py:88-103: degree 9 on triangles by
scripts/ufl_degree_estimate.py).  With a nodal Poisson ratio the integrand is RATIONAL on a triangle (E / (1 - nu^2), E / (2 (1 + nu)))
and with rho = 100 the p-norm integrand is far from a polynomial: the rule IS the value.  This script follows the reference's UFL text
symbolically (as make_sympy_golden_tri.py does: surface gradient, CellNormal, gradx = grad F^-1, local frame, Voigt strains, energy
densities, ShellStressRM with the thickness a field) on ONE tilted triangle with uhat != 0 and nodal h / E / nu, and integrates the
symbolic point values in 40-digit arithmetic with the fully symmetric rules of degree 4, 6, 9 and 12 -- whose points and weights it
derives itself from the moment equations (scripts/derive_triangle_rules.py; nothing is read from the oracle or the library):

  TR_Ke_d{6,9,12}      element matrix (27 x 27, CG2CG1) and TR_Fe_d{..} load vector with the rule of that degree
  TR_compliance_d{..}, TR_mass_d{..}    int u.u J dx + 1/2 1e-2 int grad h . grad h dx and int rho h J dx of a given state
  TR_dRdnu_d9, TR_dRdh_d9               ((dR/dnu)^T lam, (dR/dh)^T lam) for a given multiplier with the degree-9 rule
  TR_pnorm4_d4, TR_pnorm100_d4          1/alpha int (m vm)^rho J dx of the top surface with the 6-point rule of degree 4: rho = 4 (m = 2,
                                        alpha = 1) and the reference's rho = 100 (m such that m vm is of order one, alpha = 1)
  TR_pnorm100_d6                        the same integrand with the 12-point rule (what rounds 1-5 integrated it with): a different number

What a kernel (or the oracle) that uses the same rule must reproduce.   Run:  python tests/golden/make_sympy_golden_tri_rules.py   (~1 min)
"""
import importlib.util
import itertools
import os

import mpmath as mp
import numpy as np
import sympy as sm

HERE = os.path.dirname(os.path.abspath(__file__))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


rules = _load(os.path.join(os.path.dirname(os.path.dirname(HERE)), "scripts", "derive_triangle_rules.py"), "derive_triangle_rules")
xi, eta = sm.symbols("xi eta")
K_SHEAR = sm.Rational(833, 1000)
LAM = [1 - xi - eta, xi, eta]
EDGES = [(0, 1), (1, 2), (2, 0)]
N2 = [l * (2 * l - 1) for l in LAM] + [4 * LAM[i] * LAM[j] for i, j in EDGES]
N1 = list(LAM)


def vec(fn, coefs):
    return sm.Matrix([sum(fn[b] * coefs[b][c] for b in range(len(fn))) for c in range(3)])


def main():
    mp.mp.dps = 40
    R = sm.Rational
    X = [[R(0), R(0), R(0)], [R(1), R(1, 10), R(1, 5)], [R(1, 5), R(9, 10), R(-1, 10)]]
    Uhat = [[R(1, 50), R(-1, 100), R(3, 100)], [R(-1, 50), R(1, 40), R(0)], [R(1, 100), R(1, 100), R(-1, 50)]]
    hn = [R(1, 20), R(3, 50), R(1, 25)]
    En = [R(2), R(5, 2), R(9, 4)]
    nun = [R(3, 10), R(1, 5), R(2, 5)]                               # a NODAL Poisson ratio that varies over the cell: rational integrand
    fn = [[R(1), R(-2), R(1, 2)], [R(1, 3), R(0), R(2)], [R(-1), R(1), R(1)]]
    rho_m = R(27, 10)
    rs = np.random.default_rng(23)
    U = [[R(int(v), 1000) for v in row] for row in rs.integers(-40, 40, (6, 3))]
    TH = [[R(int(v), 1000) for v in row] for row in rs.integers(-60, 60, (3, 3))]
    LU = [[R(int(v), 1000) for v in row] for row in rs.integers(-50, 50, (6, 3))]
    LT = [[R(int(v), 1000) for v in row] for row in rs.integers(-50, 50, (3, 3))]

    # ---- the reference's kinematics, operation by operation (kinematics.py:12-106) ----
    x = vec(N1, X)
    Jg = x.jacobian([xi, eta])
    a = Jg[:, 0].cross(Jg[:, 1])
    detg = sm.sqrt(a.dot(a))
    n = a / detg                                                     # CellNormal
    Kinv = (Jg.T * Jg).inv() * Jg.T
    grad = lambda v: v.jacobian([xi, eta]) * Kinv                    # UFL grad on the manifold
    F = sm.eye(3) + grad(vec(N1, Uhat))                              # kinematics.py:42-44
    Finv = F.inv()
    Ju = F.det()                                                     # :32
    gradx = lambda v: grad(v) * Finv                                 # :21
    A0 = Jg[:, 0]
    E0 = A0 / sm.sqrt(A0.dot(A0))                                    # :66-67
    E1 = n.cross(E0)                                                 # :68
    T = sm.Matrix([E0.T, E1.T])                                      # :79-80
    h = sum(N1[b] * hn[b] for b in range(3)); E = sum(N1[b] * En[b] for b in range(3)); nu = sum(N1[b] * nun[b] for b in range(3))
    hK = max(sm.sqrt(sum((X[i][c] - X[j][c]) ** 2 for c in range(3))) for i, j in itertools.combinations(range(3), 2))   # CellDiameter

    def strains(Uc, THc):
        u = vec(N2, Uc); th = vec(N1, THc)
        gradu = gradx(u)                                             # linear_shell_model.py:220
        t_gu = T * gradu * T.T                                       # :222
        eps = (t_gu + t_gu.T) / 2                                    # :238
        gb = T * gradx(n.cross(th)) * T.T                            # :242
        kap = (gb + gb.T) / 2
        gam = T * (-(n.cross(th))) + T * (gradu.T * n)               # :252-257
        om = (t_gu[0, 1] - t_gu[1, 0]) / 2 + th.dot(n)               # :288-289
        return [eps[0, 0], eps[1, 1], 2 * eps[0, 1], kap[0, 0], kap[1, 1], 2 * kap[0, 1], gam[0], gam[1], om]

    def cmat(h_, E_, nu_):
        Cp = (E_ / (1 - nu_ * nu_)) * sm.Matrix([[1, nu_, 0], [nu_, 1, 0], [0, 0, (1 - nu_) / 2]])
        G = E_ / 2 / (1 + nu_)
        C = sm.zeros(9, 9)
        C[0:3, 0:3] = h_ * Cp                                        # linear_shell_model.py:136-157; membrane and bending carry no J (Q4)
        C[3:6, 3:6] = h_ ** 3 / 12 * Cp
        C[6, 6] = C[7, 7] = K_SHEAR * G * h_ * Ju                    # :275-277
        C[8, 8] = E_ * h_ ** 3 / hK ** 2 * Ju                        # :284-296
        return C

    def unit_dofs():
        for i in range(27):
            Uc = [[0] * 3 for _ in range(6)]; THc = [[0] * 3 for _ in range(3)]
            if i < 18:
                Uc[i // 3][i % 3] = 1
            else:
                THc[(i - 18) // 3][(i - 18) % 3] = 1
            yield Uc, THc

    Bsym = sm.Matrix([strains(Uc, THc) for Uc, THc in unit_dofs()]).T                     # 9 x 27
    fB = sm.lambdify((xi, eta), Bsym, "mpmath")
    fC = sm.lambdify((xi, eta), cmat(h, E, nu), "mpmath")
    Hs, Ns = sm.symbols("Hs Ns")
    fdCh = sm.lambdify((xi, eta), sm.diff(cmat(Hs, E, nu), Hs).subs(Hs, h), "mpmath")
    fdCn = sm.lambdify((xi, eta), sm.diff(cmat(h, E, Ns), Ns).subs(Ns, nu), "mpmath")
    fd = sm.lambdify((xi, eta), sm.Matrix([detg, Ju]), "mpmath")
    f = vec(N1, fn)
    fF = sm.lambdify((xi, eta), sm.Matrix([N2[a_] * f[c] for a_ in range(6) for c in range(3)]), "mpmath")
    u_mid = vec(N2, U)
    gh = sm.Matrix([[h]]).jacobian([xi, eta]) * Kinv
    ffun = sm.lambdify((xi, eta), sm.Matrix([u_mid.dot(u_mid), (gh * gh.T)[0, 0], h]), "mpmath")
    fN1 = sm.lambdify((xi, eta), sm.Matrix(N1), "mpmath")
    q = lambda r: mp.mpf(int(r.p)) / int(r.q)
    wv = mp.matrix([q(v) for row in U for v in row] + [q(v) for row in TH for v in row])
    lv = mp.matrix([q(v) for row in LU for v in row] + [q(v) for row in LT for v in row])

    # ---- ShellStressRM at the top surface, thickness a field (linear_shell_model.py:350-467; rm_shell_pde.py:112-128) ----
    theta = vec(N1, TH)
    u_top = u_mid - (R(1, 2) * h) * n.cross(theta)                   # :393-398 with xi2 = h / 2
    gl = T * gradx(u_top) * T.T                                      # :401-409
    em = (gl + gl.T) / 2
    epsv = sm.Matrix([em[0, 0], em[1, 1], 2 * em[0, 1]])             # :412-420
    D = (E / (1 - nu * nu)) * sm.Matrix([[1, nu, 0], [nu, 1, 0], [0, 0, (1 - nu) / 2]])
    sg = D * epsv                                                    # :433-442
    vm = sm.sqrt(sg[0] ** 2 - sg[0] * sg[1] + sg[1] ** 2 + 3 * sg[2] ** 2)             # :459-467
    fvm = sm.lambdify((xi, eta), vm, "mpmath")

    out = dict(TR_X=np.array(X, float), TR_uhat=np.array(Uhat, float), TR_h=np.array(hn, float), TR_E=np.array(En, float),
               TR_nu=np.array(nun, float), TR_f=np.array(fn, float), TR_rho=np.array([float(rho_m)]), TR_U=np.array(U, float),
               TR_TH=np.array(TH, float), TR_LU=np.array(LU, float), TR_LT=np.array(LT, float))
    pts_of = {}
    for deg in (4, 6, 9, 12):
        with mp.workdps(60):
            orbits, res, _ = rules.derive(deg)
            assert res < mp.mpf(10) ** -50
            pts_of[deg] = [(+x_, +y_, w_ / 2) for x_, y_, w_ in rules.points(orbits)]      # weights sum to the area 1/2
    for deg in (6, 9, 12):
        Ke = mp.zeros(27, 27); Fe = mp.zeros(18, 1); comp = mp.mpf(0); mass = mp.mpf(0)
        dRdn = [mp.mpf(0)] * 3; dRdh = [mp.mpf(0)] * 3
        for a_, b_, w_ in pts_of[deg]:
            B = fB(a_, b_); dj = fd(a_, b_); wq = w_ * dj[0]
            Ke += wq * (B.T * fC(a_, b_) * B)
            Fe += wq * dj[1] * fF(a_, b_)
            v = ffun(a_, b_)
            comp += wq * (v[0] * dj[1] + mp.mpf(1) / 200 * v[1])
            mass += wq * q(rho_m) * v[2] * dj[1]
            if deg == 9:
                Bl, Bw = B * lv, B * wv
                vn = (Bl.T * fdCn(a_, b_) * Bw)[0, 0]; vh = (Bl.T * fdCh(a_, b_) * Bw)[0, 0]
                n1 = fN1(a_, b_)
                for k in range(3):
                    dRdn[k] += wq * vn * n1[k]; dRdh[k] += wq * vh * n1[k]
        out[f"TR_Ke_d{deg}"] = np.array(Ke.tolist(), dtype=float)
        out[f"TR_Fe_d{deg}"] = np.array(Fe.tolist(), dtype=float).ravel()
        out[f"TR_compliance_d{deg}"] = np.array([float(comp)])
        out[f"TR_mass_d{deg}"] = np.array([float(mass)])
        if deg == 9:
            out["TR_dRdnu_d9"] = np.array([float(v) for v in dRdn]); out["TR_dRdh_d9"] = np.array([float(v) for v in dRdh])
        print(f"degree {deg}: {len(pts_of[deg])} points, element matrix / load / functionals done", flush=True)
    # how far the rules are from each other on this rational integrand (for the record)
    d69 = np.abs(out["TR_Ke_d6"] - out["TR_Ke_d9"]).max() / np.abs(out["TR_Ke_d9"]).max()
    d912 = np.abs(out["TR_Ke_d9"] - out["TR_Ke_d12"]).max() / np.abs(out["TR_Ke_d12"]).max()
    out["TR_Ke_rule_distance"] = np.array([d69, d912])
    print(f"element matrix: degree 6 against 9: {d69:.2e}; 9 against 12: {d912:.2e}")
    # the p-norm aggregates: m vm of order one for rho = 100 (the reference's m = 1e-6 plays that part for stresses in Pa)
    vms = [fvm(a_, b_) for a_, b_, _ in pts_of[4]]
    m100 = 1 / float(max(vms))
    m100 = float(np.float64(m100))                                   # the caller passes a double: integrate with exactly that number
    out["TR_m100"] = np.array([m100])
    for deg in (4, 6):
        p4 = mp.mpf(0); p100 = mp.mpf(0)
        for a_, b_, w_ in pts_of[deg]:
            dj = fd(a_, b_); s = fvm(a_, b_)
            p4 += w_ * dj[0] * dj[1] * (2 * s) ** 4
            p100 += w_ * dj[0] * dj[1] * (mp.mpf(m100) * s) ** 100
        out[f"TR_pnorm4_d{deg}"] = np.array([float(p4)]); out[f"TR_pnorm100_d{deg}"] = np.array([float(p100)])
    print(f"p-norm aggregate, rho = 100: degree-4 rule {out['TR_pnorm100_d4'][0]:.6e}, degree-6 rule {out['TR_pnorm100_d6'][0]:.6e} "
          f"(ratio {out['TR_pnorm100_d6'][0] / out['TR_pnorm100_d4'][0]:.4f}); rho = 4: {out['TR_pnorm4_d4'][0]:.6e} / {out['TR_pnorm4_d6'][0]:.6e}")
    path = os.path.join(HERE, "sympy_triangle_rules.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)


if __name__ == "__main__":
    main()
