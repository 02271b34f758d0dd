#!/usr/bin/env python3
"""Which quadrature degree does the reference integrate its static forms with?

The reference never chooses one: ``ShellElement`` hands out the plain ``dx`` unless degrees are passed
(femo_alpha/rm_shell/linear_shell_fenicsx/linear_shell_model.py:88-103; RMShellPDE passes none,
rm_shell/rm_shell_pde.py:27-33), so the degree is whatever fenics-ufl 2022.2.0 (environment.yml:13-19) *estimates* for
the integrand and fenics-ffcx 0.5.0 turns into a rule.  Neither package is installed here, so this
script restates the published estimation algorithm -- ``ufl/algorithms/estimate_degrees.py``, class SumDegreeEstimator,
and ``ufl/algorithms/apply_integral_scaling.py`` -- and applies it, operation by operation, to the expressions of
linear_shell_model.py:136-157 (CLT), :199-306 (strains, energies) and kinematics.py:12-106.  It is a restatement from
the published source text, not an execution of UFL: every rule used is listed in RULES below so that a maintainer with
FEniCSx at hand can check it in a minute (``ufl.algorithms.estimate_total_polynomial_degree(integrand)``).

The rules (UFL 2022.2.0):
  R1  constants, Identity: 0.  Coefficient / Argument: degree of its element; a function on a MixedElement carries the
      mixed element's degree (= the largest sub-degree), and so does everything ``split`` from it.
  R2  sum, conditional, list tensor (as_vector / as_matrix), indexing, IndexSum: the maximum over the operands.
  R3  product, inner, dot, outer, cross: the sum of the operand degrees.
  R4  division: the SUM of numerator and denominator degree ("a heuristic", the docstring says).
  R5  power with a non-negative integer exponent p: p * degree; any other power: degree + 2.
  R6  math functions (sqrt, exp, ...): degree + 2 (0 stays 0).
  R7  spatial derivative (grad): degree - 1 on simplices; NO reduction on quadrilaterals / hexahedra.
  R8  geometric quantities: 0 when cellwise constant, else the degree q of the coordinate element.  Jacobian is cellwise
      constant only on affine simplices; CellDiameter is cellwise constant; SpatialCoordinate has degree q.
      CellNormal: constant on affine simplices; on quadrilaterals the 2022.2.0 source is not at hand -- both answers are
      carried through below (c_n = 0 or q) and the conclusion does not depend on it.
  R9  inv / det of a 3 x 3 matrix are expanded by apply_algebra_lowering BEFORE the estimate: det -> 3 d,
      inv = adj / det -> 2 d + 3 d = 5 d  (R3, R4).
  R10 derivative(form, w, v) is applied before the estimate; it replaces one occurrence of w by v, which has the same
      element: the degree of energy, residual and Jacobian form is the same.
  R11 integrals of one form with the same measure and metadata are summed into ONE integrand (group_form_integrals) before
      the estimate: its degree is the maximum over the energies (R2).
  R12 apply_integral_scaling multiplies by |detJ| of the cell map and ADDS that factor's estimated degree: 0 on affine
      simplices; on a quadrilateral surface cell detJ = sqrt(|J0 x J1|^2): (q + q) * 2 + 2  (R3, R5, R6).
  FFCx 0.5.0 takes the estimate as the quadrature degree (it only warns above 30) and asks basix for the default rule: Gauss-Jacobi
  with (degree + 2) // 2 points per direction on quadrilaterals.
"""
import argparse


class Est:
    def __init__(self, cell, q=1, c_n=None, deg_u=2, deg_field=1, deg_uhat=1):
        self.quad = cell == "quadrilateral"
        self.q = q
        self.const_geom = not self.quad                  # affine simplex
        self.c_n = (0 if self.const_geom else q) if c_n is None else c_n
        self.du, self.df, self.dh = deg_u, deg_field, deg_uhat

    # R3 / R4 / R5 / R6 / R7
    @staticmethod
    def mul(*d): return sum(d)
    @staticmethod
    def div(a, b): return a + b
    @staticmethod
    def ipow(a, p): return a * p
    @staticmethod
    def fn(a): return a + 2 if a else a
    def grad(self, a): return a if self.quad else max(a - 1, 0)

    def run(self, log=print):
        E = self
        q = 0 if E.const_geom else E.q                   # Jacobian entries (R8)
        F = max(0, E.grad(E.dh))                        # I + grad(uhat)
        invF, J = 5 * F, 3 * F                          # R9
        gradx = lambda a: E.mul(E.grad(a), invF)        # dot(grad(f), inv(F))
        E0 = E.div(q, E.fn(E.mul(q, q)))                # A0 / sqrt(dot(A0, A0))
        E2 = E.c_n
        E1 = E.mul(E2, E0)                              # cross(E2, E0)
        T = max(E0, E1)                                 # as_matrix (R2)
        u = th = E.du                                   # split(w): the mixed element's degree (R1)
        gu = gradx(u)
        t_gu = E.mul(T, gu, T)                          # T[i,k] gradu[k,l] T[j,l]
        kappa = E.mul(T, gradx(E.mul(E2, th)), T)       # sym(gradv_local(gradx(cross(E2, theta)), T))
        eps = max(t_gu, E.mul(0, kappa))                # sym(t_gu) - offset * kappa, offset in DG0
        gamma = max(E.mul(E.mul(E2, th), T), E.mul(E.mul(E2, gu), T))
        h = Em = nu = E.df
        G = E.div(E.div(Em, 0), max(0, nu))             # E / 2 / (1 + nu)
        C = E.mul(E.div(Em, max(0, E.mul(nu, nu))), max(0, nu))     # E / (1 - nu nu) * as_matrix([[1, nu, 0], ...])
        A, B, D, As = E.mul(h, C), E.mul(0, C), E.mul(E.div(E.ipow(h, 3), 0), C), E.mul(0, G, h, 0)
        N = max(E.mul(A, eps), E.mul(B, kappa))
        M = max(E.mul(B, eps), E.mul(D, kappa))
        Q = E.mul(As, gamma)
        membrane = E.mul(0, E.mul(N, eps))              # 0.5 dot(N, voigt(eps)) dx   (no J, quirk Q4)
        bending = E.mul(0, E.mul(M, kappa))
        shear = E.mul(0, E.mul(Q, gamma), J)
        strain = max(E.div(t_gu, 0), E.mul(th, E2))     # (t_gu[0,1] - t_gu[1,0]) / 2 + dot(theta, E2)
        alpha = E.mul(Em, E.ipow(h, 3))
        stress = E.div(E.mul(alpha, strain), E.ipow(0, 2))          # / CellDiameter ** 2
        drilling = E.mul(0, stress, strain, J)
        integrand = max(membrane, bending, shear, drilling)         # R11
        scale = 0 if E.const_geom else E.fn(E.ipow(E.mul(q, q), 2))  # R12
        total = integrand + scale
        npts = (total + 2) // 2
        cell = "quadrilateral" if E.quad else "triangle"
        log(f"  {cell}, CellNormal degree {E.c_n}, fields in {'CG1' if E.df else 'DG0'}:")
        log(f"    inv(F) {invF}, J(uhat) {J}, E0 {E0}, E1 {E1}, gradx(u) {gu}, t_gu {t_gu}, kappa {kappa}, eps {eps}, gamma {gamma}")
        log(f"    C {C}, A {A}, D {D}, A_s {As};  membrane {membrane}, bending {bending}, shear {shear}, drilling {drilling}")
        log(f"    integrand {integrand} + cell-map scaling {scale} = estimated degree {total}"
            + (f"  ->  Gauss-Jacobi {npts} x {npts} points per cell" if E.quad else ""))
        return total


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.parse_args()
    print("Estimated quadrature degree of d^2(elastic energy)/dw^2 (the Jacobian the reference assembles), UFL 2022.2.0 rules:\n")
    out = {}
    for cell in ("quadrilateral", "triangle"):
        for c_n in ((0, 1) if cell == "quadrilateral" else (0,)):
            for df in (1, 0):
                out[(cell, c_n, df)] = Est(cell, c_n=c_n, deg_field=df).run()
    lo = min(v for (c, _, _), v in out.items() if c == "quadrilateral")
    print(f"""
Reading:
  * On quadrilaterals every variant lands far above 30 (>= {lo}): FFCx integrates the static forms with >= {(lo + 2) // 2} Gauss points
    per direction -- for a smooth integrand that is exact integration to rounding, on warped cells too.  The estimate is driven by
    the division heuristic (R4) on inv(F(uhat)), on unit(A0) and on E / (1 - nu^2), and by R7 (no reduction on quadrilaterals); it
    does not depend on whether uhat IS zero (the Function is in the expression either way).
  * On flat / affine cells with uniform E, nu the integrand is a polynomial of degree <= 7 per direction (SURVEY.md section 8c):
    the 4 x 4 Gauss rule of this repository integrates it exactly as well -- configs 1, 2 and 5 are independent of the rule.
  * On warped cells (config 3, the wing skin) the integrand is rational: the reference's answer is the n -> infinity limit of
    the n x n rule, ours is n = 4 by default.  tests/test_gpu_fullsize.py::test_quadrature_rule_sensitivity_at_config3 measures
    what n = 4 -> 5 changes; RMShellModel(..., nquad=n) / RMShellPDE(..., nquad=n) select the rule (2..5).
  * On triangles the estimate is {out[('triangle', 0, 1)]} with nodal fields ({out[('triangle', 0, 0)]} element-wise).  On flat (affine) triangles with uniform E, nu the true
    degree is <= 5 (drilling: h^3 (grad u + theta)^2 = 3 + 2), so the degree-6, 12-point rule of this repository and the
    reference's degree-{out[('triangle', 0, 1)]} rule both integrate exactly; with nodal E or nu the integrand is rational and the two rules differ.""")


if __name__ == "__main__":
    main()
