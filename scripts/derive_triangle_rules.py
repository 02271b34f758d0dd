"""Fully symmetric quadrature rules on the triangle, derived to 50 digits (mpmath) -- the literals of
``femo_alpha_amd/csrc/femo_hip.hip::triangle_rule`` and ``oracle/rm_shell_oracle.py::_TRI`` come from this script.

    python scripts/derive_triangle_rules.py            (a few seconds; prints the tables and their residuals)

Why: the quadrature table is part of the definition of the discrete problem (a last-place change of a
weight moves the 1 M-DOF answer by 3.5e-7), so both sides carry the SAME correctly rounded numbers, and 15-digit constants
from a printed table (rounds 1-5: sum of the weights 1 + 2e-15) are not that.

A fully symmetric rule is a set of orbits of the symmetric group of the triangle in barycentric coordinates:
  S3   the centroid (1 point, unknown: weight)
  S21  (a, a, 1 - 2a) and its permutations (3 points, unknowns: a, weight)
  S111 (a, b, 1 - a - b) and its permutations (6 points, unknowns: a, b, weight).
It integrates every polynomial of degree <= d exactly iff it integrates the S3-INVARIANT polynomials of degree <= d exactly,
and those are spanned by e2^i e3^j with 2 i + 3 j <= d (e2 = L0 L1 + L1 L2 + L2 L0, e3 = L0 L1 L2; e1 = 1).  For the four rules
below the number of these moments equals the number of unknowns, so the rule is an isolated solution and Newton's iteration
from the printed table (Dunavant 1985; the same point sets as Xiao-Gimbutas' tables of these sizes, which basix 0.5.0 uses for
``quadrature_degree`` <= 30 on simplices) converges to it:

  degree  4:  6 points = 2 S21                      (4 unknowns, 4 moments)     rm_shell_model.py:200-205 (p-norm stress measure)
  degree  6: 12 points = 2 S21 + 1 S111             (7, 7)                      exact for the static forms on affine cells, uniform E / nu
  degree  9: 19 points = S3 + 4 S21 + 1 S111        (12, 12)                    UFL's estimate for the static forms (scripts/ufl_degree_estimate.py)
  degree 12: 33 points = 5 S21 + 3 S111             (19, 19)                    the convergence check beyond it
"""
import itertools
import sys

import mpmath as mp
import sympy as sp

mp.mp.dps = 60

# seeds: (orbit kind, parameters..., weight) with the weights summing to one
SEEDS = {
    4: [("S21", 0.445948490915965, 0.223381589678011), ("S21", 0.091576213509771, 0.109951743655322)],
    6: [("S21", 0.063089014491502, 0.050844906370207), ("S21", 0.249286745170910, 0.116786275726379),
        ("S111", 0.053145049844817, 0.310352451033784, 0.082851075618374)],
    9: [("S3", 0.097135796282799),
        ("S21", 0.489682519198738, 0.031334700227139), ("S21", 0.437089591492937, 0.077827541004774),
        ("S21", 0.188203535619033, 0.079647738927210), ("S21", 0.044729513394453, 0.025577675658698),
        ("S111", 0.036838412054736, 0.221962989160766, 0.043283539377289)],
    12: [("S21", 0.488217389773805, 0.025731066440455), ("S21", 0.439724392294460, 0.043692544538038),
         ("S21", 0.271210385012116, 0.062858224217885), ("S21", 0.127576145541586, 0.034796112930709),
         ("S21", 0.021317350453210, 0.006166261051559),
         ("S111", 0.115343494534698, 0.275713269685514, 0.040371557766381),
         ("S111", 0.022838332222257, 0.281325580989940, 0.022356773202303),
         ("S111", 0.025734050548330, 0.116251915907597, 0.017316231108659)],
}


def invariant_moments(degree):
    """[(i, j, exact mean of e2^i e3^j over the triangle)] for 2 i + 3 j <= degree."""
    L0, L1, L2 = sp.symbols("L0 L1 L2")
    e2, e3 = L0 * L1 + L1 * L2 + L2 * L0, L0 * L1 * L2
    out = []
    for j in range(degree // 3 + 1):
        for i in range((degree - 3 * j) // 2 + 1):
            poly = sp.Poly(sp.expand(e2 ** i * e3 ** j), L0, L1, L2)
            val = sp.Integer(0)
            for (a, b, c), coef in poly.terms():
                val += coef * 2 * sp.factorial(a) * sp.factorial(b) * sp.factorial(c) / sp.factorial(a + b + c + 2)
            out.append((i, j, sp.Rational(val)))
    return out


def orbit_sums(kind, par, i, j):
    """sum over the orbit's points of e2^i e3^j (without the weight)."""
    if kind == "S3":
        l = (mp.mpf(1) / 3,) * 3
        n = 1
    elif kind == "S21":
        l = (par[0], par[0], 1 - 2 * par[0])
        n = 3
    else:
        l = (par[0], par[1], 1 - par[0] - par[1])
        n = 6
    e2 = l[0] * l[1] + l[1] * l[2] + l[2] * l[0]
    e3 = l[0] * l[1] * l[2]
    return n * e2 ** i * e3 ** j


def derive(degree):
    seeds = SEEDS[degree]
    moments = invariant_moments(degree)
    layout = []                                  # (kind, number of parameters)
    x0 = []
    for s in seeds:
        layout.append((s[0], len(s) - 2))
        x0 += [mp.mpf(repr(v)) for v in s[1:]]
    assert len(x0) == len(moments), (degree, len(x0), len(moments))

    def unpack(x):
        k, orbits = 0, []
        for kind, npar in layout:
            orbits.append((kind, x[k:k + npar], x[k + npar]))
            k += npar + 1
        return orbits

    def F(*x):
        orbits = unpack(x)
        return [sum(w * orbit_sums(kind, par, i, j) for kind, par, w in orbits) - mp.mpf(m.p) / mp.mpf(m.q) for i, j, m in moments]

    x = mp.findroot(F, x0, tol=mp.mpf(10) ** -55, maxsteps=60)
    x = [x[k] for k in range(len(x0))]
    res = max(abs(v) for v in F(*x))
    move = max(abs(a - b) for a, b in zip(x, x0))
    return unpack(x), res, move


def points(orbits):
    """(x, y, weight) in the order both implementations use: orbit by orbit; S21 (a,a) (b,a) (a,b) with b = 1 - 2a; S111 (a,b) (b,a) (a,c)
    (c,a) (b,c) (c,b) with c = 1 - a - b; the weights sum to one (the area 1/2 is applied by the caller)."""
    out = []
    for kind, par, w in orbits:
        if kind == "S3":
            out.append((mp.mpf(1) / 3, mp.mpf(1) / 3, w))
        elif kind == "S21":
            a = par[0]; b = 1 - 2 * a
            out += [(a, a, w), (b, a, w), (a, b, w)]
        else:
            a, b = par; c = 1 - a - b
            out += [(a, b, w), (b, a, w), (a, c, w), (c, a, w), (b, c, w), (c, b, w)]
    return out


def check_all_monomials(pts, degree):
    worst = mp.mpf(0)
    for a, b in itertools.product(range(degree + 1), repeat=2):
        if a + b > degree:
            continue
        exact = 2 * mp.factorial(a) * mp.factorial(b) / mp.factorial(a + b + 2)
        worst = max(worst, abs(sum(w * x ** a * y ** b for x, y, w in pts) - exact))
    return worst


def main():
    fmt = lambda v: mp.nstr(v, 25, strip_zeros=False)
    for degree in (4, 6, 9, 12):
        orbits, res, move = derive(degree)
        pts = points(orbits)
        print(f"# degree {degree}: {len(pts)} points, moment residual {mp.nstr(res, 3)}, moved {mp.nstr(move, 3)} from the printed table, "
              f"all monomials of degree <= {degree}: {mp.nstr(check_all_monomials(pts, degree), 3)}, "
              f"degree {degree + 1} is missed by {mp.nstr(check_all_monomials(pts, degree + 1), 3)}, "
              f"min weight {mp.nstr(min(w for _, _, w in pts), 5)}")
        # every coordinate as its own correctly rounded literal (no 1 - 2 a in floating point: a contraction into a fused
        # multiply-add on one side would part the two tables): S3 (w), S21 (a, b = 1 - 2a, w), S111 (a, b, c = 1 - a - b, w)
        for kind, par, w in orbits:
            full = [] if kind == "S3" else [par[0], 1 - 2 * par[0]] if kind == "S21" else [par[0], par[1], 1 - par[0] - par[1]]
            print(f"    {{{', '.join(fmt(p) for p in full + [w])}}},")
    return 0


if __name__ == "__main__":
    sys.exit(main())
