// CPU ORACLE -- test infrastructure only, never the product path.
//
// Double-double arithmetic (an unevaluated sum hi + lo of two doubles, ~32 significant digits): the portable twin of the x87 long
// double the goldens' operator is assembled in (cpu_core.inc instantiated with REAL = long double needs an x86 host; VERDICT r5,
// weak 3).  Error-free transformations after Dekker / Knuth with a hardware fused multiply-add (the library is built with
// -march=x86-64-v3; on other hosts std::fma is correct, only slower); "sloppy" addition (relative error ~2^-104), which is three
// orders of magnitude finer than the 64-bit mantissa it stands in for.  No -ffast-math anywhere in this build.
#pragma once
#include <cmath>

struct dd {
    double hi, lo;
    dd() = default;
    dd(double x) : hi(x), lo(0.0) {}
    dd(double h, double l) : hi(h), lo(l) {}
    explicit operator double() const { return hi + lo; }
};

inline dd dd_quick_two_sum(double a, double b) { const double s = a + b; return dd(s, b - (s - a)); }
inline dd dd_two_sum(double a, double b) { const double s = a + b, v = s - a; return dd(s, (a - (s - v)) + (b - v)); }
inline dd dd_two_prod(double a, double b) { const double p = a * b; return dd(p, std::fma(a, b, -p)); }

inline dd operator+(const dd& a, const dd& b) {
    dd s = dd_two_sum(a.hi, b.hi);
    const dd t = dd_two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = dd_quick_two_sum(s.hi, s.lo);
    s.lo += t.lo;
    return dd_quick_two_sum(s.hi, s.lo);
}
inline dd operator-(const dd& a) { return dd(-a.hi, -a.lo); }
inline dd operator-(const dd& a, const dd& b) { return a + (-b); }
inline dd operator*(const dd& a, const dd& b) {
    dd p = dd_two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return dd_quick_two_sum(p.hi, p.lo);
}
inline dd operator/(const dd& a, const dd& b) {
    const double q1 = a.hi / b.hi;
    dd r = a - b * dd(q1);
    const double q2 = r.hi / b.hi;
    r = r - b * dd(q2);
    const double q3 = r.hi / b.hi;
    dd q = dd_quick_two_sum(q1, q2);
    return q + dd(q3);
}
inline dd& operator+=(dd& a, const dd& b) { a = a + b; return a; }
inline dd& operator-=(dd& a, const dd& b) { a = a - b; return a; }
inline bool operator==(const dd& a, const dd& b) { return a.hi == b.hi && a.lo == b.lo; }
inline bool operator!=(const dd& a, const dd& b) { return !(a == b); }
inline dd sqrt(const dd& a) {
    if (a.hi <= 0.0) return dd(a.hi == 0.0 ? 0.0 : std::nan(""));
    const double x = 1.0 / std::sqrt(a.hi), ax = a.hi * x;      // Karp's trick: sqrt(a) ~ a x + (a - (a x)^2) x / 2
    const dd r = a - dd_two_prod(ax, ax);
    return dd_two_sum(ax, r.hi * (x * 0.5));
}
