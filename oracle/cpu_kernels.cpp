// CPU ORACLE -- test / baseline infrastructure only, never the product path.
//
// Plain C++17 + OpenMP restatement of the element assembly of oracle/rm_shell_oracle.py (itself a restatement of the
// reference's Reissner-Mindlin shell, femo_alpha/rm_shell/linear_shell_fenicsx/linear_shell_model.py:136-157,199-333,
// kinematics.py:12-106) for the "reference CPU path" column of the benchmark:
//
//   cpu_assemble_csr       K = sum_e K_e scattered into a given CSR pattern          (what dolfinx assemble_matrix +
//                          FFCx tabulate_tensor do for derivative(residual, w), fea/utils_dolfinx.py:200-206)
//   cpu_assemble_drdfield  the ndof x n_field matrices dR/dh, dR/dE, dR/dnu at a state w   (state_operation.py:283-286)
//   cpu_apply_K, cpu_drdfield_T, cpu_dcompliance   matrix-free quadrature sweeps of the adjoint chain (best-effort CPU column)
//   cpu_fronts_*           numeric phase of a multifrontal Cholesky on dense fronts with LAPACK/BLAS (the stand-in for
//                          MUMPS behind PETSc, fea/utils_dolfinx.py:466,514-531), driven level by level from Python
//
// Build: g++ -O3 -march=x86-64-v3 -fopenmp -shared -fPIC (oracle/cpu_baseline.py).  BLAS/LAPACK routines arrive as
// function pointers (scipy.linalg.cython_blas / cython_lapack capsules), Fortran calling convention.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include <omp.h>

#include "cpu_dd.h"

namespace {

constexpr double SHEAR_K = 0.833;          // linear_shell_model.py:146
constexpr int MAXLD = 39, MAXQ = 64;     // up to 8 x 8 Gauss points (the quadrature-convergence studies; the product kernels stop at 5 x 5)

struct Tables {
    int nq, nvc, npc;
    const double *N1, *dN1, *dN2, *w, *wS;   // (nq,nvc) (nq,nvc,2) (nq,npc,2) (nq) (nq)
};

}  // namespace

namespace f64 {
typedef double REAL;
#include "cpu_core.inc"
}  // namespace f64
namespace f80 {
typedef long double REAL;
#include "cpu_core.inc"
}  // namespace f80
namespace fdd {
typedef dd REAL;
#include "cpu_core.inc"
}  // namespace fdd
using namespace f64;

namespace {

inline void element_dofs(int e, int nvc, int npc, const int32_t* cells, const int32_t* cell_p2, int ndof_u, int* dofs) {
    for (int n = 0; n < npc; ++n)
        for (int c = 0; c < 3; ++c) dofs[3 * n + c] = 3 * cell_p2[(size_t)e * npc + n] + c;
    for (int b = 0; b < nvc; ++b)
        for (int c = 0; c < 3; ++c) dofs[3 * npc + 3 * b + c] = ndof_u + 3 * cells[(size_t)e * nvc + b] + c;
}

}  // namespace

extern "C" {

// fused element matrices + scatter into an existing CSR pattern (sorted column indices per row); atomics on the values
int cpu_assemble_csr(int nel, int nvc, int npc, int nq, const double* nodes, const int32_t* cells, const int32_t* cell_p2,
                     int ndof_u, const double* uhat, const double* N1, const double* dN1, const double* dN2, const double* w,
                     const double* wS, const double* h, const double* E, const double* nu, int ewm, const double* hK, int quad,
                     const int32_t* rowptr, const int32_t* colidx, double* vals, int nthreads) {
    const Tables T{nq, nvc, npc, N1, dN1, dN2, w, wS};
    const int ld = 3 * npc + 3 * nvc;
    if (ld > MAXLD || nq > MAXQ) return 1;
    int bad = 0;
#pragma omp parallel num_threads(nthreads)
    {
        std::vector<double> Ke((size_t)ld * ld);
        int dofs[MAXLD];
#pragma omp for schedule(static, 64)
        for (int e = 0; e < nel; ++e) {
            ElemIn el;
            load_elem(e, nvc, nodes, cells, uhat, h, E, nu, ewm, hK, el);
            element_matrix(T, el, quad != 0, 0, Ke.data());
            element_dofs(e, nvc, npc, cells, cell_p2, ndof_u, dofs);
            for (int i = 0; i < ld; ++i) {
                const int32_t* cb = colidx + rowptr[dofs[i]];
                const int32_t* ce = colidx + rowptr[dofs[i] + 1];
                for (int j = 0; j < ld; ++j) {
                    const int32_t* p = std::lower_bound(cb, ce, dofs[j]);
                    if (p == ce || *p != dofs[j]) { bad = 1; continue; }
#pragma omp atomic
                    vals[p - colidx] += Ke[(size_t)i * ld + j];
                }
            }
        }
    }
    return bad;
}

// ---- the operator of the GOLDENS: the same assembly with the element mathematics and the sums in an extended arithmetic
// (cpu_ext.inc): cpu_assemble_csr_ld / cpu_load_vector_ld in x87 long double, cpu_assemble_csr_dd / cpu_load_vector_dd in
// double-double (cpu_dd.h), the portable twin.
#define EXT_NS f80
#define EXT_REAL f80::REAL
#define EXT_NAME(x) x##_ld
#include "cpu_ext.inc"
#undef EXT_NS
#undef EXT_REAL
#undef EXT_NAME
#define EXT_NS fdd
#define EXT_REAL fdd::REAL
#define EXT_NAME(x) x##_dd
#include "cpu_ext.inc"
#undef EXT_NS
#undef EXT_REAL
#undef EXT_NAME

// r = b - K x with K, x, b in double-double ((n, 2) arrays of (hi, lo) pairs) and every row's sum carried in double-double: the residual
// of the goldens' refinement without x87 arithmetic.  r is returned rounded to double (it is the right-hand side of a float64 solve).
int cpu_csr_residual_dd(int64_t n, const int32_t* rowptr, const int32_t* colidx, const double* vals2, const double* x2, const double* b2,
                        double* r, int nthreads) {
    const dd* K = reinterpret_cast<const dd*>(vals2);
    const dd* x = reinterpret_cast<const dd*>(x2);
    const dd* b = reinterpret_cast<const dd*>(b2);
#pragma omp parallel for schedule(static, 1024) num_threads(nthreads)
    for (int64_t i = 0; i < n; ++i) {
        dd s = b[i];
        for (int32_t p = rowptr[i]; p < rowptr[i + 1]; ++p) s -= K[p] * x[colidx[p]];
        r[i] = (double)s;
    }
    return 0;
}

// x += dx (x in double-double, dx double)
int cpu_axpy_dd(int64_t n, double* x2, const double* dx) {
    dd* x = reinterpret_cast<dd*>(x2);
    for (int64_t i = 0; i < n; ++i) x[i] += dd(dx[i]);
    return 0;
}

// element matrices only (nel x ld x ld), for cross-checks against the numpy oracle
int cpu_element_matrices(int nel, int nvc, int npc, int nq, const double* nodes, const int32_t* cells, const double* uhat,
                         const double* N1, const double* dN1, const double* dN2, const double* w, const double* wS,
                         const double* h, const double* E, const double* nu, int ewm, const double* hK, int quad, int deriv,
                         double* Ke, int nthreads) {
    const Tables T{nq, nvc, npc, N1, dN1, dN2, w, wS};
    const int ld = 3 * npc + 3 * nvc;
    if (ld > MAXLD || nq > MAXQ) return 1;
#pragma omp parallel for schedule(static, 64) num_threads(nthreads)
    for (int e = 0; e < nel; ++e) {
        ElemIn el;
        load_elem(e, nvc, nodes, cells, uhat, h, E, nu, ewm, hK, el);
        element_matrix(T, el, quad != 0, deriv, Ke + (size_t)e * ld * ld);
    }
    return 0;
}

// dR/d field (field = h, E or nu: deriv 1..3) at the state w as COO blocks: per element ld x nfe values
// (nfe = nvc nodal values, or 1 for element-wise fields): out[e][i][b] = sum_q N1_b(q) (B^T C'_q B w_e)_i
int cpu_assemble_drdfield(int nel, int nvc, int npc, int nq, const double* nodes, const int32_t* cells, const int32_t* cell_p2,
                          int ndof_u, const double* uhat, const double* N1, const double* dN1, const double* dN2,
                          const double* w, const double* wS, const double* h, const double* E, const double* nu, int ewm,
                          const double* hK, int quad, int deriv, const double* state, double* out, int nthreads) {
    const Tables T{nq, nvc, npc, N1, dN1, dN2, w, wS};
    const int ld = 3 * npc + 3 * nvc, nfe = ewm ? 1 : nvc;
    if (ld > MAXLD || nq > MAXQ) return 1;
#pragma omp parallel for schedule(static, 64) num_threads(nthreads)
    for (int e = 0; e < nel; ++e) {
        ElemIn el;
        load_elem(e, nvc, nodes, cells, uhat, h, E, nu, ewm, hK, el);
        int dofs[MAXLD];
        element_dofs(e, nvc, npc, cells, cell_p2, ndof_u, dofs);
        double we[MAXLD], B[9][MAXLD];
        for (int i = 0; i < ld; ++i) we[i] = state[dofs[i]];
        double* o = out + (size_t)e * ld * nfe;
        std::memset(o, 0, sizeof(double) * ld * nfe);
        for (int q = 0; q < nq; ++q) {
            QP g;
            qp_B(T, q, el.X, el.has_u ? el.U : nullptr, quad != 0, B, g);
            double hq = 0, Eq = 0, nuq = 0;
            for (int b = 0; b < nvc; ++b) { const double N = N1[(size_t)q * nvc + b]; hq += N * el.h[b]; Eq += N * el.E[b]; nuq += N * el.nu[b]; }
            CQ C;
            qp_C(hq, Eq, nuq, el.hK, w[q] * g.det, wS[q] * g.det, g.Ju, deriv, C);
            double s[9] = {0}, t[9];
            for (int r = 0; r < 9; ++r)
                for (int k = 0; k < ld; ++k) s[r] += B[r][k] * we[k];
            for (int i = 0; i < 3; ++i) {
                t[i] = C.Cm[i][0] * s[0] + C.Cm[i][1] * s[1] + C.Cm[i][2] * s[2];
                t[3 + i] = C.Cb[i][0] * s[3] + C.Cb[i][1] * s[4] + C.Cb[i][2] * s[5];
            }
            t[6] = C.cs * s[6]; t[7] = C.cs * s[7]; t[8] = C.cd * s[8];
            for (int i = 0; i < ld; ++i) {
                double v = 0;
                for (int r = 0; r < 9; ++r) v += B[r][i] * t[r];
                for (int b = 0; b < nfe; ++b) o[(size_t)i * nfe + b] += v * (ewm ? 1.0 : N1[(size_t)q * nvc + b]);
            }
        }
    }
    return 0;
}

// load vector F_a = int f . N2_a J det dS (oracle load_vector, linear_shell_model.py:320): the vector assembly that stands
// for one residual evaluation of the reference's Newton loop
int cpu_load_vector(int nel, int nvc, int npc, int nq, const double* nodes, const int32_t* cells, const int32_t* cell_p2,
                    const double* uhat, const double* N1, const double* dN1, const double* dN2, const double* N2, const double* w,
                    const double* f, int ewp, int quad, double* F, int nthreads) {
    const Tables T{nq, nvc, npc, N1, dN1, dN2, w, w};
    if (3 * npc + 3 * nvc > MAXLD || nq > MAXQ) return 1;
#pragma omp parallel for schedule(static, 64) num_threads(nthreads)
    for (int e = 0; e < nel; ++e) {
        double X[4][3], U[4][3], fe[4][3], B[9][MAXLD];
        bool has_u = false;
        for (int b = 0; b < nvc; ++b) {
            const int v = cells[(size_t)e * nvc + b];
            for (int i = 0; i < 3; ++i) {
                X[b][i] = nodes[3 * (size_t)v + i];
                U[b][i] = uhat ? uhat[3 * (size_t)v + i] : 0.0;
                has_u = has_u || U[b][i] != 0.0;
                fe[b][i] = f[3 * (size_t)(ewp ? e : v) + i];
            }
        }
        double Fe[9][3] = {{0}};
        for (int q = 0; q < nq; ++q) {
            QP g;
            qp_B(T, q, X, has_u ? U : nullptr, quad != 0, B, g);          // (only det and Ju are needed here)
            double fq[3] = {0, 0, 0};
            for (int b = 0; b < nvc; ++b)
                for (int i = 0; i < 3; ++i) fq[i] += (ewp ? (b == 0 ? 1.0 : 0.0) : N1[(size_t)q * nvc + b]) * fe[b][i];
            const double wj = w[q] * g.det * g.Ju;
            for (int a = 0; a < npc; ++a)
                for (int i = 0; i < 3; ++i) Fe[a][i] += wj * N2[(size_t)q * npc + a] * fq[i];
        }
        for (int a = 0; a < npc; ++a)
            for (int i = 0; i < 3; ++i) {
#pragma omp atomic
                F[3 * (size_t)cell_p2[(size_t)e * npc + a] + i] += Fe[a][i];
            }
    }
    return 0;
}


// ---- matrix-free sweeps of the adjoint chain (no assembled matrix): what a best-effort CPU code would run
// y += K_elastic x, element by element through the strains (s = B x_e, t = C s, y_e = B^T t)
int cpu_apply_K(int nel, int nvc, int npc, int nq, const double* nodes, const int32_t* cells, const int32_t* cell_p2, int ndof_u,
                const double* uhat, const double* N1, const double* dN1, const double* dN2, const double* w, const double* wS,
                const double* h, const double* E, const double* nu, int ewm, const double* hK, int quad, const double* x, double* y,
                int nthreads) {
    const Tables T{nq, nvc, npc, N1, dN1, dN2, w, wS};
    const int ld = 3 * npc + 3 * nvc;
    if (ld > MAXLD || nq > MAXQ) return 1;
#pragma omp parallel for schedule(static, 64) num_threads(nthreads)
    for (int e = 0; e < nel; ++e) {
        ElemIn el;
        load_elem(e, nvc, nodes, cells, uhat, h, E, nu, ewm, hK, el);
        int dofs[MAXLD];
        element_dofs(e, nvc, npc, cells, cell_p2, ndof_u, dofs);
        double xe[MAXLD], ye[MAXLD] = {0}, B[9][MAXLD];
        for (int i = 0; i < ld; ++i) xe[i] = x[dofs[i]];
        for (int q = 0; q < nq; ++q) {
            QP g;
            qp_B(T, q, el.X, el.has_u ? el.U : nullptr, quad != 0, B, g);
            double hq = 0, Eq = 0, nuq = 0;
            for (int b = 0; b < nvc; ++b) { const double N = N1[(size_t)q * nvc + b]; hq += N * el.h[b]; Eq += N * el.E[b]; nuq += N * el.nu[b]; }
            CQ C;
            qp_C(hq, Eq, nuq, el.hK, w[q] * g.det, wS[q] * g.det, g.Ju, 0, C);
            double s[9] = {0}, t[9];
            for (int r = 0; r < 9; ++r)
                for (int k = 0; k < ld; ++k) s[r] += B[r][k] * xe[k];
            for (int i = 0; i < 3; ++i) {
                t[i] = C.Cm[i][0] * s[0] + C.Cm[i][1] * s[1] + C.Cm[i][2] * s[2];
                t[3 + i] = C.Cb[i][0] * s[3] + C.Cb[i][1] * s[4] + C.Cb[i][2] * s[5];
            }
            t[6] = C.cs * s[6]; t[7] = C.cs * s[7]; t[8] = C.cd * s[8];
            for (int r = 0; r < 9; ++r)
                for (int k = 0; k < ld; ++k) ye[k] += B[r][k] * t[r];
        }
        for (int i = 0; i < ld; ++i) {
#pragma omp atomic
            y[dofs[i]] += ye[i];
        }
    }
    return 0;
}

// out (n_field) += scale * (dR/d field)^T lambda at the state: sum_q N1_b(q) (B lambda_e) . C'_q (B w_e)   (deriv 1 h, 2 E, 3 nu)
int cpu_drdfield_T(int nel, int nvc, int npc, int nq, const double* nodes, const int32_t* cells, const int32_t* cell_p2, int ndof_u,
                   const double* uhat, const double* N1, const double* dN1, const double* dN2, const double* w, const double* wS,
                   const double* h, const double* E, const double* nu, int ewm, const double* hK, int quad, int deriv,
                   const double* state, const double* lam, double scale, double* out, int nthreads) {
    const Tables T{nq, nvc, npc, N1, dN1, dN2, w, wS};
    const int ld = 3 * npc + 3 * nvc;
    if (ld > MAXLD || nq > MAXQ) return 1;
#pragma omp parallel for schedule(static, 64) num_threads(nthreads)
    for (int e = 0; e < nel; ++e) {
        ElemIn el;
        load_elem(e, nvc, nodes, cells, uhat, h, E, nu, ewm, hK, el);
        int dofs[MAXLD];
        element_dofs(e, nvc, npc, cells, cell_p2, ndof_u, dofs);
        double we[MAXLD], le[MAXLD], B[9][MAXLD], ge[4] = {0, 0, 0, 0};
        for (int i = 0; i < ld; ++i) { we[i] = state[dofs[i]]; le[i] = lam[dofs[i]]; }
        for (int q = 0; q < nq; ++q) {
            QP g;
            qp_B(T, q, el.X, el.has_u ? el.U : nullptr, quad != 0, B, g);
            double hq = 0, Eq = 0, nuq = 0;
            for (int b = 0; b < nvc; ++b) { const double N = N1[(size_t)q * nvc + b]; hq += N * el.h[b]; Eq += N * el.E[b]; nuq += N * el.nu[b]; }
            CQ C;
            qp_C(hq, Eq, nuq, el.hK, w[q] * g.det, wS[q] * g.det, g.Ju, deriv, C);
            double s[9] = {0}, l[9] = {0}, t[9];
            for (int r = 0; r < 9; ++r)
                for (int k = 0; k < ld; ++k) { s[r] += B[r][k] * we[k]; l[r] += B[r][k] * le[k]; }
            for (int i = 0; i < 3; ++i) {
                t[i] = C.Cm[i][0] * s[0] + C.Cm[i][1] * s[1] + C.Cm[i][2] * s[2];
                t[3 + i] = C.Cb[i][0] * s[3] + C.Cb[i][1] * s[4] + C.Cb[i][2] * s[5];
            }
            t[6] = C.cs * s[6]; t[7] = C.cs * s[7]; t[8] = C.cd * s[8];
            double dens = 0;
            for (int r = 0; r < 9; ++r) dens += l[r] * t[r];
            if (ewm) ge[0] += dens;
            else for (int b = 0; b < nvc; ++b) ge[b] += dens * N1[(size_t)q * nvc + b];
        }
        if (ewm) out[e] += scale * ge[0];
        else for (int b = 0; b < nvc; ++b) {
#pragma omp atomic
            out[cells[(size_t)e * nvc + b]] += scale * ge[b];
        }
    }
    return 0;
}

// out (ndof) += d/du int u.u J dx = 2 int N2_a u J det   (rm_shell_pde.py:85-89), and
// gh (n_h) += alpha int grad h . grad M_b det            (the regularisation's thickness gradient, nodal thickness, :72-74)
int cpu_dcompliance(int nel, int nvc, int npc, int nq, const double* nodes, const int32_t* cells, const int32_t* cell_p2,
                    const double* uhat, const double* N1, const double* dN1, const double* dN2, const double* N2, const double* w,
                    const double* h, int ewm, double alpha, int quad, const double* state, double* out, double* gh, int nthreads) {
    const Tables T{nq, nvc, npc, N1, dN1, dN2, w, w};
    if (3 * npc + 3 * nvc > MAXLD || nq > MAXQ) return 1;
#pragma omp parallel for schedule(static, 64) num_threads(nthreads)
    for (int e = 0; e < nel; ++e) {
        double X[4][3], U[4][3], he[4], B[9][MAXLD], ue[9][3], Fe[9][3] = {{0}}, ge[4] = {0, 0, 0, 0};
        bool has_u = false;
        double area = 0;
        for (int b = 0; b < nvc; ++b) {
            const int v = cells[(size_t)e * nvc + b];
            for (int i = 0; i < 3; ++i) {
                X[b][i] = nodes[3 * (size_t)v + i];
                U[b][i] = uhat ? uhat[3 * (size_t)v + i] : 0.0;
                has_u = has_u || U[b][i] != 0.0;
            }
            he[b] = h[ewm ? e : v];
        }
        for (int a = 0; a < npc; ++a)
            for (int i = 0; i < 3; ++i) ue[a][i] = state[3 * (size_t)cell_p2[(size_t)e * npc + a] + i];
        for (int q = 0; q < nq; ++q) {
            QP g;
            qp_B(T, q, X, has_u ? U : nullptr, quad != 0, B, g);
            double uq[3] = {0, 0, 0};
            for (int a = 0; a < npc; ++a)
                for (int i = 0; i < 3; ++i) uq[i] += N2[(size_t)q * npc + a] * ue[a][i];
            const double wj = 2.0 * w[q] * g.det * g.Ju;
            for (int a = 0; a < npc; ++a)
                for (int i = 0; i < 3; ++i) Fe[a][i] += wj * N2[(size_t)q * npc + a] * uq[i];
            if (ewm) { area += w[q] * g.det; continue; }
            // surface gradient of the P1 functions (reference configuration): grad M_b = Kinv^T dN1_b
            const double* d1 = dN1 + (size_t)q * nvc * 2;
            double J0[3] = {0, 0, 0}, J1[3] = {0, 0, 0};
            for (int b = 0; b < nvc; ++b)
                for (int i = 0; i < 3; ++i) { J0[i] += X[b][i] * d1[2 * b]; J1[i] += X[b][i] * d1[2 * b + 1]; }
            const double g00 = dot3(J0, J0), g01 = dot3(J0, J1), g11 = dot3(J1, J1), gd = g00 * g11 - g01 * g01;
            double gM[4][3], grad_h[3] = {0, 0, 0};
            for (int b = 0; b < nvc; ++b)
                for (int i = 0; i < 3; ++i) {
                    gM[b][i] = ((g11 * J0[i] - g01 * J1[i]) * d1[2 * b] + (-g01 * J0[i] + g00 * J1[i]) * d1[2 * b + 1]) / gd;
                    grad_h[i] += gM[b][i] * he[b];
                }
            for (int b = 0; b < nvc; ++b) ge[b] += alpha * w[q] * g.det * dot3(gM[b], grad_h);
        }
        for (int a = 0; a < npc; ++a)
            for (int i = 0; i < 3; ++i) {
#pragma omp atomic
                out[3 * (size_t)cell_p2[(size_t)e * npc + a] + i] += Fe[a][i];
            }
        if (ewm) gh[e] += alpha * he[0] * area;
        else for (int b = 0; b < nvc; ++b) {
#pragma omp atomic
            gh[cells[(size_t)e * nvc + b]] += ge[b];
        }
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ multifrontal numeric phase
typedef void (*dpotrf_t)(const char*, const int*, double*, const int*, int*);
typedef void (*dtrsm_t)(const char*, const char*, const char*, const char*, const int*, const int*, const double*, const double*,
                        const int*, double*, const int*);
typedef void (*dsyrk_t)(const char*, const char*, const int*, const int*, const double*, const double*, const int*, const double*,
                        double*, const int*);

struct FrontPlan {
    int ntree;
    const int32_t *nf, *npiv, *parent, *left, *right, *front_dofs, *up_map, *elem_front, *elem_map;
    const int64_t *front_off, *dof_off;
};

// leaf fronts: zero, then add the element matrices of their cells (elements grouped by front: elem_order / elem_start)
int cpu_fronts_assemble(int nlist, const int32_t* fronts, const int32_t* nf, const int64_t* front_off, const int32_t* elem_start,
                        const int32_t* elem_order, const int32_t* elem_map, int nvc, int npc, int nq, const double* nodes,
                        const int32_t* cells, const double* uhat, const double* N1, const double* dN1, const double* dN2,
                        const double* w, const double* wS, const double* h, const double* E, const double* nu, int ewm, const double* hK,
                        int quad, double* F, int nthreads) {
    const Tables T{nq, nvc, npc, N1, dN1, dN2, w, wS};
    const int ld = 3 * npc + 3 * nvc;
    if (ld > MAXLD || nq > MAXQ) return 1;
#pragma omp parallel num_threads(nthreads)
    {
        std::vector<double> Ke((size_t)ld * ld);
#pragma omp for schedule(dynamic, 8)
        for (int i = 0; i < nlist; ++i) {
            const int t = fronts[i], n = nf[t];
            double* Ft = F + front_off[t];
            std::memset(Ft, 0, sizeof(double) * (size_t)n * n);
            for (int k = elem_start[t]; k < elem_start[t + 1]; ++k) {
                const int e = elem_order[k];
                ElemIn el;
                load_elem(e, nvc, nodes, cells, uhat, h, E, nu, ewm, hK, el);
                element_matrix(T, el, quad != 0, 0, Ke.data());
                const int32_t* map = elem_map + (size_t)e * ld;
                for (int a = 0; a < ld; ++a)
                    for (int b = 0; b < ld; ++b)
                        if (map[a] >= map[b]) Ft[map[a] + (size_t)n * map[b]] += Ke[(size_t)a * ld + b];     // lower triangle
            }
        }
    }
    return 0;
}

// ---- the step operator of the transient path, A = aK K + aM M (dynamic_rm_shell/plate_sim.py:131-140,190-215: midpoint rule,
// aK = 1/2, aM = 2/dt^2), for the CPU column of BASELINE config 5.  M is the inertia of oracle assemble_M:
// rho h (u.v + h_K^2 theta.eta) J dx (linear_shell_model.py:335-348).
namespace {
// Ke += aM * M_e
void element_mass_add(const Tables& T, const double* N2, const ElemIn& el, const double* rho_e, bool quad, double aM, double* Ke) {
    const int ld = 3 * T.npc + 3 * T.nvc;
    double B[9][MAXLD];
    for (int q = 0; q < T.nq; ++q) {
        if (T.w[q] == 0.0) continue;
        QP g;
        qp_B(T, q, el.X, el.has_u ? el.U : nullptr, quad, B, g);
        double hq = 0, rq = 0;
        for (int b = 0; b < T.nvc; ++b) { const double N = T.N1[(size_t)q * T.nvc + b]; hq += N * el.h[b]; rq += N * rho_e[b]; }
        const double cm = aM * T.w[q] * g.det * g.Ju * rq * hq, ct = cm * el.hK * el.hK;
        const double* n2 = N2 + (size_t)q * T.npc;
        const double* n1 = T.N1 + (size_t)q * T.nvc;
        for (int a = 0; a < T.npc; ++a)
            for (int b = 0; b < T.npc; ++b) {
                const double v = cm * n2[a] * n2[b];
                for (int c = 0; c < 3; ++c) Ke[(size_t)(3 * a + c) * ld + 3 * b + c] += v;
            }
        for (int a = 0; a < T.nvc; ++a)
            for (int b = 0; b < T.nvc; ++b) {
                const double v = ct * n1[a] * n1[b];
                for (int c = 0; c < 3; ++c) Ke[(size_t)(3 * T.npc + 3 * a + c) * ld + 3 * T.npc + 3 * b + c] += v;
            }
    }
}
inline void load_rho(int e, int nvc, const int32_t* cells, const double* rho, int ewm, double* rho_e) {
    for (int b = 0; b < nvc; ++b) rho_e[b] = rho[ewm ? e : cells[(size_t)e * nvc + b]];
}
}  // namespace

// leaf fronts of A = aK K + aM M (cpu_fronts_assemble with the inertia added)
int cpu_fronts_assemble_op(int nlist, const int32_t* fronts, const int32_t* nf, const int64_t* front_off, const int32_t* elem_start,
                           const int32_t* elem_order, const int32_t* elem_map, int nvc, int npc, int nq, const double* nodes,
                           const int32_t* cells, const double* uhat, const double* N1, const double* dN1, const double* dN2,
                           const double* w, const double* wS, const double* h, const double* E, const double* nu, int ewm, const double* hK,
                           int quad, const double* N2, const double* rho, double aK, double aM, double* F, int nthreads) {
    const Tables T{nq, nvc, npc, N1, dN1, dN2, w, wS};
    const int ld = 3 * npc + 3 * nvc;
    if (ld > MAXLD || nq > MAXQ) return 1;
#pragma omp parallel num_threads(nthreads)
    {
        std::vector<double> Ke((size_t)ld * ld);
#pragma omp for schedule(dynamic, 8)
        for (int i = 0; i < nlist; ++i) {
            const int t = fronts[i], n = nf[t];
            double* Ft = F + front_off[t];
            std::memset(Ft, 0, sizeof(double) * (size_t)n * n);
            for (int k = elem_start[t]; k < elem_start[t + 1]; ++k) {
                const int e = elem_order[k];
                ElemIn el;
                double rho_e[4];
                load_elem(e, nvc, nodes, cells, uhat, h, E, nu, ewm, hK, el);
                load_rho(e, nvc, cells, rho, ewm, rho_e);
                element_matrix(T, el, quad != 0, 0, Ke.data());
                if (aK != 1.0) for (double& v : Ke) v *= aK;
                if (aM != 0.0) element_mass_add(T, N2, el, rho_e, quad != 0, aM, Ke.data());
                const int32_t* map = elem_map + (size_t)e * ld;
                for (int a = 0; a < ld; ++a)
                    for (int b = 0; b < ld; ++b)
                        if (map[a] >= map[b]) Ft[map[a] + (size_t)n * map[b]] += Ke[(size_t)a * ld + b];     // lower triangle
            }
        }
    }
    return 0;
}

// y += (aK K + aM M) x, element by element (cpu_apply_K with the inertia added)
int cpu_apply_op(int nel, int nvc, int npc, int nq, const double* nodes, const int32_t* cells, const int32_t* cell_p2, int ndof_u,
                 const double* uhat, const double* N1, const double* dN1, const double* dN2, const double* w, const double* wS,
                 const double* h, const double* E, const double* nu, int ewm, const double* hK, int quad, const double* N2,
                 const double* rho, double aK, double aM, const double* x, double* y, int nthreads) {
    const Tables T{nq, nvc, npc, N1, dN1, dN2, w, wS};
    const int ld = 3 * npc + 3 * nvc;
    if (ld > MAXLD || nq > MAXQ) return 1;
#pragma omp parallel for schedule(static, 64) num_threads(nthreads)
    for (int e = 0; e < nel; ++e) {
        ElemIn el;
        double rho_e[4];
        load_elem(e, nvc, nodes, cells, uhat, h, E, nu, ewm, hK, el);
        load_rho(e, nvc, cells, rho, ewm, rho_e);
        int dofs[MAXLD];
        element_dofs(e, nvc, npc, cells, cell_p2, ndof_u, dofs);
        double xe[MAXLD], ye[MAXLD] = {0}, B[9][MAXLD];
        for (int i = 0; i < ld; ++i) xe[i] = x[dofs[i]];
        for (int q = 0; q < nq; ++q) {
            QP g;
            qp_B(T, q, el.X, el.has_u ? el.U : nullptr, quad != 0, B, g);
            double hq = 0, Eq = 0, nuq = 0, rq = 0;
            for (int b = 0; b < nvc; ++b) {
                const double N = N1[(size_t)q * nvc + b];
                hq += N * el.h[b]; Eq += N * el.E[b]; nuq += N * el.nu[b]; rq += N * rho_e[b];
            }
            if (aK != 0.0) {
                CQ C;
                qp_C(hq, Eq, nuq, el.hK, w[q] * g.det, wS[q] * g.det, g.Ju, 0, C);
                double s[9] = {0}, t[9];
                for (int r = 0; r < 9; ++r)
                    for (int k = 0; k < ld; ++k) s[r] += B[r][k] * xe[k];
                for (int i = 0; i < 3; ++i) {
                    t[i] = C.Cm[i][0] * s[0] + C.Cm[i][1] * s[1] + C.Cm[i][2] * s[2];
                    t[3 + i] = C.Cb[i][0] * s[3] + C.Cb[i][1] * s[4] + C.Cb[i][2] * s[5];
                }
                t[6] = C.cs * s[6]; t[7] = C.cs * s[7]; t[8] = C.cd * s[8];
                for (int r = 0; r < 9; ++r)
                    for (int k = 0; k < ld; ++k) ye[k] += aK * B[r][k] * t[r];
            }
            if (aM != 0.0 && w[q] != 0.0) {
                const double cm = aM * w[q] * g.det * g.Ju * rq * hq, ct = cm * el.hK * el.hK;
                const double* n2 = N2 + (size_t)q * npc;
                const double* n1 = N1 + (size_t)q * nvc;
                double uq[3] = {0, 0, 0}, tq[3] = {0, 0, 0};
                for (int a = 0; a < npc; ++a)
                    for (int c = 0; c < 3; ++c) uq[c] += n2[a] * xe[3 * a + c];
                for (int b = 0; b < nvc; ++b)
                    for (int c = 0; c < 3; ++c) tq[c] += n1[b] * xe[3 * npc + 3 * b + c];
                for (int a = 0; a < npc; ++a)
                    for (int c = 0; c < 3; ++c) ye[3 * a + c] += cm * n2[a] * uq[c];
                for (int b = 0; b < nvc; ++b)
                    for (int c = 0; c < 3; ++c) ye[3 * npc + 3 * b + c] += ct * n1[b] * tq[c];
            }
        }
        for (int i = 0; i < ld; ++i) {
#pragma omp atomic
            y[dofs[i]] += ye[i];
        }
    }
    return 0;
}

// one tree level: [extend-add of the children] + partial Cholesky (potrf on the pivot block, trsm, syrk) per front
int cpu_fronts_factor_level(int nlist, const int32_t* fronts, const int32_t* nf, const int32_t* npiv, const int64_t* front_off,
                            const int64_t* dof_off, const int32_t* left, const int32_t* right, const int32_t* up_map, double* F,
                            void* potrf_p, void* trsm_p, void* syrk_p, int nthreads) {
    const dpotrf_t potrf = (dpotrf_t)potrf_p;
    const dtrsm_t trsm = (dtrsm_t)trsm_p;
    const dsyrk_t syrk = (dsyrk_t)syrk_p;
    int fail = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int i = 0; i < nlist; ++i) {
        const int t = fronts[i], n = nf[t], np = npiv[t], nb = n - np;
        double* Ft = F + front_off[t];
        const int ch[2] = {left[t], right[t]};
        if (ch[0] >= 0 || ch[1] >= 0) {
            std::memset(Ft, 0, sizeof(double) * (size_t)n * n);
            for (int s = 0; s < 2; ++s) {
                const int c = ch[s];
                if (c < 0) continue;
                const int nc = nf[c], pc = npiv[c];
                const double* Fc = F + front_off[c];
                const int32_t* um = up_map + dof_off[c];
                for (int b = pc; b < nc; ++b) {
                    const int pb = um[b];
                    for (int a = b; a < nc; ++a) {
                        const int pa = um[a];
                        const double v = Fc[a + (size_t)nc * b];
                        if (pa >= pb) Ft[pa + (size_t)n * pb] += v; else Ft[pb + (size_t)n * pa] += v;
                    }
                }
            }
        }
        if (np == 0) continue;
        int info = 0;
        const double one = 1.0, mone = -1.0;
        potrf("L", &np, Ft, &n, &info);
        if (info) { fail = 1; continue; }
        if (nb > 0) {
            trsm("R", "L", "T", "N", &nb, &np, &one, Ft, &n, Ft + np, &n);
            syrk("L", "N", &nb, &np, &mone, Ft + np, &n, &one, Ft + np + (size_t)n * np, &n);
        }
    }
    return fail;
}

// triangular sweeps with the factor: forward over the listed fronts (children first), backward (parents first)
typedef void (*dtrsv_t)(const char*, const char*, const char*, const int*, const double*, const int*, double*, const int*);
typedef void (*dgemv_t)(const char*, const int*, const int*, const double*, const double*, const int*, const double*, const int*,
                        const double*, double*, const int*);
int cpu_fronts_solve(int ntree, const int32_t* order, const int32_t* nf, const int32_t* npiv, const int64_t* front_off,
                     const int64_t* dof_off, const int32_t* front_dofs, const double* F, double* x, void* trsv_p, void* gemv_p) {
    const dtrsv_t trsv = (dtrsv_t)trsv_p;
    const dgemv_t gemv = (dgemv_t)gemv_p;
    std::vector<double> buf;
    const int inc = 1;
    const double one = 1.0, mone = -1.0;
    for (int k = 0; k < ntree; ++k) {            // forward
        const int t = order[k], n = nf[t], np = npiv[t], nb = n - np;
        if (np == 0) continue;
        const int32_t* gd = front_dofs + dof_off[t];
        const double* Ft = F + front_off[t];
        buf.resize(n);
        for (int i = 0; i < n; ++i) buf[i] = i < np ? x[gd[i]] : 0.0;
        trsv("L", "N", "N", &np, Ft, &n, buf.data(), &inc);
        if (nb > 0) gemv("N", &nb, &np, &mone, Ft + np, &n, buf.data(), &inc, &one, buf.data() + np, &inc);
        for (int i = 0; i < np; ++i) x[gd[i]] = buf[i];
        for (int i = np; i < n; ++i) x[gd[i]] += buf[i];
    }
    for (int k = ntree - 1; k >= 0; --k) {       // backward
        const int t = order[k], n = nf[t], np = npiv[t], nb = n - np;
        if (np == 0) continue;
        const int32_t* gd = front_dofs + dof_off[t];
        const double* Ft = F + front_off[t];
        buf.resize(n);
        for (int i = 0; i < n; ++i) buf[i] = x[gd[i]];
        if (nb > 0) gemv("T", &nb, &np, &mone, Ft + np, &n, buf.data() + np, &inc, &one, buf.data(), &inc);
        trsv("L", "T", "N", &np, Ft, &n, buf.data(), &inc);
        for (int i = 0; i < np; ++i) x[gd[i]] = buf[i];
    }
    return 0;
}


// the same sweeps level by level: the fronts of a tree level are independent (a forward sweep adds into the boundary
// entries of x, which siblings may share: atomics), so the levels with many fronts run one front per thread; the few large
// fronts at the top run one after the other (their BLAS calls may thread).  level_off: nlevels + 1 offsets into order.
int cpu_fronts_solve_levels(int nlevels, const int32_t* level_off, const int32_t* order, const int32_t* nf, const int32_t* npiv,
                            const int64_t* front_off, const int64_t* dof_off, const int32_t* front_dofs, const double* F, double* x,
                            void* trsv_p, void* gemv_p, int nthreads) {
    const dtrsv_t trsv = (dtrsv_t)trsv_p;
    const dgemv_t gemv = (dgemv_t)gemv_p;
    const int inc = 1;
    const double one = 1.0, mone = -1.0;
    for (int L = 0; L < nlevels; ++L) {                 // forward
        const int b0 = level_off[L], cnt = level_off[L + 1] - b0;
#pragma omp parallel num_threads(cnt >= 2 * nthreads ? nthreads : 1)
        {
            std::vector<double> buf;
#pragma omp for schedule(dynamic, 4)
            for (int k = 0; k < cnt; ++k) {
                const int t = order[b0 + k], n = nf[t], np = npiv[t], nb = n - np;
                if (np == 0) continue;
                const int32_t* gd = front_dofs + dof_off[t];
                const double* Ft = F + front_off[t];
                buf.assign(n, 0.0);
                for (int i = 0; i < np; ++i) buf[i] = x[gd[i]];
                trsv("L", "N", "N", &np, Ft, &n, buf.data(), &inc);
                if (nb > 0) gemv("N", &nb, &np, &mone, Ft + np, &n, buf.data(), &inc, &one, buf.data() + np, &inc);
                for (int i = 0; i < np; ++i) x[gd[i]] = buf[i];
                for (int i = np; i < n; ++i) {
#pragma omp atomic
                    x[gd[i]] += buf[i];
                }
            }
        }
    }
    for (int L = nlevels - 1; L >= 0; --L) {            // backward
        const int b0 = level_off[L], cnt = level_off[L + 1] - b0;
#pragma omp parallel num_threads(cnt >= 2 * nthreads ? nthreads : 1)
        {
            std::vector<double> buf;
#pragma omp for schedule(dynamic, 4)
            for (int k = 0; k < cnt; ++k) {
                const int t = order[b0 + k], n = nf[t], np = npiv[t], nb = n - np;
                if (np == 0) continue;
                const int32_t* gd = front_dofs + dof_off[t];
                const double* Ft = F + front_off[t];
                buf.resize(n);
                for (int i = 0; i < n; ++i) buf[i] = x[gd[i]];
                if (nb > 0) gemv("T", &nb, &np, &mone, Ft + np, &n, buf.data() + np, &inc, &one, buf.data(), &inc);
                trsv("L", "T", "N", &np, Ft, &n, buf.data(), &inc);
                for (int i = 0; i < np; ++i) x[gd[i]] = buf[i];
            }
        }
    }
    return 0;
}

int cpu_max_threads(void) { return omp_get_max_threads(); }

}  // extern "C"
