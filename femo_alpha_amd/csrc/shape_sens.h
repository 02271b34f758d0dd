// Derivatives with respect to the mesh motion uhat (shape sensitivities), gfx950, fp64.
//
// The reference obtains them symbolically: derivative(form, uhat_function) for the outputs
// (reference femo_alpha/csdl_alpha_opt/output_operation.py:58-69) and assembleMatrix(dR/duhat) followed
// by a transposed mat-vec for the state (csdl_alpha_opt/state_operation.py:180-184, 283-286).  uhat enters
// through F = I + grad(uhat), gradx = grad . F^-1 and J = det F (linear_shell_fenicsx/kinematics.py:12-44)
// and through Nanson's formula in the penalty term (linear_shell_model.py:329-333).
//
// Here every quantity that depends on uhat is evaluated in forward-mode dual arithmetic (value + one
// directional derivative); one thread handles one (element, uhat component) pair -- 12 per quad -- and
// differentiates the element's scalar  Phi_e(uhat) = lambda_e . R_e(w_e; uhat)  or the element's share of
// a functional.  No matrix is formed.
#pragma once
#include "shell_device.h"

namespace femo {

struct D1 {
    double v, d;
};
__device__ __forceinline__ D1 mk(double v, double d = 0.0) { D1 r; r.v = v; r.d = d; return r; }
__device__ __forceinline__ D1 operator+(D1 a, D1 b) { return mk(a.v + b.v, a.d + b.d); }
__device__ __forceinline__ D1 operator-(D1 a, D1 b) { return mk(a.v - b.v, a.d - b.d); }
__device__ __forceinline__ D1 operator-(D1 a) { return mk(-a.v, -a.d); }
__device__ __forceinline__ D1 operator*(D1 a, D1 b) { return mk(a.v * b.v, a.d * b.v + a.v * b.d); }
__device__ __forceinline__ D1 operator*(double a, D1 b) { return mk(a * b.v, a * b.d); }
__device__ __forceinline__ D1 operator*(D1 a, double b) { return mk(a.v * b, a.d * b); }
__device__ __forceinline__ D1 operator+(D1 a, double b) { return mk(a.v + b, a.d); }
__device__ __forceinline__ D1 operator+(double a, D1 b) { return mk(a + b.v, b.d); }
__device__ __forceinline__ D1 operator/(D1 a, D1 b) {
    const double q = a.v / b.v;
    return mk(q, (a.d - q * b.d) / b.v);
}
__device__ __forceinline__ D1 dsqrt(D1 a) {
    const double s = sqrt(a.v);
    return mk(s, 0.5 * a.d / s);
}
__device__ __forceinline__ D1 ddot3(const double* a, const D1* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// uhat-dependent part of the quadrature-point geometry
struct QPD {
    D1 Q[2][2];
    D1 w0[3], w1[3];
    D1 Ju;
    D1 cof[3][3];      // cofactor matrix of F (= J F^-T), used by the penalty term
};

// base: geometry without mesh motion (E0,E1,E2,Q0,det from qp_geometry<.., false>); Uh: nodal uhat as duals
template <int NVC, bool QUAD>
__device__ __forceinline__ void qp_shape_dual(const double (*X)[3], const D1 (*Uh)[3], const double (*dM)[2], const QPG& g,
                                              QPD& s) {
    D1 F[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) F[i][j] = mk(i == j ? 1.0 : 0.0);
    for (int b = 0; b < NVC; ++b) {
        const double d0 = dM[b][0] * g.Q0[0][0] + dM[b][1] * g.Q0[1][0];
        const double d1 = dM[b][0] * g.Q0[0][1] + dM[b][1] * g.Q0[1][1];
        for (int j = 0; j < 3; ++j) {
            const double gm = d0 * g.E0[j] + d1 * g.E1[j];
            for (int i = 0; i < 3; ++i) F[i][j] = F[i][j] + Uh[b][i] * gm;
        }
    }
    D1(*C)[3] = s.cof;
    C[0][0] = F[1][1] * F[2][2] - F[1][2] * F[2][1];
    C[0][1] = F[1][2] * F[2][0] - F[1][0] * F[2][2];
    C[0][2] = F[1][0] * F[2][1] - F[1][1] * F[2][0];
    C[1][0] = F[0][2] * F[2][1] - F[0][1] * F[2][2];
    C[1][1] = F[0][0] * F[2][2] - F[0][2] * F[2][0];
    C[1][2] = F[0][1] * F[2][0] - F[0][0] * F[2][1];
    C[2][0] = F[0][1] * F[1][2] - F[0][2] * F[1][1];
    C[2][1] = F[0][2] * F[1][0] - F[0][0] * F[1][2];
    C[2][2] = F[0][0] * F[1][1] - F[0][1] * F[1][0];
    s.Ju = F[0][0] * C[0][0] + F[0][1] * C[0][1] + F[0][2] * C[0][2];
    // S[b][a] = E_b . F^-1 E_a,  F^-1[k][j] = C[j][k] / Ju
    D1 FE0[3], FE1[3];
    for (int k = 0; k < 3; ++k) {
        FE0[k] = (C[0][k] * g.E0[0] + C[1][k] * g.E0[1] + C[2][k] * g.E0[2]) / s.Ju;
        FE1[k] = (C[0][k] * g.E1[0] + C[1][k] * g.E1[1] + C[2][k] * g.E1[2]) / s.Ju;
    }
    const D1 S00 = ddot3(g.E0, FE0), S01 = ddot3(g.E0, FE1), S10 = ddot3(g.E1, FE0), S11 = ddot3(g.E1, FE1);
    for (int i = 0; i < 2; ++i) {
        s.Q[i][0] = g.Q0[i][0] * S00 + g.Q0[i][1] * S10;
        s.Q[i][1] = g.Q0[i][0] * S01 + g.Q0[i][1] * S11;
    }
    for (int c = 0; c < 3; ++c) s.w0[c] = s.w1[c] = mk(0.0);
    if (QUAD) {
        double J0[3] = {0, 0, 0}, J1[3] = {0, 0, 0}, tw[3];
        for (int b = 0; b < NVC; ++b)
            for (int c = 0; c < 3; ++c) {
                J0[c] += X[b][c] * dM[b][0];
                J1[c] += X[b][c] * dM[b][1];
            }
        for (int c = 0; c < 3; ++c) tw[c] = 0.25 * (X[0][c] - X[1][c] + X[2][c] - X[3][c]);
        double da0[3], da1[3];
        cross3(J0, tw, da0);
        cross3(tw, J1, da1);
        const double p0 = dot3(g.E2, da0), p1 = dot3(g.E2, da1);
        for (int c = 0; c < 3; ++c) {
            const double dn0 = (da0[c] - g.E2[c] * p0) / g.det, dn1 = (da1[c] - g.E2[c] * p1) / g.det;
            s.w0[c] = dn0 * s.Q[0][0] + dn1 * s.Q[1][0];
            s.w1[c] = dn0 * s.Q[0][1] + dn1 * s.Q[1][1];
        }
    }
}

struct GenD {
    D1 e00, e11, g01, k00, k11, k01, ga0, ga1, om;
};

__device__ __forceinline__ void dcross(const double* a, const D1* b, D1* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

template <int NPC, int NVC>
__device__ __forceinline__ GenD strains_dual(const Tables& t, int q, const QPG& g, const QPD& s, const double* xe) {
    D1 G0[3], G1[3], th[3], T0[3], T1[3];
    for (int c = 0; c < 3; ++c) G0[c] = G1[c] = th[c] = T0[c] = T1[c] = mk(0.0);
    for (int a = 0; a < NPC; ++a) {
        const double r0 = t.dN2[q][a][0], r1 = t.dN2[q][a][1];
        const D1 d0 = r0 * s.Q[0][0] + r1 * s.Q[1][0], d1 = r0 * s.Q[0][1] + r1 * s.Q[1][1];
        for (int c = 0; c < 3; ++c) {
            G0[c] = G0[c] + xe[3 * a + c] * d0;
            G1[c] = G1[c] + xe[3 * a + c] * d1;
        }
    }
    for (int b = 0; b < NVC; ++b) {
        const double r0 = t.dNR[q][b][0], r1 = t.dNR[q][b][1];          // the rotation's shape functions (N1 itself except for CG2CR1)
        const D1 m0 = r0 * s.Q[0][0] + r1 * s.Q[1][0], m1 = r0 * s.Q[0][1] + r1 * s.Q[1][1];
        const double Mb = t.NR[q][b];
        for (int c = 0; c < 3; ++c) {
            const double v = xe[3 * NPC + 3 * b + c];
            th[c] = th[c] + mk(Mb * v);
            T0[c] = T0[c] + v * m0;
            T1[c] = T1[c] + v * m1;
        }
    }
    GenD r;
    const D1 t00 = ddot3(g.E0, G0), t01 = ddot3(g.E0, G1), t10 = ddot3(g.E1, G0), t11 = ddot3(g.E1, G1);
    r.e00 = t00;
    r.e11 = t11;
    r.g01 = t01 + t10;
    r.om = 0.5 * (t01 - t10) + ddot3(g.E2, th);
    r.ga0 = ddot3(g.E1, th) + ddot3(g.E2, G0);
    r.ga1 = -ddot3(g.E0, th) + ddot3(g.E2, G1);
    D1 x00[3], x01[3], x10[3], x11[3];
    dcross(g.E0, s.w0, x00);
    dcross(g.E0, s.w1, x01);
    dcross(g.E1, s.w0, x10);
    dcross(g.E1, s.w1, x11);
    auto dd = [](const D1* a, const D1* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
    r.k00 = -ddot3(g.E1, T0) + dd(th, x00);
    r.k11 = ddot3(g.E0, T1) + dd(th, x11);
    r.k01 = -ddot3(g.E1, T1) + dd(th, x01) + ddot3(g.E0, T0) + dd(th, x10);
    return r;
}

// lam-strain . C . w-strain with the reference's J placement (membrane/bending: none; shear/drilling: J)
__device__ __forceinline__ D1 energy_density_dual(const GenD& a, const GenD& b, double h, double E, double nu, double hK,
                                                  double wdetS, double wdet, D1 Ju) {
    const double c = E / (1.0 - nu * nu), sh = 0.5 * (1.0 - nu);
    const double cm = c * h * wdetS, cb = c * h * h * h / 12.0 * wdetS;
    const D1 cs = (K_SHEAR * E / (2.0 * (1.0 + nu)) * h * wdetS) * Ju;
    const D1 cd = (E * h * h * h / (hK * hK) * wdet) * Ju;
    const D1 mem = cm * ((a.e00 + nu * a.e11) * b.e00 + (nu * a.e00 + a.e11) * b.e11 + sh * (a.g01 * b.g01));
    const D1 ben = cb * ((a.k00 + nu * a.k11) * b.k00 + (nu * a.k00 + a.k11) * b.k11 + sh * (a.k01 * b.k01));
    return mem + ben + cs * (a.ga0 * b.ga0 + a.ga1 * b.ga1) + cd * (a.om * b.om);
}

// mode 0: lam . (K w - F)   (the residual's elastic + load part; penalty handled per facet)
// mode 1: int u.u J dx      (compliance without the regularisation, which has no uhat dependence)
// mode 2: int rho h J dx    (mass)
// mode 3: 1/2 w . K w       (elastic energy)
// mode 4: int (m vm_top)^rho J dx   (p-norm stress before the 1/alpha scaling)
template <int NPC, int NVC, bool QUAD>
__global__ void __launch_bounds__(128)
k_shape_gradient(MeshDev m, FieldsDev f, const Tables* __restrict__ tab, int mode, const double* __restrict__ w,
                 const double* __restrict__ lam, double scale, double ms, double rho, double regc, double* __restrict__ out) {
    constexpr int LD = 3 * NPC + 3 * NVC;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = gid / (3 * NVC), dir = gid - e * (3 * NVC);
    if (e >= m.nel) return;
    if (mode == 4 && !cell_selected(m, e)) return;
    const int bseed = dir / 3, iseed = dir - 3 * bseed;
    Elem<NPC, NVC> el;
    load_elem<NPC, NVC, true>(m, f, e, el);
    D1 Uh[NVC][3];
    for (int b = 0; b < NVC; ++b)
        for (int i = 0; i < 3; ++i) Uh[b][i] = mk(el.Uh[b][i], (b == bseed && i == iseed) ? 1.0 : 0.0);
    double we[LD], le[LD];
    for (int a = 0; a < NPC; ++a)
        for (int c = 0; c < 3; ++c) {
            we[3 * a + c] = w[3 * el.pid[a] + c];
            le[3 * a + c] = lam ? lam[3 * el.pid[a] + c] : 0.0;
        }
    for (int b = 0; b < NVC; ++b)
        for (int c = 0; c < 3; ++c) {
            we[3 * NPC + 3 * b + c] = w[m.ndof_u + 3 * rot_node(m, el, b) + c];
            le[3 * NPC + 3 * b + c] = lam ? lam[m.ndof_u + 3 * rot_node(m, el, b) + c] : 0.0;
        }
    double fn[NVC][3], rhon[NVC];
    for (int b = 0; b < NVC; ++b) {
        for (int c = 0; c < 3; ++c) fn[b][c] = f.f[3 * (f.ewp ? e : el.vid[b]) + c];
        rhon[b] = f.rho[f.ewm ? e : el.vid[b]];
    }
    D1 phi = mk(0.0);
    const int nq = tab->nq;
    for (int q = 0; q < nq; ++q) {
        QPG g;
        double zero[NVC][3] = {};
        qp_geometry<NVC, QUAD, false>(el.X, zero, tab->N1[q], tab->dN1[q], g);
        QPD s;
        qp_shape_dual<NVC, QUAD>(el.X, Uh, tab->dN1[q], g, s);
        const double wdet = tab->w[q] * g.det, wdetS = tab->wS[q] * g.det;
        const double hq = interp<NVC>(tab->N1[q], el.hn);
        if (mode == 0 || mode == 3) {
            const double Eq = interp<NVC>(tab->N1[q], el.En), nuq = interp<NVC>(tab->N1[q], el.nun);
            const GenD sw = strains_dual<NPC, NVC>(*tab, q, g, s, we);
            if (mode == 3) {
                phi = phi + 0.5 * energy_density_dual(sw, sw, hq, Eq, nuq, el.hK, wdetS, wdet, s.Ju);
            } else {
                const GenD sl = strains_dual<NPC, NVC>(*tab, q, g, s, le);
                phi = phi + energy_density_dual(sw, sl, hq, Eq, nuq, el.hK, wdetS, wdet, s.Ju);
                double fq[3] = {0, 0, 0}, lq[3] = {0, 0, 0};
                for (int b = 0; b < NVC; ++b)
                    for (int c = 0; c < 3; ++c) fq[c] += tab->N1[q][b] * fn[b][c];
                for (int a = 0; a < NPC; ++a)
                    for (int c = 0; c < 3; ++c) lq[c] += tab->N2[q][a] * le[3 * a + c];
                phi = phi - (wdet * dot3(fq, lq)) * s.Ju;
            }
        } else if (mode == 4) {
            const double Eq = interp<NVC>(tab->N1[q], el.En), nuq = interp<NVC>(tab->N1[q], el.nun);
            const GenD sw = strains_dual<NPC, NVC>(*tab, q, g, s, we);
            double th[3] = {0, 0, 0};
            D1 gh0 = mk(0.0), gh1 = mk(0.0);
            for (int b = 0; b < NVC; ++b) {
                for (int c = 0; c < 3; ++c) th[c] += tab->NR[q][b] * we[3 * NPC + 3 * b + c];
                if (!f.ewm) {
                    const double r0 = tab->dN1[q][b][0], r1 = tab->dN1[q][b][1];
                    gh0 = gh0 + el.hn[b] * (r0 * s.Q[0][0] + r1 * s.Q[1][0]);
                    gh1 = gh1 + el.hn[b] * (r0 * s.Q[0][1] + r1 * s.Q[1][1]);
                }
            }
            const double b0 = -dot3(th, g.E1), b1 = dot3(th, g.E0), z = 0.5 * hq;
            const D1 e0 = sw.e00 - z * sw.k00 - (0.5 * b0) * gh0;
            const D1 e1 = sw.e11 - z * sw.k11 - (0.5 * b1) * gh1;
            const D1 gg = sw.g01 - z * sw.k01 - 0.5 * (b0 * gh1 + b1 * gh0);
            const double cc = Eq / (1.0 - nuq * nuq);
            const D1 s0 = cc * (e0 + nuq * e1), s1 = cc * (nuq * e0 + e1), s2 = (cc * 0.5 * (1.0 - nuq)) * gg;
            const D1 vm = dsqrt(s0 * s0 - s0 * s1 + s1 * s1 + 3.0 * (s2 * s2));
            if (vm.v > 0.0) {
                const double p = pow(ms * vm.v, rho);
                const D1 pw = mk(p, rho * p / vm.v * vm.d);
                phi = phi + wdet * (pw * s.Ju);
            }
            if (regc != 0.0) phi = phi + (wdet * regc * pow(hq, rho)) * s.Ju;
        } else if (mode == 1) {
            double uq[3] = {0, 0, 0};
            for (int a = 0; a < NPC; ++a)
                for (int c = 0; c < 3; ++c) uq[c] += tab->N2[q][a] * we[3 * a + c];
            phi = phi + (wdet * dot3(uq, uq)) * s.Ju;
        } else {
            phi = phi + (wdet * hq * interp<NVC>(tab->N1[q], rhon)) * s.Ju;
        }
    }
    atomicAdd(&out[3 * el.vid[bseed] + iseed], scale * phi.d);
}

// penalty term: lam . P(uhat) w  with  P = beta/h_K int |J F^-T N| (.,.) ds  per tagged facet
// CG1: the CG1CG1 element -- the displacement's edge block is the linear one (k_penalty_setup's U3)
template <int NVC, bool QUAD, bool CG1>
__global__ void k_shape_gradient_penalty(MeshDev m, FieldsDev f, FacetDev fd, double beta, const double* __restrict__ w,
                                         const double* __restrict__ lam, double scale, double* __restrict__ out) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = gid / (3 * NVC), dir = gid - i * (3 * NVC);
    if (i >= fd.nf) return;
    const int bseed = dir / 3, iseed = dir - 3 * bseed;
    const int e = fd.cell[i], k = fd.ledge[i];
    double X[NVC][3];
    D1 Uh[NVC][3];
    int vid[NVC];
    for (int b = 0; b < NVC; ++b) {
        const int v = m.cells[b * m.nel + e];
        vid[b] = v;
        for (int c = 0; c < 3; ++c) {
            X[b][c] = m.xyz[3 * v + c];
            Uh[b][c] = mk(f.uhat[3 * v + c], (b == bseed && c == iseed) ? 1.0 : 0.0);
        }
    }
    const int ka = k, kb = (k + 1) % NVC;
    double tv[3], len = 0.0;
    for (int c = 0; c < 3; ++c) {
        tv[c] = X[kb][c] - X[ka][c];
        len += tv[c] * tv[c];
    }
    len = sqrt(len);
    for (int c = 0; c < 3; ++c) tv[c] /= len;
    const int un[3] = {fd.unode[3 * i], fd.unode[3 * i + 1], fd.unode[3 * i + 2]};
    const int vn[2] = {fd.vnode[2 * i], fd.vnode[2 * i + 1]};
    const double gs[3] = {-0.7745966692414834, 0.0, 0.7745966692414834};
    const double gw[3] = {0.5555555555555556, 0.8888888888888888, 0.5555555555555556};
    D1 phi = mk(0.0);
    for (int q = 0; q < 3; ++q) {
        const double s = gs[q];
        double xi, eta, M[NVC], dM[NVC][2], zero[NVC][3] = {};
        edge_ref_point(QUAD, k, s, xi, eta);
        p1_shape<NVC, QUAD>(xi, eta, M, dM);
        QPG g;
        qp_geometry<NVC, QUAD, false>(X, zero, M, dM, g);
        QPD sd;
        qp_shape_dual<NVC, QUAD>(X, Uh, dM, g, sd);
        double Nf[3];
        cross3(tv, g.E2, Nf);
        D1 v[3];
        for (int a = 0; a < 3; ++a) v[a] = sd.cof[a][0] * Nf[0] + sd.cof[a][1] * Nf[1] + sd.cof[a][2] * Nf[2];
        const D1 nanson = dsqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        const double L1[2] = {0.5 * (1.0 - s), 0.5 * (1.0 + s)};
        const double L2[3] = {CG1 ? L1[0] : 0.5 * s * (s - 1.0), CG1 ? 0.0 : 1.0 - s * s, CG1 ? L1[1] : 0.5 * s * (s + 1.0)};
        double wl = 0.0;
        for (int c = 0; c < 3; ++c) {
            double wu = 0, lu = 0, wt = 0, lt = 0;
            for (int a = 0; a < 3; ++a) {
                wu += L2[a] * w[3 * un[a] + c];
                lu += L2[a] * lam[3 * un[a] + c];
            }
            if (fd.MR) {
                // CG2CR1: the rotation's trace on edge k through all three Crouzeix-Raviart functions of the cell (k_penalty_setup)
                double R[3];
                R[k % 3] = 1.0; R[(k + 1) % 3] = s; R[(k + 2) % 3] = -s;
                for (int a = 0; a < 3; ++a) {
                    wt += R[a] * w[m.ndof_u + 3 * fd.rnode[3 * i + a] + c];
                    lt += R[a] * lam[m.ndof_u + 3 * fd.rnode[3 * i + a] + c];
                }
            } else
            for (int a = 0; a < 2; ++a) {
                wt += L1[a] * w[m.ndof_u + 3 * vn[a] + c];
                lt += L1[a] * lam[m.ndof_u + 3 * vn[a] + c];
            }
            wl += wu * lu + wt * lt;
        }
        phi = phi + (gw[q] * 0.5 * len * beta / m.hK[e] * wl) * nanson;
    }
    atomicAdd(&out[3 * vid[bseed] + iseed], scale * phi.d);
}

}  // namespace femo
