// Device-side element mathematics of the CG2 x CG1 Reissner-Mindlin shell (gfx950, fp64).
//
// What is computed follows the reference's UFL definitions (paths relative to
// /root/reference/femo_alpha/rm_shell/linear_shell_fenicsx):
//   local basis E0,E1,E2 ............ kinematics.py:54-91
//   F = I + grad(uhat), gradx, J .... kinematics.py:12-44
//   CLT matrices A, D, A_s .......... linear_shell_model.py:136-157
//   eps, kappa, gamma, drilling ..... linear_shell_model.py:232-258, 284-296
//   energies / residual ............. linear_shell_model.py:275-321
// How it is computed is this build's own: no B matrix and no element matrix is formed for the
// operator; each quadrature point maps reference derivatives to local "gradx" derivatives with one
// 2x2 matrix Q = Jloc^-1 (T F^-1 T^T), reduces the nodal values to six 3-vectors
// (G0,G1 | theta,Th0,Th1), evaluates nine generalised strains and scatters the nine stresses back
// through the same six vectors.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace femo {

constexpr int MAXQ = 36;     // 6 x 6 Gauss points: strongly non-affine quadrilaterals (ShellMesh.recommended_nquad)
constexpr double K_SHEAR = 0.833;       // linear_shell_model.py:146
constexpr double REG_ALPHA1 = 1.0e-2;   // rm_shell_pde.py:67

struct Tables {
    int nq;
    int pad;
    double w[MAXQ];       // weight of the full rule: drilling, inertia, load, outputs (0 at reduced-rule-only points)
    double wS[MAXQ];      // weight for the membrane / bending / shear energies (the reference's dx_inplane, dx_shear;
                          // equal to w unless a reduced degree is requested, dynamic_rm_shell/plate_sim.py:82-91)
    double N2[MAXQ][9];
    double dN2[MAXQ][9][2];
    double N1[MAXQ][4];
    double dN1[MAXQ][4][2];
    // shape functions of the ROTATION: a copy of N1 / dN1 for CG2CG1 and CG1CG1 (rotation on the vertices); for CG2CR1 (triangles,
    // linear_shell_model.py:68-73) the Crouzeix-Raviart functions of the edge midpoints, NR_k = 1 - 2 lambda_(k+2) for edge k (vertex
    // k -> k + 1).  N1 / dN1 stay what the geometry and the nodal fields (thickness, E, nu, density, pressure) are interpolated with
    double NR[MAXQ][4];
    double dNR[MAXQ][4][2];
};

struct MeshDev {
    int nn, nel, nP2, ndof_u, ndof;
    const double* xyz;    // nn*3
    const int* cells;     // SoA [nvc][nel]
    const int* cellp2;    // SoA [npc][nel]
    const double* hK;     // nel, UFL CellDiameter
    const int* ctag;      // nel, sub-domain index of every cell (-1: none), or null
    int csel;             // sub-domain the stress aggregate integrates over; -1: the whole mesh
    int cr;               // CG2CR1: the rotation lives on the edge midpoints -- rotation node k of a cell = its P2 node 3 + k, numbered from nn
};

// false for cells outside the selected sub-domain (the reference's dxx(i) measure, rm_shell_model.py:242-253)
__device__ __forceinline__ bool cell_selected(const MeshDev& m, int e) { return m.csel < 0 || (m.ctag && m.ctag[e] == m.csel); }

struct FieldsDev {
    const double* h;
    const double* E;
    const double* nu;
    const double* rho;
    const double* f;
    const double* uhat;
    int ewm, ewp;
};

// ------------------------------------------------------------------------------------------ helpers
__device__ __forceinline__ double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ void cross3(const double* a, const double* b, double* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

// sum over the 64 lanes of a wave, returned in every lane: four DPP steps inside each row of 16 lanes
// (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror), then the four row totals through
// v_readlane -- no LDS traffic (a __shfl_down ladder costs six dependent ds_bpermute round trips)
template <int CTRL>
__device__ __forceinline__ double dpp_add(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
    return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_bcast(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wave_sum(double v) {
    v = dpp_add<0xB1>(v);
    v = dpp_add<0x4E>(v);
    v = dpp_add<0x141>(v);
    v = dpp_add<0x140>(v);
    return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
}

// Column sums of N per-lane values (N = 4, 8, 16 or 32) over the 64 lanes in one butterfly: at every step a lane keeps half of its
// values and hands the other half to a partner in the other half of the wave / row pair / row / ..., so N - 1 exchanges
// leave ONE column per group of 64 / N neighbouring lanes, which then add up among themselves.  All on the vector ALU: the two
// widest steps are gfx950's v_permlane32_swap / v_permlane16_swap (the swap puts both halves of a column's sum into the
// same lane: no select), the steps inside a row of 16 lanes are DPP moves.  Returns the sum of column `col` (the same in all
// lanes of the group; p is clobbered).  N separate wave_sum calls cost 25 N instructions -- in the transposed products of the
// triangular sweeps they, not the loads, set the pace (k_sweep_gemv_t with X^T: 25-36 -> 13-18 us per level at 1M DOF).
template <int CTRL>
__device__ __forceinline__ double dpp_get(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// a: lanes 32-63 <-> b: lanes 0-31.  Afterwards a = [a.lower | b.lower], b = [a.upper | b.upper]
__device__ __forceinline__ void permlane32_swap(double& a, double& b) {
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi[0], (int)lo[0]); b = __hiloint2double((int)hi[1], (int)lo[1]);
}
// odd rows (of 16 lanes) of a <-> even rows of b.  Afterwards a = [a.r0, b.r0, a.r2, b.r2], b = [a.r1, b.r1, a.r3, b.r3]
__device__ __forceinline__ void permlane16_swap(double& a, double& b) {
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi[0], (int)lo[0]); b = __hiloint2double((int)hi[1], (int)lo[1]);
}
// one butterfly step inside a row: the partner (DPP pattern CTRL, an involution that flips the lane bit `up` tests) gets the W
// values this lane does not keep
template <int CTRL, int W, int N>
__device__ __forceinline__ void dpp_halve(double (&p)[N], bool up) {
#pragma unroll
    for (int i = 0; i < W; ++i) {
        const double keep = up ? p[i + W] : p[i], send = up ? p[i] : p[i + W];
        p[i] = keep + dpp_get<CTRL>(send);
    }
}
template <int N>
__device__ __forceinline__ double wave_sum_cols(double (&p)[N], int lane, int& col) {
    static_assert(N == 4 || N == 8 || N == 16 || N == 32, "N: 4, 8, 16 or 32");
#pragma unroll
    for (int i = 0; i < N / 2; ++i) { permlane32_swap(p[i], p[i + N / 2]); p[i] += p[i + N / 2]; }
    col = (lane & 32) ? N / 2 : 0;
#pragma unroll
    for (int i = 0; i < N / 4; ++i) { permlane16_swap(p[i], p[i + N / 4]); p[i] += p[i + N / 4]; }
    col += (lane & 16) ? N / 4 : 0;
    if constexpr (N >= 8)  { dpp_halve<0x140, N / 8>(p, (lane & 8) != 0);  col += (lane & 8) ? N / 8 : 0; }     // row_mirror
    if constexpr (N >= 16) { dpp_halve<0x141, N / 16>(p, (lane & 4) != 0); col += (lane & 4) ? N / 16 : 0; }    // row_half_mirror
    if constexpr (N >= 32) { dpp_halve<0x4E, 1>(p, (lane & 2) != 0);       col += (lane & 2) ? 1 : 0; }         // quad_perm [2,3,0,1]
    double v = dpp_add<0xB1>(p[0]);
    if constexpr (N <= 16) v = dpp_add<0x4E>(v);
    if constexpr (N <= 8) v = dpp_add<0x141>(v);
    if constexpr (N <= 4) v = dpp_add<0x140>(v);
    return v;
}

// one atomic per block into *slot (slot may be null)
__device__ __forceinline__ void block_accumulate(double v, double* slot) {
    __shared__ double s_part[16];
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) s_part[wid] = v;
    __syncthreads();
    if (threadIdx.x == 0 && slot) {
        double t = 0.0;
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; ++i) t += s_part[i];
        atomicAdd(slot, t);
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------ element
template <int NPC, int NVC>
struct Elem {
    int vid[NVC];
    int pid[NPC];
    double X[NVC][3];
    double Uh[NVC][3];
    double hn[NVC], En[NVC], nun[NVC];
    double hK;
};

template <int NPC, int NVC, bool UHAT>
__device__ __forceinline__ void load_elem(const MeshDev& m, const FieldsDev& f, int e, Elem<NPC, NVC>& el) {
#pragma unroll
    for (int b = 0; b < NVC; ++b) {
        const int v = m.cells[b * m.nel + e];
        el.vid[b] = v;
#pragma unroll
        for (int c = 0; c < 3; ++c) el.X[b][c] = m.xyz[3 * v + c];
        if (UHAT) {
#pragma unroll
            for (int c = 0; c < 3; ++c) el.Uh[b][c] = f.uhat[3 * v + c];
        }
        const int t = f.ewm ? e : v;
        el.hn[b] = f.h[t];
        el.En[b] = f.E[t];
        el.nun[b] = f.nu[t];
    }
#pragma unroll
    for (int a = 0; a < NPC; ++a) el.pid[a] = m.cellp2[a * m.nel + e];
    el.hK = m.hK[e];
}

// node that carries the rotation DOFs of local slot b (theta entries of the state vector: ndof_u + 3 * node + c): the vertex for
// CG2CG1 / CG1CG1; for CG2CR1 the midpoint of edge b, which is also P2 node NVC + b of the cell (mesh.py: cell_p2 = vertices, edges)
template <int NPC, int NVC>
__device__ __forceinline__ int rot_node(const MeshDev& m, const Elem<NPC, NVC>& el, int b) {
    if constexpr (NPC == 6 && NVC == 3) {
        if (m.cr) return el.pid[NVC + b] - m.nn;
    }
    return el.vid[b];
}

// geometry of one quadrature point
struct QPG {
    double E0[3], E1[3], E2[3];
    double Q[2][2];    // reference derivative -> local gradx derivative: d_J = dN_0 Q[0][J] + dN_1 Q[1][J]
    double Q0[2][2];   // the same without mesh motion (plain surface gradient)
    double w0[3], w1[3];  // derivative of the unit normal along local gradx directions 0, 1
    double det;        // |J0 x J1|
    double Ju;         // det F(uhat)
};

template <int NVC, bool QUAD, bool UHAT>
__device__ __forceinline__ void qp_geometry(const double (*X)[3], const double (*Uh)[3], const double* M,
                                            const double (*dM)[2], QPG& g) {
    double J0[3] = {0, 0, 0}, J1[3] = {0, 0, 0};
#pragma unroll
    for (int b = 0; b < NVC; ++b)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            J0[c] += X[b][c] * dM[b][0];
            J1[c] += X[b][c] * dM[b][1];
        }
    double a[3];
    cross3(J0, J1, a);
    g.det = sqrt(dot3(a, a));
    const double idet = 1.0 / g.det;
    const double l0 = sqrt(dot3(J0, J0));
    const double il0 = 1.0 / l0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        g.E2[c] = a[c] * idet;
        g.E0[c] = J0[c] * il0;
    }
    cross3(g.E2, g.E0, g.E1);
    const double j01 = dot3(g.E0, J1), j11 = dot3(g.E1, J1);
    // Jloc = [[l0, j01], [0, j11]]  ->  Jloc^-1
    g.Q0[0][0] = il0;
    g.Q0[0][1] = -j01 * il0 / j11;
    g.Q0[1][0] = 0.0;
    g.Q0[1][1] = 1.0 / j11;
    g.Ju = 1.0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) g.Q[i][j] = g.Q0[i][j];
    if (UHAT) {
        // F = I + sum_b Uh_b (x) gradM_b, gradM_b = (dM_b^T Q0)_0 E0 + (dM_b^T Q0)_1 E1
        double F[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
#pragma unroll
        for (int b = 0; b < NVC; ++b) {
            const double d0 = dM[b][0] * g.Q0[0][0] + dM[b][1] * g.Q0[1][0];
            const double d1 = dM[b][0] * g.Q0[0][1] + dM[b][1] * g.Q0[1][1];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double gm = d0 * g.E0[j] + d1 * g.E1[j];
#pragma unroll
                for (int i = 0; i < 3; ++i) F[i][j] += Uh[b][i] * gm;
            }
        }
        double C[3][3];   // cofactors
        C[0][0] = F[1][1] * F[2][2] - F[1][2] * F[2][1];
        C[0][1] = F[1][2] * F[2][0] - F[1][0] * F[2][2];
        C[0][2] = F[1][0] * F[2][1] - F[1][1] * F[2][0];
        C[1][0] = F[0][2] * F[2][1] - F[0][1] * F[2][2];
        C[1][1] = F[0][0] * F[2][2] - F[0][2] * F[2][0];
        C[1][2] = F[0][1] * F[2][0] - F[0][0] * F[2][1];
        C[2][0] = F[0][1] * F[1][2] - F[0][2] * F[1][1];
        C[2][1] = F[0][2] * F[1][0] - F[0][0] * F[1][2];
        C[2][2] = F[0][0] * F[1][1] - F[0][1] * F[1][0];
        g.Ju = F[0][0] * C[0][0] + F[0][1] * C[0][1] + F[0][2] * C[0][2];
        const double iJ = 1.0 / g.Ju;
        // Finv[k][j] = C[j][k] / Ju ;  S[b][a] = E_b . Finv E_a
        double FE0[3], FE1[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            FE0[k] = (C[0][k] * g.E0[0] + C[1][k] * g.E0[1] + C[2][k] * g.E0[2]) * iJ;
            FE1[k] = (C[0][k] * g.E1[0] + C[1][k] * g.E1[1] + C[2][k] * g.E1[2]) * iJ;
        }
        const double S00 = dot3(g.E0, FE0), S01 = dot3(g.E0, FE1);
        const double S10 = dot3(g.E1, FE0), S11 = dot3(g.E1, FE1);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            g.Q[i][0] = g.Q0[i][0] * S00 + g.Q0[i][1] * S10;
            g.Q[i][1] = g.Q0[i][0] * S01 + g.Q0[i][1] * S11;
        }
    }
    if (QUAD) {
        // d2x/dxi deta of the bilinear map, then d(normal)/dxi_m
        double tw[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) tw[c] = 0.25 * (X[0][c] - X[1][c] + X[2][c] - X[3][c]);
        double da0[3], da1[3];
        cross3(J0, tw, da0);
        cross3(tw, J1, da1);
        const double p0 = dot3(g.E2, da0), p1 = dot3(g.E2, da1);
        double dn0[3], dn1[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            dn0[c] = (da0[c] - g.E2[c] * p0) * idet;
            dn1[c] = (da1[c] - g.E2[c] * p1) * idet;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            g.w0[c] = dn0[c] * g.Q[0][0] + dn1[c] * g.Q[1][0];
            g.w1[c] = dn0[c] * g.Q[0][1] + dn1[c] * g.Q[1][1];
        }
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) g.w0[c] = g.w1[c] = 0.0;
    }
}

// nine generalised strains / the nine conjugate stresses
struct Gen {
    double e00, e11, g01, k00, k11, k01, ga0, ga1, om;
};

struct Mat {
    double cm, cb, cs, cd, nu;   // membrane, bending, shear, drilling coefficients (measure included)
};

enum { DERIV_NONE = 0, DERIV_H = 1, DERIV_E = 2, DERIV_NU = 3 };

// constitutive coefficients at a point; which = derivative selector
template <int WHICH>
__device__ __forceinline__ void material(double h, double E, double nu, double hK, double wdetS, double wdet, double Ju, Mat& m,
                                         Mat& dm_dnu_extra) {
    const double om = 1.0 - nu * nu;
    const double c = E / om;
    const double G2 = 1.0 / (2.0 * (1.0 + nu));
    const double ihk2 = 1.0 / (hK * hK);
    m.nu = nu;
    if (WHICH == DERIV_NONE) {
        m.cm = c * h * wdetS;
        m.cb = c * h * h * h / 12.0 * wdetS;
        m.cs = K_SHEAR * E * G2 * h * Ju * wdetS;
        m.cd = E * h * h * h * ihk2 * Ju * wdet;
    } else if (WHICH == DERIV_H) {
        m.cm = c * wdetS;
        m.cb = c * h * h / 4.0 * wdetS;
        m.cs = K_SHEAR * E * G2 * Ju * wdetS;
        m.cd = 3.0 * E * h * h * ihk2 * Ju * wdet;
    } else if (WHICH == DERIV_E) {
        m.cm = h / om * wdetS;
        m.cb = h * h * h / 12.0 / om * wdetS;
        m.cs = K_SHEAR * G2 * h * Ju * wdetS;
        m.cd = h * h * h * ihk2 * Ju * wdet;
    } else {   // d/dnu: C' = c' P + c P',  c' = 2 nu E / (1-nu^2)^2
        const double dc = 2.0 * nu * E / (om * om);
        m.cm = dc * h * wdetS;
        m.cb = dc * h * h * h / 12.0 * wdetS;
        m.cs = -K_SHEAR * E * h * 2.0 * G2 * G2 * Ju * wdetS;
        m.cd = 0.0;
        dm_dnu_extra.cm = c * h * wdetS;                  // multiplies P' = [[0,1,0],[1,0,0],[0,0,-1/2]]
        dm_dnu_extra.cb = c * h * h * h / 12.0 * wdetS;
    }
}

__device__ __forceinline__ Gen stress_of(const Gen& s, const Mat& m) {
    Gen t;
    const double sh = 0.5 * (1.0 - m.nu);
    t.e00 = m.cm * (s.e00 + m.nu * s.e11);
    t.e11 = m.cm * (m.nu * s.e00 + s.e11);
    t.g01 = m.cm * sh * s.g01;
    t.k00 = m.cb * (s.k00 + m.nu * s.k11);
    t.k11 = m.cb * (m.nu * s.k00 + s.k11);
    t.k01 = m.cb * sh * s.k01;
    t.ga0 = m.cs * s.ga0;
    t.ga1 = m.cs * s.ga1;
    t.om = m.cd * s.om;
    return t;
}

// extra term of d/dnu: c P' strain
__device__ __forceinline__ void stress_add_dnu(const Gen& s, const Mat& x, Gen& t) {
    t.e00 += x.cm * s.e11;
    t.e11 += x.cm * s.e00;
    t.g01 += -0.5 * x.cm * s.g01;
    t.k00 += x.cb * s.k11;
    t.k11 += x.cb * s.k00;
    t.k01 += -0.5 * x.cb * s.k01;
}

__device__ __forceinline__ double gen_dot(const Gen& a, const Gen& b) {
    return a.e00 * b.e00 + a.e11 * b.e11 + a.g01 * b.g01 + a.k00 * b.k00 + a.k11 * b.k11 + a.k01 * b.k01 +
           a.ga0 * b.ga0 + a.ga1 * b.ga1 + a.om * b.om;
}

template <int NPC, int NVC>
__device__ __forceinline__ void local_derivs(const Tables& t, int q, const double Q[2][2], double (*d)[2], double (*m)[2]) {
#pragma unroll
    for (int a = 0; a < NPC; ++a) {
        const double r0 = t.dN2[q][a][0], r1 = t.dN2[q][a][1];
        d[a][0] = r0 * Q[0][0] + r1 * Q[1][0];
        d[a][1] = r0 * Q[0][1] + r1 * Q[1][1];
    }
#pragma unroll
    for (int b = 0; b < NVC; ++b) {
        const double r0 = t.dNR[q][b][0], r1 = t.dNR[q][b][1];
        m[b][0] = r0 * Q[0][0] + r1 * Q[1][0];
        m[b][1] = r0 * Q[0][1] + r1 * Q[1][1];
    }
}

// generalised strains from the six reduced 3-vectors  G_k = sum_a d[a][k] u_a,  th = sum_b M_b theta_b,
// T_k = sum_b m[b][k] theta_b
__device__ __forceinline__ Gen strains_reduced(const QPG& g, const double* G0, const double* G1, const double* th,
                                               const double* T0, const double* T1);

// strains of the element vector xe = [u_a xyz ..., theta_b xyz ...]
template <int NPC, int NVC>
__device__ __forceinline__ Gen strains(const QPG& g, const double (*d)[2], const double (*m)[2], const double* M,
                                       const double* xe) {
    double G0[3] = {0, 0, 0}, G1[3] = {0, 0, 0};
#pragma unroll
    for (int a = 0; a < NPC; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            G0[c] += xe[3 * a + c] * d[a][0];
            G1[c] += xe[3 * a + c] * d[a][1];
        }
    double th[3] = {0, 0, 0}, T0[3] = {0, 0, 0}, T1[3] = {0, 0, 0};
#pragma unroll
    for (int b = 0; b < NVC; ++b)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double v = xe[3 * NPC + 3 * b + c];
            th[c] += M[b] * v;
            T0[c] += m[b][0] * v;
            T1[c] += m[b][1] * v;
        }
    return strains_reduced(g, G0, G1, th, T0, T1);
}

__device__ __forceinline__ Gen strains_reduced(const QPG& g, const double* G0, const double* G1, const double* th,
                                               const double* T0, const double* T1) {
    Gen s;
    const double t00 = dot3(g.E0, G0), t01 = dot3(g.E0, G1), t10 = dot3(g.E1, G0), t11 = dot3(g.E1, G1);
    s.e00 = t00;
    s.e11 = t11;
    s.g01 = t01 + t10;
    s.om = 0.5 * (t01 - t10) + dot3(th, g.E2);
    s.ga0 = dot3(th, g.E1) + dot3(g.E2, G0);
    s.ga1 = -dot3(th, g.E0) + dot3(g.E2, G1);
    double x00[3], x01[3], x10[3], x11[3];
    cross3(g.E0, g.w0, x00);
    cross3(g.E0, g.w1, x01);
    cross3(g.E1, g.w0, x10);
    cross3(g.E1, g.w1, x11);
    const double b00 = -dot3(g.E1, T0) + dot3(th, x00);
    const double b01 = -dot3(g.E1, T1) + dot3(th, x01);
    const double b10 = dot3(g.E0, T0) + dot3(th, x10);
    const double b11 = dot3(g.E0, T1) + dot3(th, x11);
    s.k00 = b00;
    s.k11 = b11;
    s.k01 = b01 + b10;
    return s;
}

// ye += B^T t  (transpose of `strains`)
template <int NPC, int NVC>
__device__ __forceinline__ void strains_T(const QPG& g, const double (*d)[2], const double (*m)[2], const double* M,
                                          const Gen& t, double* ye) {
    double H0[3], H1[3];
    const double a10 = t.g01 - 0.5 * t.om, a01 = t.g01 + 0.5 * t.om;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        H0[c] = t.e00 * g.E0[c] + a10 * g.E1[c] + t.ga0 * g.E2[c];
        H1[c] = a01 * g.E0[c] + t.e11 * g.E1[c] + t.ga1 * g.E2[c];
    }
#pragma unroll
    for (int a = 0; a < NPC; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) ye[3 * a + c] += d[a][0] * H0[c] + d[a][1] * H1[c];
    double x00[3], x01[3], x10[3], x11[3];
    cross3(g.E0, g.w0, x00);
    cross3(g.E0, g.w1, x01);
    cross3(g.E1, g.w0, x10);
    cross3(g.E1, g.w1, x11);
    double Tq[3], C0[3], C1[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        Tq[c] = t.ga0 * g.E1[c] - t.ga1 * g.E0[c] + t.om * g.E2[c] + t.k00 * x00[c] + t.k11 * x11[c] +
                t.k01 * (x01[c] + x10[c]);
        C0[c] = -t.k00 * g.E1[c] + t.k01 * g.E0[c];
        C1[c] = -t.k01 * g.E1[c] + t.k11 * g.E0[c];
    }
#pragma unroll
    for (int b = 0; b < NVC; ++b)
#pragma unroll
        for (int c = 0; c < 3; ++c) ye[3 * NPC + 3 * b + c] += M[b] * Tq[c] + m[b][0] * C0[c] + m[b][1] * C1[c];
}

template <int NVC>
__device__ __forceinline__ double interp(const double* M, const double* v) {
    double r = 0.0;
#pragma unroll
    for (int b = 0; b < NVC; ++b) r += M[b] * v[b];
    return r;
}

// What every lane of a wave that forms ONE element's matrix needs at a quadrature point -- local frame, derivative map,
// the local derivatives of all shape functions, the constitutive coefficients -- is the same for the 39 lanes: there is no
// scalar fp64 unit to compute it once, so the lanes used to compute it 39 times over (two square roots and seven divisions
// among it: about half of the kernel's instructions).  Here lane q computes point q ONCE, all points side by side, and parks
// it in LDS; the column loop reads it back as broadcast operands.
template <int NPC, int NVC>
struct QPoint {
    QPG g;
    Mat mat;
    double d[NPC][2], mm[NVC][2];
    double hq;
};
template <int NPC, int NVC, bool QUAD, bool UHAT>
__device__ __forceinline__ void stage_qpoints(const Tables* __restrict__ tab, const Elem<NPC, NVC>& el, double aK, int lane, int nlanes,
                                              QPoint<NPC, NVC>* sq) {
    const int nq = tab->nq;
    for (int q = lane; q < nq; q += nlanes) {
        QPoint<NPC, NVC> p;
        qp_geometry<NVC, QUAD, UHAT>(el.X, el.Uh, tab->N1[q], tab->dN1[q], p.g);
        local_derivs<NPC, NVC>(*tab, q, p.g.Q, p.d, p.mm);
        Mat ex;
        p.hq = interp<NVC>(tab->N1[q], el.hn);
        material<DERIV_NONE>(p.hq, interp<NVC>(tab->N1[q], el.En), interp<NVC>(tab->N1[q], el.nun), el.hK, tab->wS[q] * p.g.det,
                             tab->w[q] * p.g.det, p.g.Ju, p.mat, ex);
        p.mat.cm *= aK; p.mat.cb *= aK; p.mat.cs *= aK; p.mat.cd *= aK;
        sq[q] = p;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------ kernels
// register-lean variants for the production operator: local derivatives are recomputed from the
// (scalar-cached) tables where they are used instead of being kept in 26 register pairs
template <int NPC, int NVC>
__device__ __forceinline__ Gen strains_q(const Tables& t, int q, const QPG& g, const double* xe) {
    double G0[3] = {0, 0, 0}, G1[3] = {0, 0, 0};
#pragma unroll
    for (int a = 0; a < NPC; ++a) {
        const double r0 = t.dN2[q][a][0], r1 = t.dN2[q][a][1];
        const double d0 = r0 * g.Q[0][0] + r1 * g.Q[1][0], d1 = r0 * g.Q[0][1] + r1 * g.Q[1][1];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            G0[c] += xe[3 * a + c] * d0;
            G1[c] += xe[3 * a + c] * d1;
        }
    }
    double th[3] = {0, 0, 0}, T0[3] = {0, 0, 0}, T1[3] = {0, 0, 0};
#pragma unroll
    for (int b = 0; b < NVC; ++b) {
        const double r0 = t.dNR[q][b][0], r1 = t.dNR[q][b][1];
        const double m0 = r0 * g.Q[0][0] + r1 * g.Q[1][0], m1 = r0 * g.Q[0][1] + r1 * g.Q[1][1];
        const double Mb = t.NR[q][b];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double v = xe[3 * NPC + 3 * b + c];
            th[c] += Mb * v;
            T0[c] += m0 * v;
            T1[c] += m1 * v;
        }
    }
    Gen s;
    const double t00 = dot3(g.E0, G0), t01 = dot3(g.E0, G1), t10 = dot3(g.E1, G0), t11 = dot3(g.E1, G1);
    s.e00 = t00;
    s.e11 = t11;
    s.g01 = t01 + t10;
    s.om = 0.5 * (t01 - t10) + dot3(th, g.E2);
    s.ga0 = dot3(th, g.E1) + dot3(g.E2, G0);
    s.ga1 = -dot3(th, g.E0) + dot3(g.E2, G1);
    double x00[3], x01[3], x10[3], x11[3];
    cross3(g.E0, g.w0, x00);
    cross3(g.E0, g.w1, x01);
    cross3(g.E1, g.w0, x10);
    cross3(g.E1, g.w1, x11);
    s.k00 = -dot3(g.E1, T0) + dot3(th, x00);
    s.k11 = dot3(g.E0, T1) + dot3(th, x11);
    s.k01 = -dot3(g.E1, T1) + dot3(th, x01) + dot3(g.E0, T0) + dot3(th, x10);
    return s;
}

template <int NPC, int NVC>
__device__ __forceinline__ void strains_T_q(const Tables& t, int q, const QPG& g, const Gen& tt, double* ye) {
    double H0[3], H1[3];
    const double a10 = tt.g01 - 0.5 * tt.om, a01 = tt.g01 + 0.5 * tt.om;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        H0[c] = tt.e00 * g.E0[c] + a10 * g.E1[c] + tt.ga0 * g.E2[c];
        H1[c] = a01 * g.E0[c] + tt.e11 * g.E1[c] + tt.ga1 * g.E2[c];
    }
#pragma unroll
    for (int a = 0; a < NPC; ++a) {
        const double r0 = t.dN2[q][a][0], r1 = t.dN2[q][a][1];
        const double d0 = r0 * g.Q[0][0] + r1 * g.Q[1][0], d1 = r0 * g.Q[0][1] + r1 * g.Q[1][1];
#pragma unroll
        for (int c = 0; c < 3; ++c) ye[3 * a + c] += d0 * H0[c] + d1 * H1[c];
    }
    double x00[3], x01[3], x10[3], x11[3];
    cross3(g.E0, g.w0, x00);
    cross3(g.E0, g.w1, x01);
    cross3(g.E1, g.w0, x10);
    cross3(g.E1, g.w1, x11);
    double Tq[3], C0[3], C1[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        Tq[c] = tt.ga0 * g.E1[c] - tt.ga1 * g.E0[c] + tt.om * g.E2[c] + tt.k00 * x00[c] + tt.k11 * x11[c] +
                tt.k01 * (x01[c] + x10[c]);
        C0[c] = -tt.k00 * g.E1[c] + tt.k01 * g.E0[c];
        C1[c] = -tt.k01 * g.E1[c] + tt.k11 * g.E0[c];
    }
#pragma unroll
    for (int b = 0; b < NVC; ++b) {
        const double r0 = t.dNR[q][b][0], r1 = t.dNR[q][b][1];
        const double m0 = r0 * g.Q[0][0] + r1 * g.Q[1][0], m1 = r0 * g.Q[0][1] + r1 * g.Q[1][1];
        const double Mb = t.NR[q][b];
#pragma unroll
        for (int c = 0; c < 3; ++c) ye[3 * NPC + 3 * b + c] += Mb * Tq[c] + m0 * C0[c] + m1 * C1[c];
    }
}

// inertia of the dynamic shell, y += cm * M x at one quadrature point:  rho h (u.v + h_K^2 theta.eta) J
// (reference linear_shell_model.py:335-348 -- the vendored sibling of the un-vendored DynamicElasticModel,
// dynamic_rm_shell/plate_sim.py:197-201); cm already holds aM * rho * h * w * det * Ju
template <int NPC, int NVC>
__device__ __forceinline__ void mass_qp(const Tables& t, int q, double cm, double hK, const double* xe, double* ye) {
    double uq[3] = {0, 0, 0}, tq[3] = {0, 0, 0};
#pragma unroll
    for (int a = 0; a < NPC; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) uq[c] += t.N2[q][a] * xe[3 * a + c];
#pragma unroll
    for (int b = 0; b < NVC; ++b)
#pragma unroll
        for (int c = 0; c < 3; ++c) tq[c] += t.NR[q][b] * xe[3 * NPC + 3 * b + c];
#pragma unroll
    for (int a = 0; a < NPC; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) ye[3 * a + c] += cm * t.N2[q][a] * uq[c];
    const double ct = cm * hK * hK;
#pragma unroll
    for (int b = 0; b < NVC; ++b)
#pragma unroll
        for (int c = 0; c < 3; ++c) ye[3 * NPC + 3 * b + c] += ct * t.NR[q][b] * tq[c];
}

// ---- the production element operator: 4 lanes per element (each lane a quarter of the quadrature points),
// nodal values staged in LDS, partial results combined inside the quad with DPP, XCD-aware block order.
__device__ __forceinline__ double quad_xor_sum(double v) {
    // v + lanes xor 1 + xor 2 + xor 3 within each group of 4 lanes (DPP quad_perm, no LDS)
    int lo = __double2loint(v), hi = __double2hiint(v);
    int lo1 = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    int hi1 = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
    v += __hiloint2double(hi1, lo1);
    lo = __double2loint(v); hi = __double2hiint(v);
    lo1 = __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xF, 0xF, true);       // quad_perm [2,3,0,1]
    hi1 = __builtin_amdgcn_mov_dpp(hi, 0x4E, 0xF, 0xF, true);
    return v + __hiloint2double(hi1, lo1);
}

constexpr int YSTRIDE = 40;     // doubles per element slot of the element-result buffer (39 or 27 used)

// second pass of the element operator: y(node) = sum of the contributions of the elements around the node,
// read through the inverted connectivity (n2e_off / n2e_ent = slot * NPC + local node); also clears ghosts
template <int NPC, int NVC>
__global__ void __launch_bounds__(256)
k_gather_sum(int nP2, int nV, int ndof_u, int ndof, const int* __restrict__ n2e_off, const int* __restrict__ n2e_ent,
             const double* __restrict__ ybuf, double* __restrict__ y, int cr, int nrot) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < nP2) {
        double s0 = 0, s1 = 0, s2 = 0, t0 = 0, t1 = 0, t2 = 0;
        const int b = n2e_off[p], e = n2e_off[p + 1];
        for (int k = b; k < e; ++k) {
            const int ent = n2e_ent[k];
            const int slot = ent / NPC, a = ent - slot * NPC;
            const double* src = ybuf + (size_t)slot * YSTRIDE;
            s0 += src[3 * a]; s1 += src[3 * a + 1]; s2 += src[3 * a + 2];
            // the rotation results of the slot sit behind its displacement results: with the vertices for CG2CG1 / CG1CG1 (local
            // P2 node a < NVC), with the edge midpoints for CG2CR1 (local P2 node NVC + k carries rotation slot k)
            const int rb = cr ? a - NVC : a;
            if (rb >= 0 && rb < NVC) {
                t0 += src[3 * NPC + 3 * rb]; t1 += src[3 * NPC + 3 * rb + 1]; t2 += src[3 * NPC + 3 * rb + 2];
            }
        }
        y[3 * p] = s0; y[3 * p + 1] = s1; y[3 * p + 2] = s2;
        const int rn = cr ? p - nV : p;                 // rotation node of this P2 node, if it carries one
        if (rn >= 0 && rn < nrot) {
            y[ndof_u + 3 * rn] = t0; y[ndof_u + 3 * rn + 1] = t1; y[ndof_u + 3 * rn + 2] = t2;
        }
    } else {
        const int g = ndof_u + 3 * nrot + (p - nP2);    // ghost entries: no element touches them
        if (g < ndof) y[g] = 0.0;
    }
}

// blocks that share an XCD (blockIdx % 8, observed round-robin placement -- a speed hint only) work on a
// contiguous range of the element order, so that their gathers hit the same L2
__device__ __forceinline__ int xcd_block(int b, int nb) {
    const int per = (nb + 7) / 8;
    const int lb = (b & 7) * per + (b >> 3);
    return lb;
}

// LPE lanes per element.  4: the lanes of a DPP quad (16 elements per wave; the default).  5 (option apply_lanes, an experiment of
// round 6): twelve elements on 60 lanes of a wave -- for the 25 points of the 5 x 5 rule, which four lanes share as 7 + 6 + 6 + 6 (the
// quad waits for the lane with seven) and five as 5 each; the partial results then meet through ds_bpermute (the LDS crossbar, no LDS
// memory) instead of DPP.  Measured slower (wing1m: 118.6 against 103.5 us per application; 4 x 4 points: 39.2 against 35.2): the kernel
// is bound by the number of vector instructions it issues (the ALU issues ~62 % of the time at two waves per SIMD), not by the slowest
// lane of a quad, and five lanes issue more in total -- a third more waves with the same staging, reduction and stores
// (profiles/r6_apply_lanes.txt).
__device__ __forceinline__ double lane_get(double v, int src_lane) {
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
// sum over the five lanes base .. base + 4 of a group (s1, s2, s4: the lanes 1, 2 and 4 places further round the group)
__device__ __forceinline__ double group5_sum(double v, int s1, int s2, int s4) {
    const double t = v + lane_get(v, s1);
    const double u = t + lane_get(t, s2);
    return u + lane_get(v, s4);
}
__device__ __host__ constexpr int apply_epb(int lpe) { return 4 * (64 / lpe); }      // elements per block of four waves

template <int NPC, int NVC, bool QUAD, bool UHAT, bool MASS, int LPE>
__global__ void __launch_bounds__(256, 2)
k_apply4(MeshDev m, FieldsDev f, const Tables* __restrict__ tab, const int* __restrict__ eorder, double aK, double aM,
         const double* __restrict__ x, double* __restrict__ ybuf, double* dotslot, double* zero_a, double* zero_b) {
    constexpr int LD = 3 * NPC + 3 * NVC;
    constexpr int EPW = 64 / LPE;               // elements per wave
    constexpr int EPB = apply_epb(LPE);         // elements per block (four waves)
    constexpr int GEO = 3 * NVC + 3 * NVC + 4 * NVC;    // per element: X, uhat, (h, E, nu, rho) at the vertices
    // (r4: SQ_LDS_BANK_CONFLICT is 70 % of this kernel's LDS-active cycles; rows of an odd number of doubles -- 16 cells on 16 different
    //  bank pairs instead of every fourth cell colliding at 40 doubles = 64 B modulo 256 B -- changed nothing: 109.7 against 108 us.
    //  The LDS pipe is not what the kernel waits for.)
    __shared__ double sx[EPB][LD + 1];
    __shared__ double sg[EPB][GEO + 1];
    __shared__ double scm[MASS ? EPB : 1][MASS ? MAXQ : 1];     // inertia coefficient per (element, point); written and read by the same lane
    // the lanes of a quad work on different quadrature points, so the tables are indexed per lane: keep
    // them in LDS (a per-lane global/scalar load would park ~40 doubles of table data in VGPRs)
    __shared__ Tables stab;
    {
        const double* src = reinterpret_cast<const double*>(tab);
        double* dst = reinterpret_cast<double*>(&stab);
        for (int i = threadIdx.x; i < (int)(sizeof(Tables) / sizeof(double)); i += blockDim.x) dst[i] = src[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (zero_a) *zero_a = 0.0;
        if (zero_b) *zero_b = 0.0;
    }
    const int lb = xcd_block(blockIdx.x, gridDim.x);
    const int lane = threadIdx.x & 63;
    const int le = LPE == 4 ? threadIdx.x >> 2 : (threadIdx.x >> 6) * EPW + min(lane / LPE, EPW - 1);
    const int sub = LPE == 4 ? threadIdx.x & 3 : lane % LPE;
    const int pos = lb * EPB + le;
    const bool active = lb * EPB < m.nel && pos < m.nel && (LPE == 4 || lane < EPW * LPE);
    double local = 0.0;
    const int e = active ? (eorder ? eorder[pos] : pos) : 0;
    double hK = 0.0;
    if (active) {
        // lane `sub` of the quad stages vertex `sub` (geometry, fields) and a quarter of the element vector in LDS:
        // nothing element-specific stays in registers across the quadrature loop except the 39 partial results
        hK = m.hK[e];
        int pid[NPC], vid[NVC];
#pragma unroll
        for (int b = 0; b < NVC; ++b) vid[b] = m.cells[b * m.nel + e];
#pragma unroll
        for (int a = 0; a < NPC; ++a) pid[a] = m.cellp2[a * m.nel + e];
#pragma unroll
        for (int b = 0; b < NVC; ++b) {
            if (b == sub || (NVC < 4 && sub == 3 && b == 0)) {
                if (b == sub) {
                    const int v = vid[b];
                    const int tq = f.ewm ? e : v;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        sg[le][3 * b + c] = m.xyz[3 * v + c];
                        sg[le][3 * NVC + 3 * b + c] = UHAT ? f.uhat[3 * v + c] : 0.0;
                    }
                    sg[le][6 * NVC + b] = f.h[tq];
                    sg[le][7 * NVC + b] = f.E[tq];
                    sg[le][8 * NVC + b] = f.nu[tq];
                    sg[le][9 * NVC + b] = MASS ? f.rho[tq] : 0.0;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < LD; ++i) {
            if ((i % LPE) == sub) {
                const int node = i / 3, c = i - 3 * node;
                const int rb = node >= NPC ? node - NPC : 0;
                const int rnode = (NPC == 6 && NVC == 3 && m.cr) ? pid[(NVC + rb) % NPC] - m.nn : vid[rb];     // CG2CR1: the edge midpoint
                const int g = node < NPC ? 3 * pid[node < NPC ? node : 0] + c : m.ndof_u + 3 * rnode + c;
                sx[le][i] = x[g];
            }
        }
    }
    __syncthreads();
    if (active) {
        double ye[LD];
#pragma unroll
        for (int i = 0; i < LD; ++i) ye[i] = 0.0;
        const int nq = stab.nq;
        for (int q = sub; q < nq; q += LPE) {
            // re-derive the LDS rows every iteration: keeps the compiler from hoisting the staged values
            // out of the loop into registers
            int row = le;
            asm volatile("" : "+v"(row));
            const double* xe = sx[row];
            const double* ge = sg[row];
            const double (*X)[3] = reinterpret_cast<const double (*)[3]>(ge);
            const double (*Uh)[3] = reinterpret_cast<const double (*)[3]>(ge + 3 * NVC);
            QPG g;
            qp_geometry<NVC, QUAD, UHAT>(X, Uh, stab.N1[q], stab.dN1[q], g);
            Mat mat, ex;
            const double hq = interp<NVC>(stab.N1[q], ge + 6 * NVC);
            material<DERIV_NONE>(hq, interp<NVC>(stab.N1[q], ge + 7 * NVC), interp<NVC>(stab.N1[q], ge + 8 * NVC), hK,
                                 stab.wS[q] * g.det, stab.w[q] * g.det, g.Ju, mat, ex);
            mat.cm *= aK; mat.cb *= aK; mat.cs *= aK; mat.cd *= aK;
            const Gen s = strains_q<NPC, NVC>(stab, q, g, xe);
            const Gen t = stress_of(s, mat);
            strains_T_q<NPC, NVC>(stab, q, g, t, ye);
            if (MASS)            // the inertia term's coefficient at this point, parked in LDS for the loop below
                scm[row][q] = aM * interp<NVC>(stab.N1[q], ge + 9 * NVC) * hq * stab.w[q] * g.det * g.Ju;
        }
        if (MASS) {
            // The inertia term in a loop of its own (compiled out of the static operator): inside the stiffness loop its interpolated
            // vectors pushed the kernel from 242 registers to 256 + 148 B of scratch
            for (int q = sub; q < nq; q += LPE) {
                int row = le;
                asm volatile("" : "+v"(row));
                mass_qp<NPC, NVC>(stab, q, scm[row][q], hK, sx[row], ye);
            }
        }
        if (LPE == 4) {
#pragma unroll
            for (int i = 0; i < LD; ++i) ye[i] = quad_xor_sum(ye[i]);
        } else {
            const int base = lane - sub;
            const int s1 = base + (sub + 1) % LPE, s2 = base + (sub + 2) % LPE, s4 = base + (sub + 4) % LPE;
#pragma unroll
            for (int i = 0; i < LD; ++i) ye[i] = group5_sum(ye[i], s1, s2, s4);
        }
        // lane `sub` owns the outputs i == sub (mod LPE); the element's 39 results go to its own slot of ybuf
        // (plain stores, 320 contiguous bytes per element) -- k_gather_sum adds them up per node: no atomics,
        // and a fixed summation order
        double* out = ybuf + (size_t)pos * YSTRIDE;
#pragma unroll
        for (int i = 0; i < LD; ++i) {
            if ((i % LPE) == sub) {
                out[i] = ye[i];
                local += sx[le][i] * ye[i];
            }
        }
    }
    if (dotslot) block_accumulate(local, dotslot);
}

// diag += diag(K_elastic)
template <int NPC, int NVC, bool QUAD, bool UHAT>
__global__ void __launch_bounds__(128)
k_diag(MeshDev m, FieldsDev f, const Tables* __restrict__ tab, double* __restrict__ diag) {
    constexpr int LD = 3 * NPC + 3 * NVC;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m.nel) return;
    Elem<NPC, NVC> el;
    load_elem<NPC, NVC, UHAT>(m, f, e, el);
    double de[LD];
#pragma unroll
    for (int i = 0; i < LD; ++i) de[i] = 0.0;
    const int nq = tab->nq;
    for (int q = 0; q < nq; ++q) {
        QPG g;
        qp_geometry<NVC, QUAD, UHAT>(el.X, el.Uh, tab->N1[q], tab->dN1[q], g);
        double d[NPC][2], mm[NVC][2];
        local_derivs<NPC, NVC>(*tab, q, g.Q, d, mm);
        Mat mat, ex;
        material<DERIV_NONE>(interp<NVC>(tab->N1[q], el.hn), interp<NVC>(tab->N1[q], el.En),
                             interp<NVC>(tab->N1[q], el.nun), el.hK, tab->wS[q] * g.det, tab->w[q] * g.det, g.Ju, mat, ex);
        const double sh = 0.5 * (1.0 - mat.nu);
#pragma unroll
        for (int a = 0; a < NPC; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double b0 = g.E0[c] * d[a][0], b1 = g.E1[c] * d[a][1];
                const double b2 = g.E0[c] * d[a][1] + g.E1[c] * d[a][0];
                const double b6 = g.E2[c] * d[a][0], b7 = g.E2[c] * d[a][1];
                const double b8 = 0.5 * (g.E0[c] * d[a][1] - g.E1[c] * d[a][0]);
                de[3 * a + c] += mat.cm * (b0 * b0 + 2.0 * mat.nu * b0 * b1 + b1 * b1 + sh * b2 * b2) +
                                 mat.cs * (b6 * b6 + b7 * b7) + mat.cd * b8 * b8;
            }
        double x00[3], x01[3], x10[3], x11[3];
        cross3(g.E0, g.w0, x00);
        cross3(g.E0, g.w1, x01);
        cross3(g.E1, g.w0, x10);
        cross3(g.E1, g.w1, x11);
#pragma unroll
        for (int b = 0; b < NVC; ++b) {
            const double Mb = tab->NR[q][b];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double k00 = -g.E1[c] * mm[b][0] + Mb * x00[c];
                const double k11 = g.E0[c] * mm[b][1] + Mb * x11[c];
                const double k01 = -g.E1[c] * mm[b][1] + Mb * x01[c] + g.E0[c] * mm[b][0] + Mb * x10[c];
                const double g0 = Mb * g.E1[c], g1 = -Mb * g.E0[c], om = Mb * g.E2[c];
                de[3 * NPC + 3 * b + c] += mat.cb * (k00 * k00 + 2.0 * mat.nu * k00 * k11 + k11 * k11 + sh * k01 * k01) +
                                           mat.cs * (g0 * g0 + g1 * g1) + mat.cd * om * om;
            }
        }
    }
#pragma unroll
    for (int a = 0; a < NPC; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) atomicAdd(&diag[3 * el.pid[a] + c], de[3 * a + c]);
#pragma unroll
    for (int b = 0; b < NVC; ++b)
#pragma unroll
        for (int c = 0; c < 3; ++c) atomicAdd(&diag[m.ndof_u + 3 * rot_node(m, el, b) + c], de[3 * NPC + 3 * b + c]);
}

// F += int N_a f J dx   (sign: the residual subtracts it)
template <int NPC, int NVC, bool QUAD, bool UHAT>
__global__ void __launch_bounds__(128)
k_load(MeshDev m, FieldsDev f, const Tables* __restrict__ tab, double* __restrict__ F, double scale) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m.nel) return;
    Elem<NPC, NVC> el;
    load_elem<NPC, NVC, UHAT>(m, f, e, el);
    double fn[NVC][3];
#pragma unroll
    for (int b = 0; b < NVC; ++b)
#pragma unroll
        for (int c = 0; c < 3; ++c) fn[b][c] = f.f[3 * (f.ewp ? e : el.vid[b]) + c];
    double Fe[3 * NPC];
#pragma unroll
    for (int i = 0; i < 3 * NPC; ++i) Fe[i] = 0.0;
    const int nq = tab->nq;
    for (int q = 0; q < nq; ++q) {
        QPG g;
        qp_geometry<NVC, QUAD, UHAT>(el.X, el.Uh, tab->N1[q], tab->dN1[q], g);
        const double wj = tab->w[q] * g.det * g.Ju;
        double fq[3] = {0, 0, 0};
#pragma unroll
        for (int b = 0; b < NVC; ++b)
#pragma unroll
            for (int c = 0; c < 3; ++c) fq[c] += tab->N1[q][b] * fn[b][c];
#pragma unroll
        for (int a = 0; a < NPC; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) Fe[3 * a + c] += wj * tab->N2[q][a] * fq[c];
    }
#pragma unroll
    for (int a = 0; a < NPC; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) atomicAdd(&F[3 * el.pid[a] + c], scale * Fe[3 * a + c]);
}

// scalar functionals: slot[0] += int u.u J dx ; slot[1] += regularisation ; slot[2] += mass ; slot[3] += volume
template <int NPC, int NVC, bool QUAD, bool UHAT>
__global__ void __launch_bounds__(128)
k_functionals(MeshDev m, FieldsDev f, const Tables* __restrict__ tab, const double* __restrict__ w, double* slots) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    double uu = 0.0, reg = 0.0, mass = 0.0, vol = 0.0, area = 0.0;
    if (e < m.nel && cell_selected(m, e)) {
        Elem<NPC, NVC> el;
        load_elem<NPC, NVC, UHAT>(m, f, e, el);
        double ue[3 * NPC];
#pragma unroll
        for (int a = 0; a < NPC; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) ue[3 * a + c] = w[3 * el.pid[a] + c];
        double rhon[NVC];
#pragma unroll
        for (int b = 0; b < NVC; ++b) rhon[b] = f.rho[f.ewm ? e : el.vid[b]];
        const int nq = tab->nq;
        for (int q = 0; q < nq; ++q) {
            QPG g;
            qp_geometry<NVC, QUAD, UHAT>(el.X, el.Uh, tab->N1[q], tab->dN1[q], g);
            const double wd = tab->w[q] * g.det;
            double uq[3] = {0, 0, 0};
#pragma unroll
            for (int a = 0; a < NPC; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) uq[c] += tab->N2[q][a] * ue[3 * a + c];
            uu += wd * g.Ju * dot3(uq, uq);
            const double hq = interp<NVC>(tab->N1[q], el.hn);
            mass += wd * g.Ju * hq * interp<NVC>(tab->N1[q], rhon);
            vol += wd * g.Ju * hq;                          // int h J dx (dynamic_rm_shell/volume_operation.py:68-70)
            area += wd * g.Ju;                              // int J dx (rm_shell_pde.py:104-105)
            if (f.ewm) {
                reg += 0.5 * REG_ALPHA1 * wd * hq * hq;                    // L2, rm_shell_pde.py:79-81
            } else {
                double g0 = 0.0, g1 = 0.0;                                  // H1, rm_shell_pde.py:72-74
#pragma unroll
                for (int b = 0; b < NVC; ++b) {
                    g0 += el.hn[b] * (tab->dN1[q][b][0] * g.Q0[0][0] + tab->dN1[q][b][1] * g.Q0[1][0]);
                    g1 += el.hn[b] * (tab->dN1[q][b][0] * g.Q0[0][1] + tab->dN1[q][b][1] * g.Q0[1][1]);
                }
                reg += 0.5 * REG_ALPHA1 * wd * (g0 * g0 + g1 * g1);
            }
        }
    }
    block_accumulate(uu, slots + 0);
    block_accumulate(reg, slots + 1);
    block_accumulate(mass, slots + 2);
    block_accumulate(vol, slots + 3);
    block_accumulate(area, slots + 4);
}

// out_u += 2 int N_a u J dx  (d compliance / d w)
template <int NPC, int NVC, bool QUAD, bool UHAT>
__global__ void __launch_bounds__(128)
k_dcompliance_du(MeshDev m, FieldsDev f, const Tables* __restrict__ tab, const double* __restrict__ w,
                 double* __restrict__ out, double scale = 1.0) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m.nel || !cell_selected(m, e)) return;
    Elem<NPC, NVC> el;
    load_elem<NPC, NVC, UHAT>(m, f, e, el);
    double ue[3 * NPC], ge[3 * NPC];
#pragma unroll
    for (int a = 0; a < NPC; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            ue[3 * a + c] = w[3 * el.pid[a] + c];
            ge[3 * a + c] = 0.0;
        }
    const int nq = tab->nq;
    for (int q = 0; q < nq; ++q) {
        QPG g;
        qp_geometry<NVC, QUAD, UHAT>(el.X, el.Uh, tab->N1[q], tab->dN1[q], g);
        const double wj = 2.0 * scale * tab->w[q] * g.det * g.Ju;
        double uq[3] = {0, 0, 0};
#pragma unroll
        for (int a = 0; a < NPC; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) uq[c] += tab->N2[q][a] * ue[3 * a + c];
#pragma unroll
        for (int a = 0; a < NPC; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) ge[3 * a + c] += wj * tab->N2[q][a] * uq[c];
    }
#pragma unroll
    for (int a = 0; a < NPC; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) atomicAdd(&out[3 * el.pid[a] + c], ge[3 * a + c]);
}

// field-space gradients that need no state:  mode 0: d reg / d h ; 1: d mass / d h ; 2: d mass / d rho ; 3: d volume / d h
template <int NPC, int NVC, bool QUAD, bool UHAT>
__global__ void __launch_bounds__(128)
k_field_grad(MeshDev m, FieldsDev f, const Tables* __restrict__ tab, int mode, double* __restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m.nel) return;
    Elem<NPC, NVC> el;
    load_elem<NPC, NVC, UHAT>(m, f, e, el);
    double rhon[NVC], ge[NVC];
#pragma unroll
    for (int b = 0; b < NVC; ++b) {
        rhon[b] = f.rho[f.ewm ? e : el.vid[b]];
        ge[b] = 0.0;
    }
    const int nq = tab->nq;
    for (int q = 0; q < nq; ++q) {
        QPG g;
        qp_geometry<NVC, QUAD, UHAT>(el.X, el.Uh, tab->N1[q], tab->dN1[q], g);
        const double wd = tab->w[q] * g.det;
        if (mode == 0) {
            if (f.ewm) {
                ge[0] += REG_ALPHA1 * wd * el.hn[0];
            } else {
                double g0 = 0.0, g1 = 0.0, d0[NVC], d1[NVC];
#pragma unroll
                for (int b = 0; b < NVC; ++b) {
                    d0[b] = tab->dN1[q][b][0] * g.Q0[0][0] + tab->dN1[q][b][1] * g.Q0[1][0];
                    d1[b] = tab->dN1[q][b][0] * g.Q0[0][1] + tab->dN1[q][b][1] * g.Q0[1][1];
                    g0 += el.hn[b] * d0[b];
                    g1 += el.hn[b] * d1[b];
                }
#pragma unroll
                for (int b = 0; b < NVC; ++b) ge[b] += REG_ALPHA1 * wd * (d0[b] * g0 + d1[b] * g1);
            }
        } else {
            const double other = (mode == 1) ? interp<NVC>(tab->N1[q], rhon) : (mode == 2 ? interp<NVC>(tab->N1[q], el.hn) : 1.0);
            if (f.ewm) {
                ge[0] += wd * g.Ju * other;
            } else {
#pragma unroll
                for (int b = 0; b < NVC; ++b) ge[b] += wd * g.Ju * other * tab->N1[q][b];
            }
        }
    }
    if (f.ewm) {
        out[e] += ge[0];
    } else {
#pragma unroll
        for (int b = 0; b < NVC; ++b) atomicAdd(&out[el.vid[b]], ge[b]);
    }
}

// out += scale * lam^T (dK/d field) w  per field DOF  (field = h, E or nu)
template <int NPC, int NVC, bool QUAD, bool UHAT, int WHICH>
__global__ void __launch_bounds__(128)
k_dRdfield_T(MeshDev m, FieldsDev f, const Tables* __restrict__ tab, const double* __restrict__ w,
             const double* __restrict__ lam, double scale, double* __restrict__ out) {
    constexpr int LD = 3 * NPC + 3 * NVC;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m.nel) return;
    Elem<NPC, NVC> el;
    load_elem<NPC, NVC, UHAT>(m, f, e, el);
    double we[LD], le[LD];
#pragma unroll
    for (int a = 0; a < NPC; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            we[3 * a + c] = w[3 * el.pid[a] + c];
            le[3 * a + c] = lam[3 * el.pid[a] + c];
        }
#pragma unroll
    for (int b = 0; b < NVC; ++b)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            we[3 * NPC + 3 * b + c] = w[m.ndof_u + 3 * rot_node(m, el, b) + c];
            le[3 * NPC + 3 * b + c] = lam[m.ndof_u + 3 * rot_node(m, el, b) + c];
        }
    double ge[NVC];
#pragma unroll
    for (int b = 0; b < NVC; ++b) ge[b] = 0.0;
    const int nq = tab->nq;
    for (int q = 0; q < nq; ++q) {
        QPG g;
        qp_geometry<NVC, QUAD, UHAT>(el.X, el.Uh, tab->N1[q], tab->dN1[q], g);
        double d[NPC][2], mm[NVC][2];
        local_derivs<NPC, NVC>(*tab, q, g.Q, d, mm);
        Mat mat, ex;
        material<WHICH>(interp<NVC>(tab->N1[q], el.hn), interp<NVC>(tab->N1[q], el.En),
                        interp<NVC>(tab->N1[q], el.nun), el.hK, tab->wS[q] * g.det, tab->w[q] * g.det, g.Ju, mat, ex);
        const Gen sw = strains<NPC, NVC>(g, d, mm, tab->NR[q], we);
        const Gen sl = strains<NPC, NVC>(g, d, mm, tab->NR[q], le);
        Gen t = stress_of(sw, mat);
        if (WHICH == DERIV_NU) stress_add_dnu(sw, ex, t);
        const double dens = gen_dot(t, sl);
        if (f.ewm) {
            ge[0] += dens;
        } else {
#pragma unroll
            for (int b = 0; b < NVC; ++b) ge[b] += dens * tab->N1[q][b];
        }
    }
    if (f.ewm) {
        out[e] += scale * ge[0];
    } else {
#pragma unroll
        for (int b = 0; b < NVC; ++b) atomicAdd(&out[el.vid[b]], scale * ge[b]);
    }
}

// y += (aK dK/dh[dh] + aM dM/dh[dh]) x : the directional derivative of the step operator along a thickness perturbation dh
// (forward mode of the transient operator, state_operation_dynamic.py:295-316: dRdt assembled per level and multiplied by
// d_inputs['thickness']) -- matrix-free, one thread per cell.  Not a hot kernel: T launches per forward-mode product.
template <int NPC, int NVC, bool QUAD, bool UHAT>
__global__ void __launch_bounds__(128)
k_apply_dh(MeshDev m, FieldsDev f, const Tables* __restrict__ tab, const double* __restrict__ dh, double aK, double aM,
           const double* __restrict__ x, double* __restrict__ y) {
    constexpr int LD = 3 * NPC + 3 * NVC;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m.nel) return;
    Elem<NPC, NVC> el;
    load_elem<NPC, NVC, UHAT>(m, f, e, el);
    double xe[LD], ye[LD], dhn[NVC];
    for (int a = 0; a < NPC; ++a)
        for (int c = 0; c < 3; ++c) xe[3 * a + c] = x[3 * el.pid[a] + c];
    for (int b = 0; b < NVC; ++b) {
        for (int c = 0; c < 3; ++c) xe[3 * NPC + 3 * b + c] = x[m.ndof_u + 3 * rot_node(m, el, b) + c];
        dhn[b] = dh[f.ewm ? e : el.vid[b]];
    }
    for (int i = 0; i < LD; ++i) ye[i] = 0.0;
    const int nq = tab->nq;
    for (int q = 0; q < nq; ++q) {
        QPG g;
        qp_geometry<NVC, QUAD, UHAT>(el.X, el.Uh, tab->N1[q], tab->dN1[q], g);
        const double hq = interp<NVC>(tab->N1[q], el.hn), dq = interp<NVC>(tab->N1[q], dhn);
        if (aK != 0.0) {
            Mat mat, ex;
            material<DERIV_H>(hq, interp<NVC>(tab->N1[q], el.En), interp<NVC>(tab->N1[q], el.nun), el.hK, tab->wS[q] * g.det,
                              tab->w[q] * g.det, g.Ju, mat, ex);
            const double sc = aK * dq;
            mat.cm *= sc; mat.cb *= sc; mat.cs *= sc; mat.cd *= sc;
            const Gen s = strains_q<NPC, NVC>(*tab, q, g, xe);
            const Gen t = stress_of(s, mat);
            strains_T_q<NPC, NVC>(*tab, q, g, t, ye);
        }
        if (aM != 0.0 && tab->w[q] != 0.0) {
            double rq = 0.0;
            for (int b = 0; b < NVC; ++b) rq += tab->N1[q][b] * f.rho[f.ewm ? e : el.vid[b]];
            mass_qp<NPC, NVC>(*tab, q, aM * rq * dq * tab->w[q] * g.det * g.Ju, el.hK, xe, ye);
        }
    }
    for (int a = 0; a < NPC; ++a)
        for (int c = 0; c < 3; ++c) atomicAdd(&y[3 * el.pid[a] + c], ye[3 * a + c]);
    for (int b = 0; b < NVC; ++b)
        for (int c = 0; c < 3; ++c) atomicAdd(&y[m.ndof_u + 3 * rot_node(m, el, b) + c], ye[3 * NPC + 3 * b + c]);
}

// out += scale * y^T (dM/dh) x  per thickness DOF:  int rho M_b (x_u.y_u + h_K^2 x_theta.y_theta) J dx
template <int NPC, int NVC, bool QUAD, bool UHAT>
__global__ void __launch_bounds__(128)
k_dMdh_T(MeshDev m, FieldsDev f, const Tables* __restrict__ tab, const double* __restrict__ x, const double* __restrict__ y,
         double scale, double* __restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m.nel) return;
    Elem<NPC, NVC> el;
    load_elem<NPC, NVC, UHAT>(m, f, e, el);
    double ge[NVC];
    for (int b = 0; b < NVC; ++b) ge[b] = 0.0;
    const int nq = tab->nq;
    for (int q = 0; q < nq; ++q) {
        if (tab->w[q] == 0.0) continue;
        QPG g;
        qp_geometry<NVC, QUAD, UHAT>(el.X, el.Uh, tab->N1[q], tab->dN1[q], g);
        double xu[3] = {0, 0, 0}, yu[3] = {0, 0, 0}, xt[3] = {0, 0, 0}, yt[3] = {0, 0, 0}, rq = 0.0;
        for (int a = 0; a < NPC; ++a)
            for (int c = 0; c < 3; ++c) {
                xu[c] += tab->N2[q][a] * x[3 * el.pid[a] + c];
                yu[c] += tab->N2[q][a] * y[3 * el.pid[a] + c];
            }
        for (int b = 0; b < NVC; ++b) {
            rq += tab->N1[q][b] * f.rho[f.ewm ? e : el.vid[b]];
            for (int c = 0; c < 3; ++c) {
                xt[c] += tab->NR[q][b] * x[m.ndof_u + 3 * rot_node(m, el, b) + c];
                yt[c] += tab->NR[q][b] * y[m.ndof_u + 3 * rot_node(m, el, b) + c];
            }
        }
        const double dens = tab->w[q] * g.det * g.Ju * rq * (dot3(xu, yu) + el.hK * el.hK * dot3(xt, yt));
        if (f.ewm) ge[0] += dens;
        else
            for (int b = 0; b < NVC; ++b) ge[b] += dens * tab->N1[q][b];
    }
    if (f.ewm) out[e] += scale * ge[0];
    else
        for (int b = 0; b < NVC; ++b) atomicAdd(&out[el.vid[b]], scale * ge[b]);
}

// out += scale * int M_b lam_u J dx    ((dR/df)^T lam uses scale = -1)
template <int NPC, int NVC, bool QUAD, bool UHAT>
__global__ void __launch_bounds__(128)
k_dRdf_T(MeshDev m, FieldsDev f, const Tables* __restrict__ tab, const double* __restrict__ lam, double scale,
         double* __restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m.nel) return;
    Elem<NPC, NVC> el;
    load_elem<NPC, NVC, UHAT>(m, f, e, el);
    double le[3 * NPC];
#pragma unroll
    for (int a = 0; a < NPC; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) le[3 * a + c] = lam[3 * el.pid[a] + c];
    double ge[NVC][3];
#pragma unroll
    for (int b = 0; b < NVC; ++b)
#pragma unroll
        for (int c = 0; c < 3; ++c) ge[b][c] = 0.0;
    const int nq = tab->nq;
    for (int q = 0; q < nq; ++q) {
        QPG g;
        qp_geometry<NVC, QUAD, UHAT>(el.X, el.Uh, tab->N1[q], tab->dN1[q], g);
        const double wj = tab->w[q] * g.det * g.Ju;
        double lq[3] = {0, 0, 0};
#pragma unroll
        for (int a = 0; a < NPC; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) lq[c] += tab->N2[q][a] * le[3 * a + c];
        if (f.ewp) {
#pragma unroll
            for (int c = 0; c < 3; ++c) ge[0][c] += wj * lq[c];
        } else {
#pragma unroll
            for (int b = 0; b < NVC; ++b)
#pragma unroll
                for (int c = 0; c < 3; ++c) ge[b][c] += wj * tab->N1[q][b] * lq[c];
        }
    }
    if (f.ewp) {
#pragma unroll
        for (int c = 0; c < 3; ++c) out[3 * e + c] += scale * ge[0][c];
    } else {
#pragma unroll
        for (int b = 0; b < NVC; ++b)
#pragma unroll
            for (int c = 0; c < 3; ++c) atomicAdd(&out[3 * el.vid[b] + c], scale * ge[b][c]);
    }
}

// dense element matrices, one wave per element: K_e[i][j] = sum_q B_i^T C B_j
// Column j of K_e is the operator applied to the unit vector e_j, so lanes own columns.
template <int NPC, int NVC, bool QUAD, bool UHAT>
__global__ void __launch_bounds__(64)          // 238 registers, two waves per SIMD (three: 1.05 against 0.95 ms; four: 3.4 ms, 212 B of scratch)
k_element_matrices(MeshDev m, FieldsDev f, const Tables* __restrict__ tab, int first, int count, double* __restrict__ Ke) {
    constexpr int LD = 3 * NPC + 3 * NVC;
    const int le = blockIdx.x;
    if (le >= count) return;
    const int e = first + le;
    const int j = threadIdx.x;
    Elem<NPC, NVC> el;
    load_elem<NPC, NVC, UHAT>(m, f, e, el);
    extern __shared__ double sq_raw[];
    QPoint<NPC, NVC>* sq = reinterpret_cast<QPoint<NPC, NVC>*>(sq_raw);     // nq points (stage_qpoints)
    stage_qpoints<NPC, NVC, QUAD, UHAT>(tab, el, 1.0, j, 64, sq);
    if (j >= LD) return;
    double ye[LD];
#pragma unroll
    for (int i = 0; i < LD; ++i) ye[i] = 0.0;
    // the lane's unit vector e_j: displacement component cj of P2 node aj, or rotation component cj of vertex aj -- its strains
    // without the 39-entry reduction (as k_front_assemble does)
    const bool is_u = j < 3 * NPC;
    const int aj = is_u ? j / 3 : (j - 3 * NPC) / 3;
    const int cj = j - 3 * (is_u ? aj : NPC + aj);
    const int nq = tab->nq;
    for (int q = 0; q < nq; ++q) {
        const QPoint<NPC, NVC>& p = sq[q];
        const double r0 = is_u ? tab->dN2[q][aj][0] : tab->dNR[q][aj][0], r1 = is_u ? tab->dN2[q][aj][1] : tab->dNR[q][aj][1];
        const double dk0 = r0 * p.g.Q[0][0] + r1 * p.g.Q[1][0], dk1 = r0 * p.g.Q[0][1] + r1 * p.g.Q[1][1];
        const double Mj = is_u ? 0.0 : tab->NR[q][aj];
        double G0[3], G1[3], th[3], T0[3], T1[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double ec = (c == cj) ? 1.0 : 0.0;
            G0[c] = is_u ? dk0 * ec : 0.0;
            G1[c] = is_u ? dk1 * ec : 0.0;
            th[c] = Mj * ec;
            T0[c] = is_u ? 0.0 : dk0 * ec;
            T1[c] = is_u ? 0.0 : dk1 * ec;
        }
        const Gen s = strains_reduced(p.g, G0, G1, th, T0, T1);
        const Gen t = stress_of(s, p.mat);
        strains_T<NPC, NVC>(p.g, p.d, p.mm, tab->NR[q], t, ye);
    }
    double* out = Ke + (size_t)le * LD * LD;
#pragma unroll
    for (int i = 0; i < LD; ++i) out[i * LD + j] = ye[i];
}

// ------------------------------------------------------------------------------------------ CSR assembly
// Scatter-add of the element matrices into CSR values with wave-level segmentation: the nel*LD*LD element
// contributions are visited in the order of their CSR destination (perm), so the 64 lanes of a wave hold runs of
// equal destinations.  A ballot of the run heads drives a segmented shuffle scan; the last lane of every run
// writes the run's sum -- a plain, coalesced store when the run lies inside the wave, an atomic add only for runs
// cut by a wave boundary.  (Scattered fp64 atomics run at ~10 G/s on this chip; sorted destinations avoid them.)
// A wave takes CSR_R consecutive chunks of 64 contributions: all its index loads and gathers are in flight together (the kernel
// is a chain index -> gather -> scan -> store: with one contribution per lane it ran at 1.1 TB/s of its own traffic), runs that
// cross a chunk boundary inside the wave are joined in registers, and only the runs at the two ends of the wave's 512
// contributions need an atomic.
constexpr int CSR_R = 8;       // measured at 1M DOF: 1 chunk 2.0 ms, 4 chunks 1.18 ms, 8 chunks 0.96 ms (memset of the values included)
__global__ void __launch_bounds__(256)
k_csr_segmented(long long ncontrib, const int* __restrict__ perm, const int* __restrict__ dest,
                const double* __restrict__ Ke, double* __restrict__ vals) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long base = ((long long)blockIdx.x * 4 + wv) * (64 * CSR_R);
    if (base >= ncontrib) return;
    int d[CSR_R], p[CSR_R];
    double v[CSR_R];
#pragma unroll
    for (int r = 0; r < CSR_R; ++r) {
        const long long t = base + 64 * r + lane;
        const bool act = t < ncontrib;
        d[r] = act ? dest[t] : -1 - r;                    // inactive lanes: a run of their own that is never stored
        p[r] = act ? perm[t] : 0;
    }
#pragma unroll
    for (int r = 0; r < CSR_R; ++r) v[r] = d[r] >= 0 ? Ke[p[r]] : 0.0;
    int d_prev_last = -100;                               // destination of lane 63 of the chunk before (none before chunk 0)
    double carry = 0.0;                                   // sum of the run that ends the chunk before, if it goes on
    bool carry_from_start = true;                         // ... and whether that run reaches back to the wave's first contribution
#pragma unroll
    for (int r = 0; r < CSR_R; ++r) {
        const int dr = d[r];
        const int d_prev = __shfl_up(dr, 1, 64);
        const bool head = lane == 0 || dr != d_prev;
        const unsigned long long heads = __ballot(head);
        const unsigned long long below = heads & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
        const int my_head = 63 - __clzll((long long)below);
        double x = v[r];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double up = __shfl_up(x, o, 64);
            if (lane - o >= my_head) x += up;
        }
        const int d0 = __builtin_amdgcn_readfirstlane(dr);
        const bool joins = r > 0 && d0 == d_prev_last;                 // the chunk's first run continues the previous chunk's last
        const bool first_run = my_head == 0;
        if (first_run && joins) x += carry;
        // does my run reach back to the wave's first contribution?  (then another wave may hold the start of it: atomic)
        const bool from_start = first_run && (r == 0 || (joins && carry_from_start));
        const int d_next_in = __shfl_down(dr, 1, 64);
        const int d_next0 = r + 1 < CSR_R ? __builtin_amdgcn_readfirstlane(d[r + 1]) : -200;
        const int d_next = lane == 63 ? d_next0 : d_next_in;
        const bool last_of_wave = r == CSR_R - 1 && lane == 63;
        const bool tail = last_of_wave || dr != d_next;
        if (tail && dr >= 0) {
            if (from_start || last_of_wave) atomicAdd(&vals[dr], x);
            else vals[dr] = x;
        }
        // hand the last run over to the next chunk
        d_prev_last = __builtin_amdgcn_readlane(dr, 63);
        carry = lane_bcast(x, 63);
        const int head63 = __builtin_amdgcn_readlane(my_head, 63);
        carry_from_start = head63 == 0 && (r == 0 || (joins && carry_from_start));
    }
}

// ------------------------------------------------------------------------------------------ penalty facets
struct FacetDev {
    int nf;
    const int* cell;     // nf
    const int* ledge;    // nf
    const int* unode;    // nf*3  P2 nodes (a, mid, b)
    const int* vnode;    // nf*2  vertices (a, b)
    double* M2;          // nf*9  beta/h_K * int N2_i N2_j |J F^-T N| ds
    double* M1;          // nf*4
    // CG2CR1 only (null otherwise): the rotation's trace on a facet involves ALL THREE Crouzeix-Raviart functions of the cell (on edge k:
    // NR_k = 1, NR_(k+1) = s, NR_(k+2) = -s, s in [-1, 1] along the edge), so its block is 3 x 3 over the cell's three edge midpoints
    const int* rnode;    // nf*3  rotation nodes of the facet's cell, in the cell's local order
    double* MR;          // nf*9
};

__device__ __forceinline__ void edge_ref_point(bool quad, int k, double s, double& xi, double& eta) {
    if (quad) {
        switch (k) {
            case 0: xi = s; eta = -1.0; break;
            case 1: xi = 1.0; eta = s; break;
            case 2: xi = -s; eta = 1.0; break;
            default: xi = -1.0; eta = -s; break;
        }
    } else {
        const double t = 0.5 * (s + 1.0);
        switch (k) {
            case 0: xi = t; eta = 0.0; break;
            case 1: xi = 1.0 - t; eta = t; break;
            default: xi = 0.0; eta = 1.0 - t; break;
        }
    }
}

template <int NVC, bool QUAD>
__device__ __forceinline__ void p1_shape(double xi, double eta, double* M, double (*dM)[2]) {
    if (QUAD) {
        const double sx[4] = {-1, 1, 1, -1}, sy[4] = {-1, -1, 1, 1};
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            M[b] = 0.25 * (1 + sx[b] * xi) * (1 + sy[b] * eta);
            dM[b][0] = 0.25 * sx[b] * (1 + sy[b] * eta);
            dM[b][1] = 0.25 * sy[b] * (1 + sx[b] * xi);
        }
    } else {
        M[0] = 1 - xi - eta; M[1] = xi; M[2] = eta;
        dM[0][0] = -1; dM[0][1] = -1; dM[1][0] = 1; dM[1][1] = 0; dM[2][0] = 0; dM[2][1] = 1;
    }
}

template <int NPC, int NVC, bool QUAD, bool UHAT>
__global__ void k_penalty_setup(MeshDev m, FieldsDev f, FacetDev fd, double beta) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= fd.nf) return;
    const int e = fd.cell[i], k = fd.ledge[i];
    double X[NVC][3], Uh[NVC][3];
#pragma unroll
    for (int b = 0; b < NVC; ++b) {
        const int v = m.cells[b * m.nel + e];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            X[b][c] = m.xyz[3 * v + c];
            Uh[b][c] = UHAT ? f.uhat[3 * v + c] : 0.0;
        }
    }
    const int ka = k, kb = (k + 1) % NVC;
    double tv[3], len = 0.0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        tv[c] = X[kb][c] - X[ka][c];
        len += tv[c] * tv[c];
    }
    len = sqrt(len);
#pragma unroll
    for (int c = 0; c < 3; ++c) tv[c] /= len;
    const double gs[3] = {-0.7745966692414834, 0.0, 0.7745966692414834};
    const double gw[3] = {0.5555555555555556, 0.8888888888888888, 0.5555555555555556};
    double M2[9] = {0}, M1[4] = {0}, MR[9] = {0};
    for (int q = 0; q < 3; ++q) {
        const double s = gs[q];
        double nanson = 1.0;
        if (UHAT) {
            double xi, eta, M[NVC], dM[NVC][2];
            edge_ref_point(QUAD, k, s, xi, eta);
            p1_shape<NVC, QUAD>(xi, eta, M, dM);
            QPG g;
            qp_geometry<NVC, QUAD, true>(X, Uh, M, dM, g);
            // F again (qp_geometry keeps only Q and Ju): v = Ju F^-T N,  N = t x n
            double F[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
            for (int b = 0; b < NVC; ++b) {
                const double d0 = dM[b][0] * g.Q0[0][0] + dM[b][1] * g.Q0[1][0];
                const double d1 = dM[b][0] * g.Q0[0][1] + dM[b][1] * g.Q0[1][1];
                for (int jj = 0; jj < 3; ++jj) {
                    const double gm = d0 * g.E0[jj] + d1 * g.E1[jj];
                    for (int ii = 0; ii < 3; ++ii) F[ii][jj] += Uh[b][ii] * gm;
                }
            }
            double Nf[3];
            cross3(tv, g.E2, Nf);
            // Ju F^-T N = cof(F) N
            double C[3][3];
            C[0][0] = F[1][1] * F[2][2] - F[1][2] * F[2][1];
            C[0][1] = F[1][2] * F[2][0] - F[1][0] * F[2][2];
            C[0][2] = F[1][0] * F[2][1] - F[1][1] * F[2][0];
            C[1][0] = F[0][2] * F[2][1] - F[0][1] * F[2][2];
            C[1][1] = F[0][0] * F[2][2] - F[0][2] * F[2][0];
            C[1][2] = F[0][1] * F[2][0] - F[0][0] * F[2][1];
            C[2][0] = F[0][1] * F[1][2] - F[0][2] * F[1][1];
            C[2][1] = F[0][2] * F[1][0] - F[0][0] * F[1][2];
            C[2][2] = F[0][0] * F[1][1] - F[0][1] * F[1][0];
            double v[3];
            for (int ii = 0; ii < 3; ++ii) v[ii] = C[ii][0] * Nf[0] + C[ii][1] * Nf[1] + C[ii][2] * Nf[2];
            nanson = sqrt(dot3(v, v));
        }
        const double wq = gw[q] * 0.5 * len * nanson * beta / m.hK[e];
        const double L2[3] = {0.5 * s * (s - 1.0), 1.0 - s * s, 0.5 * s * (s + 1.0)};
        const double L1[2] = {0.5 * (1.0 - s), 0.5 * (1.0 + s)};
        // CG1CG1 (NPC == NVC): the displacement has no mid-edge node -- its edge block is the linear one, kept in the 3 x 3 slot
        // with an empty middle row and column (the host sets the "mid" node to the first vertex)
        const double U3[3] = {NPC == NVC ? L1[0] : L2[0], NPC == NVC ? 0.0 : L2[1], NPC == NVC ? L1[1] : L2[2]};
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) M2[3 * a + b] += wq * U3[a] * U3[b];
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b) M1[2 * a + b] += wq * L1[a] * L1[b];
        if (fd.MR) {
            double R[3];
            R[k % 3] = 1.0; R[(k + 1) % 3] = s; R[(k + 2) % 3] = -s;
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) MR[3 * a + b] += wq * R[a] * R[b];
        }
    }
    for (int a = 0; a < 9; ++a) fd.M2[9 * i + a] = M2[a];
    for (int a = 0; a < 4; ++a) fd.M1[4 * i + a] = M1[a];
    if (fd.MR)
        for (int a = 0; a < 9; ++a) fd.MR[9 * i + a] = MR[a];
}

// y += P x (mode 0) or diag += diag(P) (mode 1); *dotslot += x.Px
__global__ void k_penalty_apply(FacetDev fd, int ndof_u, int mode, const double* __restrict__ x, double* __restrict__ y,
                                double* dotslot) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double local = 0.0;
    if (i < fd.nf) {
        const int u0 = fd.unode[3 * i], u1 = fd.unode[3 * i + 1], u2 = fd.unode[3 * i + 2];
        const int v0 = fd.vnode[2 * i], v1 = fd.vnode[2 * i + 1];
        const double* A = fd.M2 + 9 * i;
        const double* B = fd.M1 + 4 * i;
        const int un[3] = {u0, u1, u2};
        const int vn[2] = {v0, v1};
        for (int c = 0; c < 3; ++c) {
            if (mode == 1) {
                for (int a = 0; a < 3; ++a) atomicAdd(&y[3 * un[a] + c], A[4 * a]);
                if (fd.MR) {
                    for (int a = 0; a < 3; ++a) atomicAdd(&y[ndof_u + 3 * fd.rnode[3 * i + a] + c], fd.MR[9 * i + 4 * a]);
                } else {
                    for (int a = 0; a < 2; ++a) atomicAdd(&y[ndof_u + 3 * vn[a] + c], B[3 * a]);
                }
            } else {
                double xu[3], xv[2];
                for (int a = 0; a < 3; ++a) xu[a] = x[3 * un[a] + c];
                for (int a = 0; a < 2; ++a) xv[a] = fd.MR ? 0.0 : x[ndof_u + 3 * vn[a] + c];
                for (int a = 0; a < 3; ++a) {
                    const double r = A[3 * a] * xu[0] + A[3 * a + 1] * xu[1] + A[3 * a + 2] * xu[2];
                    local += r * xu[a];
                    atomicAdd(&y[3 * un[a] + c], r);
                }
                if (fd.MR) {
                    const double* R = fd.MR + 9 * i;
                    double xr[3];
                    for (int a = 0; a < 3; ++a) xr[a] = x[ndof_u + 3 * fd.rnode[3 * i + a] + c];
                    for (int a = 0; a < 3; ++a) {
                        const double r = R[3 * a] * xr[0] + R[3 * a + 1] * xr[1] + R[3 * a + 2] * xr[2];
                        local += r * xr[a];
                        atomicAdd(&y[ndof_u + 3 * fd.rnode[3 * i + a] + c], r);
                    }
                } else
                for (int a = 0; a < 2; ++a) {
                    const double r = B[2 * a] * xv[0] + B[2 * a + 1] * xv[1];
                    local += r * xv[a];
                    atomicAdd(&y[ndof_u + 3 * vn[a] + c], r);
                }
            }
        }
    }
    if (dotslot) block_accumulate(local, dotslot);
}

// ------------------------------------------------------------------------------------------ vector kernels (PCG)
// scalar slots: [0,1] pAp, [2,3] rz, [4,5] rr, [6] bb, [7] spare
__global__ void k_fill(double* __restrict__ a, double v, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) a[i] = v;
}

__global__ void k_mask_zero(double* __restrict__ a, const unsigned char* __restrict__ mask, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (mask[i]) a[i] = 0.0;
}

// strong-BC rows of the operator: y_i = x_i
__global__ void k_mask_identity(double* __restrict__ y, const double* __restrict__ x, const unsigned char* __restrict__ mask,
                                int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (mask[i]) y[i] = x[i];
}

__global__ void k_invert_diag(double* __restrict__ d, const unsigned char* __restrict__ mask, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        d[i] = (mask && mask[i]) ? 1.0 : 1.0 / d[i];
}

// eq = dinv^(1/2) (mode 1) or the nearest power of two (mode 2) -- the symmetric diagonal scaling of option "equilibrate"
__global__ void k_eq_scale(double* __restrict__ eq, const double* __restrict__ dinv, const unsigned char* __restrict__ mask, int mode, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double s = (mask && mask[i]) ? 1.0 : sqrt(dinv[i]);
        if (mode == 2) s = ldexp(1.0, (int)nearbyint(log2(s)));
        eq[i] = s;
    }
}

__global__ void k_axpby(double* __restrict__ y, double a, const double* __restrict__ x, double b, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = a * x[i] + b * y[i];
}

// y *= d (Jacobi preconditioner as a vector operation)
__global__ void k_mul(double* __restrict__ y, const double* __restrict__ d, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] *= d[i];
}

// PCG with the multifrontal preconditioner: the two vector updates read their coefficients from device scalars
// (scal[0] = r.z of the previous iteration, [1] = r.z, [2] = p.Ap, [3] = r.r), so that an iteration needs ONE host
// synchronisation (the convergence test) instead of one per dot product
__global__ void k_pcgf_direction(double* __restrict__ p, const double* __restrict__ z, const double* __restrict__ scal, int first, int64_t n) {
    const double beta = first ? 0.0 : scal[1] / scal[0];
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        p[i] = first ? z[i] : z[i] + beta * p[i];
}
__global__ void k_pcgf_update(double* __restrict__ x, double* __restrict__ r, const double* __restrict__ p, const double* __restrict__ Ap,
                              double* __restrict__ scal, int64_t n) {
    const double alpha = scal[1] / scal[2];
    double local = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        x[i] += alpha * p[i];
        const double ri = r[i] - alpha * Ap[i];
        r[i] = ri;
        local += ri * ri;
    }
    block_accumulate(local, scal + 3);
    if (blockIdx.x == 0 && threadIdx.x == 0) scal[0] = scal[1];       // nobody reads slot 0 in this kernel
}

__global__ void k_dot(const double* __restrict__ a, const double* __restrict__ b, int64_t n, double* slot) {
    // four independent partial sums: a thread has 4 pairs of loads in flight per trip of the grid-stride loop
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        s0 += a[i] * b[i]; s1 += a[i + stride] * b[i + stride];
        s2 += a[i + 2 * stride] * b[i + 2 * stride]; s3 += a[i + 3 * stride] * b[i + 3 * stride];
    }
    for (; i < n; i += stride) s0 += a[i] * b[i];
    block_accumulate((s0 + s1) + (s2 + s3), slot);
}

// slot[0] += sum (a - b)^2, slot[1] += sum b^2: how far a field has moved from a snapshot (option "stale_factor")
__global__ void k_sq_change(const double* __restrict__ a, const double* __restrict__ b, int64_t n, double* slot) {
    double d = 0.0, r = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride) {
        const double x = a[i] - b[i];
        d += x * x; r += b[i] * b[i];
    }
    block_accumulate(d, slot);
    block_accumulate(r, slot + 1);
}

// ---- consistent mass matrix of the pressure space [CG1]^3, matrix-free: y += A x (diag != null: diag += diag(A) instead), one thread
// per cell, node-major xyz vectors of length 3 nn.  A = int Pv . w dx on the undeformed surface (rm_shell_pde.py:194-209): 3 x 3
// Gauss points on quadrilaterals, the 3-point rule on triangles (exact for the bilinear / linear basis on affine cells).
template <int NVC>
__global__ void __launch_bounds__(128)
k_vmass_apply(MeshDev m, const double* __restrict__ x, double* __restrict__ y, double* __restrict__ diag) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m.nel) return;
    int vid[NVC];
    double X[NVC][3], xe[NVC][3], ye[NVC][3], de[NVC];
#pragma unroll
    for (int b = 0; b < NVC; ++b) {
        vid[b] = m.cells[(size_t)b * m.nel + e];
        de[b] = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) { X[b][c] = m.xyz[3 * (size_t)vid[b] + c]; xe[b][c] = x ? x[3 * (size_t)vid[b] + c] : 0.0; ye[b][c] = 0.0; }
    }
    constexpr int NQ = NVC == 4 ? 9 : 3;
    for (int q = 0; q < NQ; ++q) {
        double N[NVC], dN[NVC][2], wq;
        if (NVC == 4) {
            const double g[3] = {-0.7745966692414834, 0.0, 0.7745966692414834}, gw[3] = {5.0 / 9.0, 8.0 / 9.0, 5.0 / 9.0};
            const double xi = g[q / 3], eta = g[q % 3];
            const double sx[4] = {-1, 1, 1, -1}, sy[4] = {-1, -1, 1, 1};
            wq = gw[q / 3] * gw[q % 3];
#pragma unroll
            for (int b = 0; b < NVC; ++b) {
                N[b] = 0.25 * (1 + sx[b] * xi) * (1 + sy[b] * eta);
                dN[b][0] = 0.25 * sx[b] * (1 + sy[b] * eta);
                dN[b][1] = 0.25 * sy[b] * (1 + sx[b] * xi);
            }
        } else {
            const double P[3][2] = {{1.0 / 6, 1.0 / 6}, {2.0 / 3, 1.0 / 6}, {1.0 / 6, 2.0 / 3}};
            wq = 1.0 / 6;
            N[0] = 1 - P[q][0] - P[q][1]; N[1] = P[q][0]; N[2] = P[q][1];
            dN[0][0] = -1; dN[0][1] = -1; dN[1][0] = 1; dN[1][1] = 0; dN[2][0] = 0; dN[2][1] = 1;
        }
        double J0[3] = {0, 0, 0}, J1[3] = {0, 0, 0}, nrm[3];
#pragma unroll
        for (int b = 0; b < NVC; ++b)
#pragma unroll
            for (int c = 0; c < 3; ++c) { J0[c] += X[b][c] * dN[b][0]; J1[c] += X[b][c] * dN[b][1]; }
        cross3(J0, J1, nrm);
        const double wd = wq * sqrt(dot3(nrm, nrm));
        double xq[3] = {0, 0, 0};
#pragma unroll
        for (int b = 0; b < NVC; ++b)
#pragma unroll
            for (int c = 0; c < 3; ++c) xq[c] += N[b] * xe[b][c];
#pragma unroll
        for (int b = 0; b < NVC; ++b) {
            de[b] += wd * N[b] * N[b];
#pragma unroll
            for (int c = 0; c < 3; ++c) ye[b][c] += wd * N[b] * xq[c];
        }
    }
#pragma unroll
    for (int b = 0; b < NVC; ++b)
#pragma unroll
        for (int c = 0; c < 3; ++c) atomicAdd(diag ? &diag[3 * (size_t)vid[b] + c] : &y[3 * (size_t)vid[b] + c], diag ? de[b] : ye[b][c]);
}

__global__ void k_div(double* __restrict__ z, const double* __restrict__ r, const double* __restrict__ d, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) z[i] = r[i] / d[i];
}

// ---- element-partitioned (multi-GPU) driver: the replicated separator entries of a vector are packed into one
// contiguous buffer for the all-reduce and written back; global dot products weigh every entry by 1 / (ranks holding it)
__global__ void k_gather_idx(double* __restrict__ out, const double* __restrict__ v, const int* __restrict__ idx, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = v[idx[i]];
}
__global__ void k_scatter_idx(double* __restrict__ v, const double* __restrict__ in, const int* __restrict__ idx, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[idx[i]] = in[i];
}
// out = v[idx] - save   (what the local forward sweep added to the replicated entries)
__global__ void k_top_delta(double* __restrict__ out, const double* __restrict__ v, const double* __restrict__ save,
                            const int* __restrict__ idx, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = v[idx[i]] - save[i];
}
// v[idx] = save + delta   (delta summed over the ranks)
__global__ void k_top_restore(double* __restrict__ v, const double* __restrict__ save, const double* __restrict__ delta,
                              const int* __restrict__ idx, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[idx[i]] = save[i] + delta[i];
}
// slot += sum_i w_i a_i b_i
__global__ void k_wdot(const double* __restrict__ a, const double* __restrict__ b, const double* __restrict__ w, int64_t n, double* slot) {
    double s0 = 0.0, s1 = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    for (; i + stride < n; i += 2 * stride) { s0 += w[i] * a[i] * b[i]; s1 += w[i + stride] * a[i + stride] * b[i + stride]; }
    for (; i < n; i += stride) s0 += w[i] * a[i] * b[i];
    block_accumulate(s0 + s1, slot);
}
// k_pcgf_update of the partitioned driver: r.r weighted as above (scal[3] then holds this rank's share of the global r.r)
__global__ void k_pcgf_update_w(double* __restrict__ x, double* __restrict__ r, const double* __restrict__ p, const double* __restrict__ Ap,
                                const double* __restrict__ w, double* __restrict__ scal, int64_t n) {
    const double alpha = scal[1] / scal[2];
    double local = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        x[i] += alpha * p[i];
        const double ri = r[i] - alpha * Ap[i];
        r[i] = ri;
        local += w[i] * ri * ri;
    }
    block_accumulate(local, scal + 3);
    if (blockIdx.x == 0 && threadIdx.x == 0) scal[0] = scal[1];
}
__global__ void k_copy_scalar(double* __restrict__ dst, const double* __restrict__ src) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *dst = *src;
}
// lower triangle (column by column) of a column-major n x n block <-> packed n (n + 1) / 2 doubles
__global__ void k_tril_pack(double* __restrict__ packed, const double* __restrict__ full, int n) {
    const int c = blockIdx.y;
    const long long base = (long long)c * n - (long long)c * (c - 1) / 2;           // entries of the columns before c
    for (int r = c + blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x)
        packed[base + (r - c)] = full[r + (size_t)n * c];
}
__global__ void k_tril_unpack(double* __restrict__ full, const double* __restrict__ packed, int n) {
    const int c = blockIdx.y;
    const long long base = (long long)c * n - (long long)c * (c - 1) / 2;
    for (int r = c + blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x)
        full[r + (size_t)n * c] = packed[base + (r - c)];
}

// ---- transient march (femo_newmark_*): the vector algebra of one step / one adjoint step, fused
// rhs = F (+ Fsw) + y1 + y2, masked rows zero
__global__ void k_newmark_rhs(double* __restrict__ b, const double* __restrict__ Fsw, const double* __restrict__ y1, const double* __restrict__ y2,
                              const unsigned char* __restrict__ mask, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double v = b[i] + y1[i] + y2[i];
        if (Fsw) v += Fsw[i];
        b[i] = (mask && mask[i]) ? 0.0 : v;
    }
}
// wdot <- bb (w_new - w_old) - wdot     (plate_sim.py:243-244, 333)
__global__ void k_newmark_wdot(double* __restrict__ wdot, const double* __restrict__ wn, const double* __restrict__ wo, double bb, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        wdot[i] = bb * (wn[i] - wo[i]) - wdot[i];
}
// adjoint step: mu_i = bb keep (M lam) - mu_next ; rhs = G_i + keep ((a M - K/2) lam) - bb mu_next ; b = rhs + bb mu_i (masked rows of b zero);
// out_rhs (level 0 only) = rhs
__global__ void k_newmark_adj(double* __restrict__ b, double* __restrict__ mu_i, double* __restrict__ rhs_out, const double* __restrict__ G,
                              const double* __restrict__ Mlam, const double* __restrict__ AKlam, const double* __restrict__ mu_next,
                              const unsigned char* __restrict__ mask, double bb, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const bool mk = mask && mask[i];
        const double keep = mk ? 0.0 : 1.0;
        const double mi = bb * Mlam[i] * keep - mu_next[i];
        const double rhs = G[i] + AKlam[i] * keep - bb * mu_next[i];
        mu_i[i] = mi;
        if (rhs_out) rhs_out[i] = rhs;
        b[i] = mk ? 0.0 : rhs + bb * mi;
    }
}
// out = mask ? src : 0   (the Dirichlet rows of a tangent right-hand side)
__global__ void k_mask_select(double* __restrict__ out, const double* __restrict__ src, const unsigned char* __restrict__ mask, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (mask && mask[i]) ? src[i] : 0.0;
}
// out = a x + c y + d z
__global__ void k_lincomb3(double* __restrict__ out, double a, const double* __restrict__ x, double c, const double* __restrict__ y, double d,
                           const double* __restrict__ z, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = a * x[i] + c * y[i] + (z ? d * z[i] : 0.0);
}

// r = b - Ap (or r = b when Ap == null); masked rows zero; z = dinv r; p = z; Ap = 0; rz[0] += r.z; rr[0] += r.r
__global__ void k_pcg_init(const double* __restrict__ b, double* __restrict__ Ap, const double* __restrict__ dinv,
                           const unsigned char* __restrict__ mask, double* __restrict__ r, double* __restrict__ z,
                           double* __restrict__ p, int64_t n, double* scal, int have_Ap) {
    double rz = 0.0, rr = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double ri = have_Ap ? b[i] - Ap[i] : b[i];
        if (mask && mask[i]) ri = 0.0;
        const double zi = dinv[i] * ri;
        r[i] = ri;
        z[i] = zi;
        p[i] = zi;
        Ap[i] = 0.0;
        rz += ri * zi;
        rr += ri * ri;
    }
    block_accumulate(rz, scal + 2);
    block_accumulate(rr, scal + 4);
}

// alpha = rz[s]/pAp[s]; x += alpha p; r -= alpha Ap; z = dinv r; rz[1-s] += r.z; rr[1-s] += r.r
__global__ void k_pcg_update(double* __restrict__ x, double* __restrict__ r, double* __restrict__ z,
                             const double* __restrict__ p, const double* __restrict__ Ap, const double* __restrict__ dinv,
                             const unsigned char* __restrict__ mask, int64_t n, double* scal, int s) {
    const double alpha = scal[2 + s] / scal[0 + s];
    double rz = 0.0, rr = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        x[i] += alpha * p[i];
        double ri = r[i] - alpha * Ap[i];
        if (mask && mask[i]) ri = 0.0;
        const double zi = dinv[i] * ri;
        r[i] = ri;
        z[i] = zi;
        rz += ri * zi;
        rr += ri * ri;
    }
    block_accumulate(rz, scal + 2 + (1 - s));
    block_accumulate(rr, scal + 4 + (1 - s));
}

// beta = rz[1-s]/rz[s]; p = z + beta p; Ap = 0; slot pAp[1-s] = 0
__global__ void k_pcg_direction(double* __restrict__ p, const double* __restrict__ z, double* __restrict__ Ap, int64_t n,
                                double* scal, int s) {
    const double beta = scal[2 + (1 - s)] / scal[2 + s];
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        p[i] = z[i] + beta * p[i];
        Ap[i] = 0.0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) scal[0 + (1 - s)] = 0.0;
}

}  // namespace femo
