// Stress outputs of the RM shell: von Mises stress on the top surface, its p-norm aggregate and the
// element-wise (DG1) L2 projection, with the partial gradients the adjoint needs.
//
// Reference: ShellStressRM (femo_alpha/rm_shell/linear_shell_fenicsx/linear_shell_model.py:350-473):
//   u(xi2) = u_mid - xi2 (E2 x theta), eps = sym(T gradx(u(xi2)) T^T), sigma = C eps, vm = sqrt(s0^2 - s0 s1 + s1^2 + 3 s2^2);
// RMShellPDE.pnorm_stress (rm_shell/rm_shell_pde.py:112-128): 1/alpha int (m vm(h/2))^rho J dx with the degree-4
// measure of rm_shell_model.py:200-205 and alpha = reference area; von_Mises_stress(surface='Top') (:153-166)
// projected onto ('DG',1) (rm_shell_model.py:234-239, fea/utils_dolfinx.py:568-602).
// xi2 = h/2 is a *field*, so gradx(u(xi2)) contains a thickness-gradient term -1/2 (E2 x theta) (x) gradx(h).
#pragma once
#include "shell_device.h"

namespace femo {

struct TopStrain {
    double e0, e1, g;        // eps00, eps11, 2 eps01 on the top surface
    double k00, k11, k01;    // curvature part (for d/dh)
    double b0, b1;           // local components of E2 x theta
    double gh0, gh1;         // local gradx of the thickness
};

// zf: through-thickness coordinate as a fraction of the thickness, xi2 = zf h -- 1/2 top, 0 mid, -1/2 bottom surface
// (RMShellPDE.von_Mises_stress(surface=...), rm_shell_pde.py:153-165)
template <int NPC, int NVC>
__device__ __forceinline__ TopStrain top_strain(const Tables& t, int q, const QPG& g, const double* hn, bool ewm,
                                               const double* xe, double hq, double zf = 0.5) {
    const Gen s = strains_q<NPC, NVC>(t, q, g, xe);
    double th[3] = {0, 0, 0};
    double gh0 = 0.0, gh1 = 0.0;
    for (int b = 0; b < NVC; ++b) {
        const double Mb = t.NR[q][b];           // the rotation's shape function (N1 itself except for CG2CR1)
        for (int c = 0; c < 3; ++c) th[c] += Mb * xe[3 * NPC + 3 * b + c];
        if (!ewm) {
            const double r0 = t.dN1[q][b][0], r1 = t.dN1[q][b][1];
            gh0 += hn[b] * (r0 * g.Q[0][0] + r1 * g.Q[1][0]);
            gh1 += hn[b] * (r0 * g.Q[0][1] + r1 * g.Q[1][1]);
        }
    }
    TopStrain r;
    r.b0 = -dot3(th, g.E1);
    r.b1 = dot3(th, g.E0);
    r.gh0 = gh0; r.gh1 = gh1;
    r.k00 = s.k00; r.k11 = s.k11; r.k01 = s.k01;
    const double z = zf * hq;
    r.e0 = s.e00 - z * s.k00 - zf * r.b0 * gh0;
    r.e1 = s.e11 - z * s.k11 - zf * r.b1 * gh1;
    r.g = s.g01 - z * s.k01 - zf * (r.b0 * gh1 + r.b1 * gh0);
    return r;
}

__device__ __forceinline__ double von_mises(const TopStrain& e, double E, double nu, double* sig) {
    const double c = E / (1.0 - nu * nu);
    sig[0] = c * (e.e0 + nu * e.e1);
    sig[1] = c * (nu * e.e0 + e.e1);
    sig[2] = c * 0.5 * (1.0 - nu) * e.g;
    return sqrt(sig[0] * sig[0] - sig[0] * sig[1] + sig[1] * sig[1] + 3.0 * sig[2] * sig[2]);
}

// d vm / d (e0, e1, g)
__device__ __forceinline__ void dvm_deps(const double* sig, double vm, double E, double nu, double* d) {
    const double c = E / (1.0 - nu * nu);
    const double a0 = (2.0 * sig[0] - sig[1]) / (2.0 * vm), a1 = (2.0 * sig[1] - sig[0]) / (2.0 * vm), a2 = 3.0 * sig[2] / vm;
    d[0] = c * (a0 + nu * a1);
    d[1] = c * (nu * a0 + a1);
    d[2] = c * 0.5 * (1.0 - nu) * a2;
}

// mode 0: slots[0] += int (m vm)^rho J dx ; slots[1] += int J dx (area)
// mode 1: out_w  += d/dw   int (m vm)^rho J dx
// mode 2: out_f  += d/dh, mode 3: d/dE, mode 4: d/dnu   (field-space gradients)
template <int NPC, int NVC, bool QUAD, bool UHAT>
__global__ void __launch_bounds__(128)
k_pnorm(MeshDev m, FieldsDev f, const Tables* __restrict__ tab, int mode, double ms, double rho, double scale, double regc,
        const double* __restrict__ w, double* __restrict__ out, double* slots) {
    constexpr int LD = 3 * NPC + 3 * NVC;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0.0, area = 0.0;
    if (e < m.nel && cell_selected(m, e)) {
        Elem<NPC, NVC> el;
        load_elem<NPC, NVC, UHAT>(m, f, e, el);
        double xe[LD], ye[LD], ge[NVC];
        for (int a = 0; a < NPC; ++a)
            for (int c = 0; c < 3; ++c) xe[3 * a + c] = w[3 * el.pid[a] + c];
        for (int b = 0; b < NVC; ++b)
            for (int c = 0; c < 3; ++c) xe[3 * NPC + 3 * b + c] = w[m.ndof_u + 3 * rot_node(m, el, b) + c];
        for (int i = 0; i < LD; ++i) ye[i] = 0.0;
        for (int b = 0; b < NVC; ++b) ge[b] = 0.0;
        const int nq = tab->nq;
        for (int q = 0; q < nq; ++q) {
            QPG g;
            qp_geometry<NVC, QUAD, UHAT>(el.X, el.Uh, tab->N1[q], tab->dN1[q], g);
            const double wj = tab->w[q] * g.det * g.Ju;
            const double hq = interp<NVC>(tab->N1[q], el.hn), Eq = interp<NVC>(tab->N1[q], el.En),
                         nuq = interp<NVC>(tab->N1[q], el.nun);
            const TopStrain ts = top_strain<NPC, NVC>(*tab, q, g, el.hn, f.ewm != 0, xe, hq);
            double sig[3];
            const double vm = von_mises(ts, Eq, nuq, sig);
            const double p = pow(ms * vm, rho);
            // regc: coefficient of pnorm_stress(regularization=True), + regc int h^rho J dx with regc = 0.5e3 (rm_shell_pde.py:120-122)
            if (mode == 0) {
                acc += wj * p;
                if (regc != 0.0) acc += wj * regc * pow(hq, rho);
                area += tab->w[q] * g.det;          // alpha: area of the reference configuration (rm_shell_pde.py:124-127)
                continue;
            }
            if (mode == 2 && regc != 0.0) {
                const double dr = wj * regc * rho * pow(hq, rho - 1.0);
                for (int b = 0; b < NVC; ++b) {
                    ge[b] += dr * (f.ewm ? 1.0 : tab->N1[q][b]);
                    if (f.ewm) break;
                }
            }
            if (!(vm > 0.0)) continue;
            const double dp = wj * rho * p / vm;              // d/dvm of wj (m vm)^rho
            if (mode == 1) {
                double de[3];
                dvm_deps(sig, vm, Eq, nuq, de);
                Gen t;
                const double z = 0.5 * hq;
                t.e00 = dp * de[0]; t.e11 = dp * de[1]; t.g01 = dp * de[2];
                t.k00 = -z * t.e00; t.k11 = -z * t.e11; t.k01 = -z * t.g01;
                t.ga0 = t.ga1 = t.om = 0.0;
                strains_T_q<NPC, NVC>(*tab, q, g, t, ye);
                // -1/2 b (x) gradx(h): b0 = -theta.E1, b1 = theta.E0
                const double cb0 = -0.5 * (t.e00 * ts.gh0 + t.g01 * ts.gh1), cb1 = -0.5 * (t.e11 * ts.gh1 + t.g01 * ts.gh0);
                for (int b = 0; b < NVC; ++b)
                    for (int c = 0; c < 3; ++c)
                        ye[3 * NPC + 3 * b + c] += tab->N1[q][b] * (-cb0 * g.E1[c] + cb1 * g.E0[c]);
            } else if (mode == 2) {
                double de[3];
                dvm_deps(sig, vm, Eq, nuq, de);
                for (int b = 0; b < NVC; ++b) {
                    const double Mb = f.ewm ? 1.0 : tab->N1[q][b];
                    double d0 = -0.5 * Mb * ts.k00, d1 = -0.5 * Mb * ts.k11, d2 = -0.5 * Mb * ts.k01;
                    if (!f.ewm) {
                        const double r0 = tab->dN1[q][b][0], r1 = tab->dN1[q][b][1];
                        const double m0 = r0 * g.Q[0][0] + r1 * g.Q[1][0], m1 = r0 * g.Q[0][1] + r1 * g.Q[1][1];
                        d0 -= 0.5 * ts.b0 * m0;
                        d1 -= 0.5 * ts.b1 * m1;
                        d2 -= 0.5 * (ts.b0 * m1 + ts.b1 * m0);
                    }
                    ge[b] += dp * (de[0] * d0 + de[1] * d1 + de[2] * d2);
                    if (f.ewm) break;
                }
            } else if (mode == 3) {
                for (int b = 0; b < NVC; ++b) {
                    ge[b] += dp * vm / Eq * (f.ewm ? 1.0 : tab->N1[q][b]);
                    if (f.ewm) break;
                }
            } else {
                const double om = 1.0 - nuq * nuq, c = Eq / om, dc = 2.0 * nuq * Eq / (om * om);
                const double ds0 = dc * (ts.e0 + nuq * ts.e1) + c * ts.e1, ds1 = dc * (nuq * ts.e0 + ts.e1) + c * ts.e0;
                const double ds2 = dc * 0.5 * (1.0 - nuq) * ts.g - 0.5 * c * ts.g;
                const double dv = ((2.0 * sig[0] - sig[1]) * ds0 + (2.0 * sig[1] - sig[0]) * ds1 + 6.0 * sig[2] * ds2) / (2.0 * vm);
                for (int b = 0; b < NVC; ++b) {
                    ge[b] += dp * dv * (f.ewm ? 1.0 : tab->N1[q][b]);
                    if (f.ewm) break;
                }
            }
        }
        if (mode == 1) {
            for (int a = 0; a < NPC; ++a)
                for (int c = 0; c < 3; ++c) atomicAdd(&out[3 * el.pid[a] + c], scale * ye[3 * a + c]);
            for (int b = 0; b < NVC; ++b)
                for (int c = 0; c < 3; ++c) atomicAdd(&out[m.ndof_u + 3 * rot_node(m, el, b) + c], scale * ye[3 * NPC + 3 * b + c]);
        } else if (mode >= 2) {
            if (f.ewm) out[e] += scale * ge[0];
            else
                for (int b = 0; b < NVC; ++b) atomicAdd(&out[el.vid[b]], scale * ge[b]);
        }
    }
    if (mode == 0) {
        block_accumulate(acc, slots + 0);
        block_accumulate(area, slots + 1);
    }
}

// DG1 projection of the top-surface von Mises stress: per cell  M_e c = b_e,  M_e = int phi_i phi_j dx,
// b_e = int vm phi_i dx  (phi = the cell's P1/Q1 basis); out[NVC * e + i]
template <int NPC, int NVC, bool QUAD, bool UHAT>
__global__ void __launch_bounds__(128)
k_stress_field(MeshDev m, FieldsDev f, const Tables* __restrict__ tab, const double* __restrict__ w, double zf, double* __restrict__ out) {
    constexpr int LD = 3 * NPC + 3 * NVC;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m.nel) return;
    Elem<NPC, NVC> el;
    load_elem<NPC, NVC, UHAT>(m, f, e, el);
    double xe[LD];
    for (int a = 0; a < NPC; ++a)
        for (int c = 0; c < 3; ++c) xe[3 * a + c] = w[3 * el.pid[a] + c];
    for (int b = 0; b < NVC; ++b)
        for (int c = 0; c < 3; ++c) xe[3 * NPC + 3 * b + c] = w[m.ndof_u + 3 * rot_node(m, el, b) + c];
    double A[NVC][NVC + 1];
    for (int i = 0; i < NVC; ++i)
        for (int j = 0; j <= NVC; ++j) A[i][j] = 0.0;
    const int nq = tab->nq;
    for (int q = 0; q < nq; ++q) {
        QPG g;
        qp_geometry<NVC, QUAD, UHAT>(el.X, el.Uh, tab->N1[q], tab->dN1[q], g);
        const double wd = tab->w[q] * g.det;            // project() uses the plain dx (utils_dolfinx.py:582,589)
        const double hq = interp<NVC>(tab->N1[q], el.hn);
        const TopStrain ts = top_strain<NPC, NVC>(*tab, q, g, el.hn, f.ewm != 0, xe, hq, zf);
        double sig[3];
        const double vm = von_mises(ts, interp<NVC>(tab->N1[q], el.En), interp<NVC>(tab->N1[q], el.nun), sig);
        for (int i = 0; i < NVC; ++i) {
            for (int j = 0; j < NVC; ++j) A[i][j] += wd * tab->N1[q][i] * tab->N1[q][j];
            A[i][NVC] += wd * tab->N1[q][i] * vm;
        }
    }
    // Gaussian elimination of the NVC x NVC SPD system
    for (int k = 0; k < NVC; ++k) {
        const double ip = 1.0 / A[k][k];
        for (int i = k + 1; i < NVC; ++i) {
            const double fct = A[i][k] * ip;
            for (int j = k; j <= NVC; ++j) A[i][j] -= fct * A[k][j];
        }
    }
    double c[NVC];
    for (int i = NVC - 1; i >= 0; --i) {
        double s = A[i][NVC];
        for (int j = i + 1; j < NVC; ++j) s -= A[i][j] * c[j];
        c[i] = s / A[i][i];
    }
    for (int i = 0; i < NVC; ++i) out[NVC * e + i] = c[i];
}

// RMShellPDE.sum_stress_subdomain (rm_shell_pde.py:130-150): the six integrals  int sigma_ij J dx  over the selected
// sub-domain of the top-surface in-plane stress "in global coordinates" -- restated AS WRITTEN in
// ShellStressRM.inplaneStress (linear_shell_model.py:446-458): sigma_ij = sum_kl E012[i][k] s3d[k][l] E012[j][l] with
// E012[i][k] the k-th Cartesian component of the i-th local basis vector and s3d = [[s0, s2, 0], [s2, s1, 0], [0, 0, 0]],
// i.e. only the x and y components of the basis vectors enter.  slots[0..5] += (xx, yy, zz, xy, xz, yz).
template <int NPC, int NVC, bool QUAD, bool UHAT>
__global__ void __launch_bounds__(128)
k_stress_sums(MeshDev m, FieldsDev f, const Tables* __restrict__ tab, const double* __restrict__ w, double* slots) {
    constexpr int LD = 3 * NPC + 3 * NVC;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    if (e < m.nel && cell_selected(m, e)) {
        Elem<NPC, NVC> el;
        load_elem<NPC, NVC, UHAT>(m, f, e, el);
        double xe[LD];
        for (int a = 0; a < NPC; ++a)
            for (int c = 0; c < 3; ++c) xe[3 * a + c] = w[3 * el.pid[a] + c];
        for (int b = 0; b < NVC; ++b)
            for (int c = 0; c < 3; ++c) xe[3 * NPC + 3 * b + c] = w[m.ndof_u + 3 * rot_node(m, el, b) + c];
        const int nq = tab->nq;
        for (int q = 0; q < nq; ++q) {
            QPG g;
            qp_geometry<NVC, QUAD, UHAT>(el.X, el.Uh, tab->N1[q], tab->dN1[q], g);
            const double wj = tab->w[q] * g.det * g.Ju;
            const double hq = interp<NVC>(tab->N1[q], el.hn);
            const TopStrain ts = top_strain<NPC, NVC>(*tab, q, g, el.hn, f.ewm != 0, xe, hq);
            double sg[3];
            von_mises(ts, interp<NVC>(tab->N1[q], el.En), interp<NVC>(tab->N1[q], el.nun), sg);
            const double* Eb[3] = {g.E0, g.E1, g.E2};
            auto comp = [&](int i, int j) {
                return Eb[i][0] * (sg[0] * Eb[j][0] + sg[2] * Eb[j][1]) + Eb[i][1] * (sg[2] * Eb[j][0] + sg[1] * Eb[j][1]);
            };
            acc[0] += wj * comp(0, 0); acc[1] += wj * comp(1, 1); acc[2] += wj * comp(2, 2);
            acc[3] += wj * comp(0, 1); acc[4] += wj * comp(0, 2); acc[5] += wj * comp(1, 2);
        }
    }
    for (int k = 0; k < 6; ++k) block_accumulate(acc[k], slots + k);
}

}  // namespace femo
