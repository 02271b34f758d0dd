// The staged element-matrix column loop (stage_qpoints: lane q computes quadrature point q once, LDS) against the plain one
// (every lane computes every point) on a warped quadrilateral with random tables.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include "shell_device.h"
using namespace femo;
__global__ void k(const Tables* tab, const double* Xin, double* out) {
    constexpr int NPC = 9, NVC = 4, LD = 39;
    const int j = threadIdx.x;
    Elem<NPC, NVC> el;
    for (int b = 0; b < 4; ++b) { for (int c = 0; c < 3; ++c) { el.X[b][c] = Xin[3 * b + c]; el.Uh[b][c] = 0; } el.hn[b] = 0.02 + 0.001 * b; el.En[b] = 7e10; el.nun[b] = 0.3; }
    el.hK = 0.3;
    extern __shared__ double sq_raw[];
    QPoint<NPC, NVC>* sq = reinterpret_cast<QPoint<NPC, NVC>*>(sq_raw);
    stage_qpoints<NPC, NVC, true, false>(tab, el, 1.0, j, 64, sq);
    if (j >= LD) return;
    double y1[LD], y2[LD];
    for (int i = 0; i < LD; ++i) y1[i] = y2[i] = 0.0;
    const bool is_u = j < 3 * NPC;
    const int aj = is_u ? j / 3 : (j - 3 * NPC) / 3;
    const int cj = j - 3 * (is_u ? aj : NPC + aj);
    const int nq = tab->nq;
    for (int q = 0; q < nq; ++q) {
        {
            const QPoint<NPC, NVC>& p = sq[q];
            const double r0 = is_u ? tab->dN2[q][aj][0] : tab->dN1[q][aj][0], r1 = is_u ? tab->dN2[q][aj][1] : tab->dN1[q][aj][1];
            const double dk0 = r0 * p.g.Q[0][0] + r1 * p.g.Q[1][0], dk1 = r0 * p.g.Q[0][1] + r1 * p.g.Q[1][1];
            const double Mj = is_u ? 0.0 : tab->N1[q][aj];
            double G0[3], G1[3], th[3], T0[3], T1[3];
            for (int c = 0; c < 3; ++c) {
                const double ec = (c == cj) ? 1.0 : 0.0;
                G0[c] = is_u ? dk0 * ec : 0.0; G1[c] = is_u ? dk1 * ec : 0.0; th[c] = Mj * ec;
                T0[c] = is_u ? 0.0 : dk0 * ec; T1[c] = is_u ? 0.0 : dk1 * ec;
            }
            const Gen s = strains_reduced(p.g, G0, G1, th, T0, T1);
            const Gen t = stress_of(s, p.mat);
            strains_T<NPC, NVC>(p.g, p.d, p.mm, tab->N1[q], t, y2);
        }
        {
            QPG g;
            qp_geometry<NVC, true, false>(el.X, el.Uh, tab->N1[q], tab->dN1[q], g);
            double d[NPC][2], mm[NVC][2];
            local_derivs<NPC, NVC>(*tab, q, g.Q, d, mm);
            Mat mat, ex;
            material<DERIV_NONE>(interp<NVC>(tab->N1[q], el.hn), interp<NVC>(tab->N1[q], el.En), interp<NVC>(tab->N1[q], el.nun), el.hK,
                                 tab->wS[q] * g.det, tab->w[q] * g.det, g.Ju, mat, ex);
            const double r0 = is_u ? tab->dN2[q][aj][0] : tab->dN1[q][aj][0], r1 = is_u ? tab->dN2[q][aj][1] : tab->dN1[q][aj][1];
            const double dk0 = r0 * g.Q[0][0] + r1 * g.Q[1][0], dk1 = r0 * g.Q[0][1] + r1 * g.Q[1][1];
            const double Mj = is_u ? 0.0 : tab->N1[q][aj];
            double G0[3], G1[3], th[3], T0[3], T1[3];
            for (int c = 0; c < 3; ++c) {
                const double ec = (c == cj) ? 1.0 : 0.0;
                G0[c] = is_u ? dk0 * ec : 0.0; G1[c] = is_u ? dk1 * ec : 0.0; th[c] = Mj * ec;
                T0[c] = is_u ? 0.0 : dk0 * ec; T1[c] = is_u ? 0.0 : dk1 * ec;
            }
            const Gen s = strains_reduced(g, G0, G1, th, T0, T1);
            const Gen t = stress_of(s, mat);
            strains_T<NPC, NVC>(g, d, mm, tab->N1[q], t, y1);
        }
    }
    double e = 0, n = 0;
    for (int i = 0; i < LD; ++i) { e = fmax(e, fabs(y1[i] - y2[i])); n = fmax(n, fabs(y1[i])); }
    out[2 * j] = e; out[2 * j + 1] = n;
}
int main() {
    Tables T;
    T.nq = 16;
    srand(5);
    auto r = []() { return rand() / (double)RAND_MAX - 0.5; };
    for (int q = 0; q < MAXQ; ++q) {
        T.w[q] = T.wS[q] = 0.1 + 0.01 * q;
        for (int a = 0; a < 9; ++a) { T.N2[q][a] = r(); T.dN2[q][a][0] = r(); T.dN2[q][a][1] = r(); }
        const double x = 0.8 * r(), y = 0.8 * r();
        const double sx[4] = {-1, 1, 1, -1}, sy[4] = {-1, -1, 1, 1};
        for (int b = 0; b < 4; ++b) { T.N1[q][b] = 0.25 * (1 + sx[b] * x) * (1 + sy[b] * y); T.dN1[q][b][0] = 0.25 * sx[b] * (1 + sy[b] * y); T.dN1[q][b][1] = 0.25 * sy[b] * (1 + sx[b] * x); }
    }
    const double X[12] = {0, 0, 0, 1, 0.1, 0.2, 1.1, 0.9, -0.15, -0.05, 1, 0.3};
    Tables* dT; double *dX, *dout;
    (void)hipMalloc(&dT, sizeof T); (void)hipMalloc(&dX, sizeof X); (void)hipMalloc(&dout, 78 * 8);
    (void)hipMemcpy(dT, &T, sizeof T, hipMemcpyHostToDevice); (void)hipMemcpy(dX, X, sizeof X, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), T.nq * sizeof(QPoint<9, 4>), 0, dT, dX, dout);
    double h[78]; (void)hipMemcpy(h, dout, sizeof h, hipMemcpyDeviceToHost);
    double e = 0, n = 0;
    for (int j = 0; j < 39; ++j) { e = fmax(e, h[2 * j]); n = fmax(n, h[2 * j + 1]); if (h[2 * j] > 1e-12 * h[2 * j + 1]) printf("lane %d: err %g of %g\n", j, h[2 * j], h[2 * j + 1]); }
    printf("staged column loop against the plain one: max err %g, max entry %g\n", e, n);
    return !(e < 1e-12 * n);
}
