// micro-benchmark: phases of k_panel (wall_clock64 stamps of workgroup (0,0)) and launch durations of the
// panel / trailing kernels on a batch of synthetic fronts.   usage: panel_micro [nfronts nf npiv]
#define FEMO_PANEL_STAMPS
#include "../../femo_alpha_amd/csrc/frontal.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace femo;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <class T> T* up(const std::vector<T>& h) {
    T* d; CK(hipMalloc(&d, sizeof(T) * h.size())); CK(hipMemcpy(d, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice)); return d;
}

int main(int argc, char** argv) {
    const int nfr = argc > 1 ? atoi(argv[1]) : 8, nf = argc > 2 ? atoi(argv[2]) : 3200, np = argc > 3 ? atoi(argv[3]) : 1024;
    std::vector<int> h_nf(nfr, nf), h_np(nfr, np), lev(nfr);
    std::vector<long long> foff(nfr + 1), linvoff(nfr + 1);
    for (int i = 0; i <= nfr; ++i) { foff[i] = (long long)i * nf * nf; linvoff[i] = (long long)i * ((np + NB - 1) / NB) * NB * NB; }
    for (int i = 0; i < nfr; ++i) lev[i] = i;
    std::vector<double> A((size_t)nf * nf);
    srand(1);
    for (int c = 0; c < nf; ++c) for (int r = c; r < nf; ++r) A[r + (size_t)nf * c] = (r == c) ? nf + 1.0 : (rand() / (double)RAND_MAX - 0.5);
    FrontDev fd{};
    fd.ntree = nfr; fd.nf = up(h_nf); fd.npiv = up(h_np); fd.foff = up(foff); fd.linvoff = up(linvoff);
    double* F; CK(hipMalloc(&F, sizeof(double) * foff[nfr]));
    double* Li; CK(hipMalloc(&Li, sizeof(double) * linvoff[nfr]));
    fd.F = F; fd.Linv = Li;
    int* dlev = up(lev); int* info; CK(hipMalloc(&info, 4)); CK(hipMemset(info, 0, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        for (int i = 0; i < nfr; ++i) CK(hipMemcpy(F + foff[i], A.data(), sizeof(double) * nf * nf, hipMemcpyHostToDevice));
        double tp = 0, tt = 0;
        for (int C0 = 0; C0 < np; C0 += NBO) {
            const int kw = std::min(NBO, np - C0);
            for (int c0 = C0; c0 < C0 + kw; c0 += NB) {
                const int tiles = std::max(1, (nf - c0 - NB + TS - 1) / TS);
                const int gx = std::max(1, std::min(tiles, 1024 / nfr));
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_panel, dim3(gx, nfr), dim3(256), 0, 0, fd, dlev, C0, c0, info);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tp += ms;
                if (rep == 1 && (C0 == 0 || C0 + NBO >= np)) {
                    long long st[16]; CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st)));
                    printf("panel c0=%4d kprev=%3d gx=%3d: %6.1f us | load %5.2f diag-upd %5.2f chol %5.2f", c0, c0 - C0, gx, ms * 1e3,
                           (st[1] - st[0]) * 0.01, (st[2] - st[1]) * 0.01, (st[3] - st[2]) * 0.01);
                    long long prev = st[3];
                    for (int m = 0; m < (c0 - C0) / NB; ++m) { printf(" chunk%d %5.2f", m, (st[4 + m] - prev) * 0.01); prev = st[4 + m]; }
                    printf(" ->trsm-start %5.2f trsm %5.2f (last tile)\n", (st[8] - prev) * 0.01, (st[9] - st[8]) * 0.01);
                }
            }
            const int nt = (nf - C0 - kw + TS - 1) / TS;
            if (nt > 0) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_trailing_mfma, dim3(nt, nt, nfr), dim3(256), 0, 0, fd, dlev, C0, NBO);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tt += ms;
                if (rep == 1) printf("  trailing C0=%4d nt=%3d: %7.1f us\n", C0, nt, ms * 1e3);
            }
        }
        printf("rep %d: panels %.2f ms, trailing %.2f ms\n", rep, tp, tt);
    }
    int hinfo; CK(hipMemcpy(&hinfo, info, 4, hipMemcpyDeviceToHost));
    printf("bad pivots %d\n", hinfo);
    return 0;
}
