// micro-benchmark: launch durations of the diagonal-block / panel-rows / trailing kernels on a batch of synthetic
// fronts.   usage: panel_micro [nfronts nf npiv]
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
    const int variant = argc > 4 ? atoi(argv[4]) : 0;          // 0: k_diag_block, 1: k_diag_block2 (overlapped schedule, LDL)
    const int nb = nf - np, ldp = ldp_of(nf);
    std::vector<int> h_nf(nfr, nf), h_np(nfr, np), lev(nfr);
    std::vector<long long> poff(nfr + 1), soff(nfr), linvoff(nfr + 1);
    for (int i = 0; i <= nfr; ++i) { poff[i] = (long long)i * ldp * np; linvoff[i] = (long long)i * ((np + NB - 1) / NB) * NB * NB; }
    for (int i = 0; i < nfr; ++i) { soff[i] = (long long)i * nb * nb; lev[i] = i; }
    std::vector<double> A((size_t)nf * nf);
    srand(1);
    for (int c = 0; c < nf; ++c) for (int r = c; r < nf; ++r) A[r + (size_t)nf * c] = (r == c) ? nf + 1.0 : (rand() / (double)RAND_MAX - 0.5);
    std::vector<double> hP((size_t)ldp * np, 0.0), hS((size_t)std::max(nb, 1) * std::max(nb, 1), 0.0);
    for (int c = 0; c < nf; ++c)
        for (int r = c; r < nf; ++r) {
            if (c < np) hP[r + (size_t)ldp * c] = A[r + (size_t)nf * c];
            else hS[(r - np) + (size_t)nb * (c - np)] = A[r + (size_t)nf * c];
        }
    FrontDev fd{};
    fd.ntree = nfr; fd.nf = up(h_nf); fd.npiv = up(h_np); fd.poff = up(poff); fd.soff = up(soff); fd.linvoff = up(linvoff);
    double *P, *S, *Li;
    CK(hipMalloc(&P, sizeof(double) * poff[nfr])); CK(hipMalloc(&S, sizeof(double) * std::max<long long>(1, (long long)nfr * nb * nb)));
    CK(hipMalloc(&Li, sizeof(double) * linvoff[nfr]));
    fd.P = P; fd.S = S; fd.Linv = Li;
    double* Sw; CK(hipMalloc(&Sw, sizeof(double) * (size_t)nfr * SPD * SPD));
    CK(hipFuncSetAttribute((const void*)k_diag_block, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(diag_block_lds_blocks(4) * sizeof(blk32))));
    CK(hipFuncSetAttribute((const void*)k_diag_block2<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(diag_block2_lds_blocks(4) * sizeof(blk32))));
    int* dlev = up(lev); int* info; CK(hipMalloc(&info, 4)); CK(hipMemset(info, 0, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        for (int i = 0; i < nfr; ++i) {
            CK(hipMemcpy(P + poff[i], hP.data(), sizeof(double) * hP.size(), hipMemcpyHostToDevice));
            if (nb > 0) CK(hipMemcpy(S + soff[i], hS.data(), sizeof(double) * (size_t)nb * nb, hipMemcpyHostToDevice));
        }
        double tp = 0, tt = 0;
        for (int C0 = 0; C0 < np; C0 += NBO) {
            const int kw = std::min(NBO, np - C0);
            {
                CK(hipEventRecord(e0));
                if (variant == 0) hipLaunchKernelGGL(k_diag_block, dim3(nfr), dim3(256), diag_block_lds_blocks(4) * sizeof(blk32), 0, fd, dlev, 0, 4, C0, Sw, info, 0);
                else hipLaunchKernelGGL(k_diag_block2<false>, dim3(nfr), dim3(256), diag_block2_lds_blocks(4) * sizeof(blk32), 0, fd, dlev, 0, 4, C0, Sw, info, 0);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tp += ms;
                const int tiles = (nf - C0 - kw + TS - 1) / TS;
                float ms2 = 0;
                if (tiles > 0) {
                    CK(hipEventRecord(e0));
                    hipLaunchKernelGGL(k_panel_rows, dim3(tiles, nfr), dim3(256), 0, 0, fd, dlev, 0, C0, Sw, 0);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&ms2, e0, e1)); tp += ms2;
                }
                if (rep == 1) printf("C0=%4d: diag block %6.1f us, rows (%d tiles) %6.1f us\n", C0, ms * 1e3, tiles, ms2 * 1e3);
                if (rep == 1 && C0 == 0) {
                    long long st[32]; CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st)));
                    printf("   (us, 100 MHz clock) load+store-to-LDS %.2f sync %.2f |", (st[0] - st[31]) * 0.01, (st[1] - st[0]) * 0.01);
                    if (variant == 0)
                    for (int j = 0; j < 4; ++j) printf(" step%d: chol %.2f trsm %.2f upd %.2f |", j, (st[3 + 4 * j] - st[2 + 4 * j]) * 0.01,
                                                       j < 3 ? (st[4 + 4 * j] - st[3 + 4 * j]) * 0.01 : 0.0, j < 3 ? (st[2 + 4 * (j + 1)] - st[4 + 4 * j]) * 0.01 : 0.0);
                    else
                    for (int j = 0; j < 4; ++j) printf(" stage%d: factor || shadow %.2f, between %.2f |", j, (st[3 + 4 * j] - st[2 + 4 * j]) * 0.01,
                                                       j < 3 ? (st[4 + 4 * j] - st[3 + 4 * j]) * 0.01 : 0.0);
                    printf(" S-phase %.2f | whole kernel %.2f\n", (st[21] - st[20]) * 0.01, (st[21] - st[31]) * 0.01);
                }
            }
            const int nt = (nf - C0 - kw + 1 + TS - 1) / TS;
            if (nt > 0) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_trailing_mfma<false>, dim3(nt * (nt + 1) / 2, 1, nfr), dim3(256), 0, 0, fd, dlev, 0, C0, 2, 0, NBO, (const unsigned char*)nullptr, 0, 0);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tt += ms;
                if (rep == 1) printf("  trailing C0=%4d nt=%3d: %7.1f us\n", C0, nt, ms * 1e3);
            }
        }
        printf("rep %d: panels %.2f ms, trailing %.2f ms\n", rep, tp, tt);
    }
    {   // checksums of everything the diagonal-block kernel writes (compare the variants)
        std::vector<double> p((size_t)ldp * np), li(linvoff[1]), sw((size_t)SPD * SPD);
        CK(hipMemcpy(p.data(), P, sizeof(double) * p.size(), hipMemcpyDeviceToHost));
        CK(hipMemcpy(li.data(), Li, sizeof(double) * li.size(), hipMemcpyDeviceToHost));
        CK(hipMemcpy(sw.data(), Sw, sizeof(double) * sw.size(), hipMemcpyDeviceToHost));
        auto cs = [](const std::vector<double>& v) { double a = 0, b = 0; for (size_t i = 0; i < v.size(); ++i) { if (v[i] == v[i]) { a += v[i] * (1 + (i % 7)); b += v[i] * v[i]; } } return std::make_pair(a, b); };
        const int kwl = np - (np - 1) / NBO * NBO;
        double ss = 0, s2 = 0;
        for (int c = 0; c < kwl; ++c) for (int r = c; r < kwl; ++r) { const double x = sw[r + (size_t)SPD * c]; ss += x * (1 + (r % 5)); s2 += x * x; }
        printf("checksums: P %.15e %.15e | Linv %.15e %.15e | S(last panel, lower) %.15e %.15e\n", cs(p).first, cs(p).second, cs(li).first, cs(li).second, ss, s2);
    }
    int hinfo; CK(hipMemcpy(&hinfo, info, 4, hipMemcpyDeviceToHost));
    printf("bad pivots %d\n", hinfo);
    return 0;
}
