// micro-benchmark: the rank-k update kernels on in-situ shapes -- nfr fronts of nf rows, one update of everything behind the
// first KW columns (schur 2), 64 x 64 tiles (k_trailing_mfma) against 128 x 128 tiles (k_trailing_big), with and without the
// XCD super-tile map.     usage: tile_micro [nfr nf KW reps]
#include "../../femo_alpha_amd/csrc/frontal.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace femo;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <class T> T* up(const std::vector<T>& h) { T* d; CK(hipMalloc(&d, sizeof(T) * h.size())); CK(hipMemcpy(d, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice)); return d; }

int main(int argc, char** argv) {
    const int nfr = argc > 1 ? atoi(argv[1]) : 8, nf = argc > 2 ? atoi(argv[2]) : 3800, KW = argc > 3 ? atoi(argv[3]) : 512, reps = argc > 4 ? atoi(argv[4]) : 5;
    const int np = KW;                       // the pivot columns are the K panel; everything behind is the Schur complement
    const int nb = nf - np, ldp = ldp_of(nf);
    std::vector<int> h_nf(nfr, nf), h_np(nfr, np), lev(nfr);
    std::vector<long long> poff(nfr + 1), soff(nfr + 1);
    for (int i = 0; i <= nfr; ++i) { poff[i] = (long long)i * ldp * np; soff[i] = (long long)i * nb * nb; }
    for (int i = 0; i < nfr; ++i) lev[i] = i;
    FrontDev fd{};
    fd.ntree = nfr; fd.nf = up(h_nf); fd.npiv = up(h_np); fd.poff = up(poff); fd.soff = up(soff);
    double *P, *S;
    CK(hipMalloc(&P, sizeof(double) * poff[nfr])); CK(hipMalloc(&S, sizeof(double) * (soff[nfr] + 2)));
    CK(hipMemset(P, 0, sizeof(double) * poff[nfr])); CK(hipMemset(S, 0, sizeof(double) * (soff[nfr] + 2)));
    fd.P = P; fd.S = S;
    int* dlev = up(lev);
    const size_t shm = 4 * sizeof(double) * 16 * LSTRB;
    CK(hipFuncSetAttribute((const void*)k_trailing_big<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    const double flops = (double)nfr * ((double)nb * (nb + 1) / 2) * 2.0 * KW;
    const int ntr = (nb + (np & 1) + TS - 1) / TS, ntb = (ntr + 1) / 2;
    const int nst = (ntr + 3) / 4, nsup = nst * (nst + 1) / 2;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int var = 0; var < 3; ++var) {
        float best = 1e30f;
        for (int r = 0; r < reps; ++r) {
            CK(hipEventRecord(e0));
            if (var == 0) hipLaunchKernelGGL(k_trailing_mfma<false>, dim3(ntr * (ntr + 1) / 2, 1, nfr), dim3(256), 0, 0, fd, dlev, 0, 0, 2, 0, KW, (const unsigned char*)nullptr, 0, 0);
            else if (var == 1) hipLaunchKernelGGL(k_trailing_mfma<false>, dim3((nsup + 7) / 8 * 128, 1, nfr), dim3(256), 0, 0, fd, dlev, 0, 0, 2, 0, KW, (const unsigned char*)nullptr, 0, 1);
            else hipLaunchKernelGGL(k_trailing_big<false>, dim3(ntb * (ntb + 1) / 2, 1, nfr), dim3(256), shm, 0, fd, dlev, 0, 0, 2, 0, KW, (const unsigned char*)nullptr);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        printf("%d fronts nf %d K %d: %-28s %8.1f us  %6.1f TFLOP/s (%d workgroups)\n", nfr, nf, KW, var == 0 ? "64x64 tiles" : var == 1 ? "64x64, XCD super-tiles" : "128x128 tiles",
               best * 1e3, flops / (best * 1e-3) / 1e12, var == 2 ? ntb * (ntb + 1) / 2 * nfr : ntr * (ntr + 1) / 2 * nfr);
    }
    return 0;
}
