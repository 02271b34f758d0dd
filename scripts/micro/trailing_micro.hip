// micro-benchmark: what bounds the trailing update?  Variants of k_trailing_mfma with parts switched off.
//   mode 0: full; 1: no C read-modify-write (store only one value per lane); 2: no L loads from global (LDS left as is)
//   usage: trailing_micro [nfronts nf kw]
#include "../../femo_alpha_amd/csrc/frontal.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace femo;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256)
k_var(double* Fall, int nf, int kc0, int kw) {
    double* F = Fall + (size_t)blockIdx.z * nf * nf;
    const int col_lo = kc0 + kw;
    const int cj = col_lo + blockIdx.y * TS;
    if (cj >= nf) return;
    const int ri = cj + blockIdx.x * TS;
    if (ri >= nf) return;
    constexpr int KC = 16;
    __shared__ double si[KC][LSTR];
    __shared__ double sj[KC][LSTR];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = (wv & 1) * 32, wc = (wv >> 1) * 32;
    const int l15 = lane & 15, l4 = lane >> 4;
    mfma_d4 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) acc[a][b] = (mfma_d4){0.0, 0.0, 0.0, 0.0};
    if (MODE == 2) { for (int idx = tid; idx < KC * LSTR; idx += 256) { (&si[0][0])[idx] = 1e-3 * idx; (&sj[0][0])[idx] = 1e-3; } }
    for (int k0 = 0; k0 < kw; k0 += KC) {
        __syncthreads();
        if (MODE != 2)
        for (int idx = tid; idx < KC * TS; idx += 256) {
            const int r = idx % TS, c = idx / TS;
            si[c][r] = (ri + r < nf) ? F[(ri + r) + (size_t)nf * (kc0 + k0 + c)] : 0.0;
            sj[c][r] = (cj + r < nf) ? F[(cj + r) + (size_t)nf * (kc0 + k0 + c)] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KC; kk += 4) {
            const double a0 = sj[kk + l4][wc + l15], a1 = sj[kk + l4][wc + 16 + l15];
            const double b0 = si[kk + l4][wr + l15], b1 = si[kk + l4][wr + 16 + l15];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    if (MODE == 1) {
        double s = 0;
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int reg = 0; reg < 4; ++reg) s += acc[a][b][reg];
        if (s == 1.2345) F[ri + (size_t)nf * cj] = s;
        return;
    }
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int reg = 0; reg < 4; ++reg) {
        const int cc = cj + wc + 16 * a + l4 + 4 * reg;
        const int r = ri + wr + 16 * b + l15;
        if (r < nf && cc < nf && r >= cc) F[r + (size_t)nf * cc] -= acc[a][b][reg];
    }
}

// pipelined variant: next stage's L values prefetched into registers during the MFMAs, C tile preloaded at the start,
// optional non-temporal C accesses
template <bool NT, bool PRELOAD, bool PIPE>
__global__ void __launch_bounds__(256)
k_pipe(double* Fall, int nf, int kc0, int kw) {
    double* F = Fall + (size_t)blockIdx.z * nf * nf;
    const int col_lo = kc0 + kw;
    const int cj = col_lo + blockIdx.y * TS;
    if (cj >= nf) return;
    const int ri = cj + blockIdx.x * TS;
    if (ri >= nf) return;
    constexpr int KC = 16;
    __shared__ double si[KC][LSTR];
    __shared__ double sj[KC][LSTR];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = (wv & 1) * 32, wc = (wv >> 1) * 32;
    const int l15 = lane & 15, l4 = lane >> 4;
    mfma_d4 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) acc[a][b] = (mfma_d4){0.0, 0.0, 0.0, 0.0};
    double cpre[2][2][4];
    if (PRELOAD) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int cc = cj + wc + 16 * a + l4 + 4 * reg;
                    const int r = ri + wr + 16 * b + l15;
                    const double* p = &F[r + (size_t)nf * cc];
                    cpre[a][b][reg] = (r < nf && cc < nf && r >= cc) ? (NT ? __builtin_nontemporal_load(p) : *p) : 0.0;
                }
    }
    // thread's slots of a stage: 4 rows-of-16-columns pieces for si and sj
    const int lr = tid % TS, lc = tid / TS;          // lc in 0..3; columns lc, lc+4, lc+8, lc+12
    double pi[4], pj[4];
    const bool iok = ri + lr < nf, jok = cj + lr < nf;
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = lc + 4 * q;
            pi[q] = (iok && k0 + c < kw) ? F[(ri + lr) + (size_t)nf * (kc0 + k0 + c)] : 0.0;
            pj[q] = (jok && k0 + c < kw) ? F[(cj + lr) + (size_t)nf * (kc0 + k0 + c)] : 0.0;
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < kw; k0 += KC) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) { si[lc + 4 * q][lr] = pi[q]; sj[lc + 4 * q][lr] = pj[q]; }
        __syncthreads();
        if (PIPE) { if (k0 + KC < kw) fetch(k0 + KC); }
#pragma unroll
        for (int kk = 0; kk < KC; kk += 4) {
            const double a0 = sj[kk + l4][wc + l15], a1 = sj[kk + l4][wc + 16 + l15];
            const double b0 = si[kk + l4][wr + l15], b1 = si[kk + l4][wr + 16 + l15];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (!PIPE) { if (k0 + KC < kw) fetch(k0 + KC); }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int cc = cj + wc + 16 * a + l4 + 4 * reg;
                const int r = ri + wr + 16 * b + l15;
                if (r < nf && cc < nf && r >= cc) {
                    double* p = &F[r + (size_t)nf * cc];
                    const double old = PRELOAD ? cpre[a][b][reg] : (NT ? __builtin_nontemporal_load(p) : *p);
                    if (NT) __builtin_nontemporal_store(old - acc[a][b][reg], p); else *p = old - acc[a][b][reg];
                }
            }
}

// 128x128 tile per workgroup, wave = 64x64 (4x4 MFMA blocks), register prefetch of the next stage
template <bool PIPE>
__global__ void __launch_bounds__(256, 2)
k_big(double* Fall, int nf, int kc0, int kw) {
    constexpr int TB = 128, KC = 16, LS = TB + 16;
    double* F = Fall + (size_t)blockIdx.z * nf * nf;
    const int col_lo = kc0 + kw;
    const int cj = col_lo + blockIdx.y * TB;
    if (cj >= nf) return;
    const int ri = cj + blockIdx.x * TB;
    if (ri >= nf) return;
    __shared__ double si[KC][LS];
    __shared__ double sj[KC][LS];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = (wv & 1) * 64, wc = (wv >> 1) * 64;
    const int l15 = lane & 15, l4 = lane >> 4;
    mfma_d4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (mfma_d4){0.0, 0.0, 0.0, 0.0};
    const int lr = tid % TB, lc = tid / TB;          // lc in 0..1; columns lc, lc+2, ...
    double pi[8], pj[8];
    const bool iok = ri + lr < nf, jok = cj + lr < nf;
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int c = lc + 2 * q;
            pi[q] = (iok && k0 + c < kw) ? F[(ri + lr) + (size_t)nf * (kc0 + k0 + c)] : 0.0;
            pj[q] = (jok && k0 + c < kw) ? F[(cj + lr) + (size_t)nf * (kc0 + k0 + c)] : 0.0;
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < kw; k0 += KC) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q) { si[lc + 2 * q][lr] = pi[q]; sj[lc + 2 * q][lr] = pj[q]; }
        __syncthreads();
        if (PIPE) { if (k0 + KC < kw) fetch(k0 + KC); }
#pragma unroll
        for (int kk = 0; kk < KC; kk += 4) {
            double av[4], bv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) { av[a] = sj[kk + l4][wc + 16 * a + l15]; bv[a] = si[kk + l4][wr + 16 * a + l15]; }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
        if (!PIPE) { if (k0 + KC < kw) fetch(k0 + KC); }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int cc = cj + wc + 16 * a + l4 + 4 * reg;
                const int r = ri + wr + 16 * b + l15;
                if (r < nf && cc < nf && r >= cc) F[r + (size_t)nf * cc] -= acc[a][b][reg];
            }
}

// 64x64 tile, C preloaded, LDS double-buffered: one barrier per 16-column stage
template <bool NT, int KC, bool PRE = true>
__global__ void __launch_bounds__(256)
k_db(double* Fall, int nf, int kc0, int kw) {
    double* F = Fall + (size_t)blockIdx.z * nf * nf;
    const int col_lo = kc0 + kw;
    const int cj = col_lo + blockIdx.y * TS;
    if (cj >= nf) return;
    const int ri = cj + blockIdx.x * TS;
    if (ri >= nf) return;
    __shared__ double si[2][KC][LSTR];
    __shared__ double sj[2][KC][LSTR];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = (wv & 1) * 32, wc = (wv >> 1) * 32;
    const int l15 = lane & 15, l4 = lane >> 4;
    mfma_d4 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) acc[a][b] = (mfma_d4){0.0, 0.0, 0.0, 0.0};
    constexpr int NQ = KC / 4;
    const int lr = tid % TS, lc = tid / TS;
    double pi[NQ], pj[NQ];
    const bool iok = ri + lr < nf, jok = cj + lr < nf;
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int c = lc + 4 * q;
            pi[q] = (iok && k0 + c < kw) ? F[(ri + lr) + (size_t)nf * (kc0 + k0 + c)] : 0.0;
            pj[q] = (jok && k0 + c < kw) ? F[(cj + lr) + (size_t)nf * (kc0 + k0 + c)] : 0.0;
        }
    };
    fetch(0);
    double cpre[2][2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int cc = cj + wc + 16 * a + l4 + 4 * reg;
                const int r = ri + wr + 16 * b + l15;
                const double* p = &F[r + (size_t)nf * cc];
                cpre[a][b][reg] = (PRE && r < nf && cc < nf && r >= cc) ? (NT ? __builtin_nontemporal_load(p) : *p) : 0.0;
            }
#pragma unroll
    for (int q = 0; q < NQ; ++q) { si[0][lc + 4 * q][lr] = pi[q]; sj[0][lc + 4 * q][lr] = pj[q]; }
    __syncthreads();
    int cur = 0;
    for (int k0 = 0; k0 < kw; k0 += KC) {
        const bool more = k0 + KC < kw;
        if (more) fetch(k0 + KC);
#pragma unroll
        for (int kk = 0; kk < KC; kk += 4) {
            const double a0 = sj[cur][kk + l4][wc + l15], a1 = sj[cur][kk + l4][wc + 16 + l15];
            const double b0 = si[cur][kk + l4][wr + l15], b1 = si[cur][kk + l4][wr + 16 + l15];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) { si[cur ^ 1][lc + 4 * q][lr] = pi[q]; sj[cur ^ 1][lc + 4 * q][lr] = pj[q]; }
        }
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int cc = cj + wc + 16 * a + l4 + 4 * reg;
                const int r = ri + wr + 16 * b + l15;
                if (r < nf && cc < nf && r >= cc) {
                    double* p = &F[r + (size_t)nf * cc];
                    const double old = PRE ? cpre[a][b][reg] : *p;
                    if (NT) __builtin_nontemporal_store(old - acc[a][b][reg], p); else *p = old - acc[a][b][reg];
                }
            }
}

// 128 x 64 tile per workgroup (wave = 64 rows x 32 columns: 4 x 2 MFMA blocks), LDS double-buffered
template <bool PRE>
__global__ void __launch_bounds__(256)
k_rect(double* Fall, int nf, int kc0, int kw) {
    constexpr int TR = 128, TC = 64, KC = 16, LSR = TR + 16, LSC = TC + 16;
    double* F = Fall + (size_t)blockIdx.z * nf * nf;
    const int col_lo = kc0 + kw;
    const int cj = col_lo + blockIdx.y * TC;
    if (cj >= nf) return;
    const int ri = cj + blockIdx.x * TR;
    if (ri >= nf) return;
    __shared__ double si[2][KC][LSR];
    __shared__ double sj[2][KC][LSC];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = (wv & 1) * 64, wc = (wv >> 1) * 32;
    const int l15 = lane & 15, l4 = lane >> 4;
    mfma_d4 acc[2][4];                                  // [column block][row block]
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 4; ++b) acc[a][b] = (mfma_d4){0.0, 0.0, 0.0, 0.0};
    // staging: rows tile 128 x 16 -> 8 per thread; cols tile 64 x 16 -> 4 per thread
    const int lr = tid % TR, lcr = tid / TR;            // lcr in 0..1: columns lcr, lcr+2, ...
    const int lc = tid % TC, lcc = tid / TC;            // lcc in 0..3
    double pi[8], pj[4];
    const bool iok = ri + lr < nf, jok = cj + lc < nf;
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) { const int c = lcr + 2 * q; pi[q] = (iok && k0 + c < kw) ? F[(ri + lr) + (size_t)nf * (kc0 + k0 + c)] : 0.0; }
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int c = lcc + 4 * q; pj[q] = (jok && k0 + c < kw) ? F[(cj + lc) + (size_t)nf * (kc0 + k0 + c)] : 0.0; }
    };
    fetch(0);
    double cpre[2][4][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int cc = cj + wc + 16 * a + l4 + 4 * reg;
                const int r = ri + wr + 16 * b + l15;
                cpre[a][b][reg] = (PRE && r < nf && cc < nf && r >= cc) ? F[r + (size_t)nf * cc] : 0.0;
            }
#pragma unroll
    for (int q = 0; q < 8; ++q) si[0][lcr + 2 * q][lr] = pi[q];
#pragma unroll
    for (int q = 0; q < 4; ++q) sj[0][lcc + 4 * q][lc] = pj[q];
    __syncthreads();
    int cur = 0;
    for (int k0 = 0; k0 < kw; k0 += KC) {
        const bool more = k0 + KC < kw;
        if (more) fetch(k0 + KC);
#pragma unroll
        for (int kk = 0; kk < KC; kk += 4) {
            double av[2], bv[4];
#pragma unroll
            for (int a = 0; a < 2; ++a) av[a] = sj[cur][kk + l4][wc + 16 * a + l15];
#pragma unroll
            for (int b = 0; b < 4; ++b) bv[b] = si[cur][kk + l4][wr + 16 * b + l15];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
        if (more) {
#pragma unroll
            for (int q = 0; q < 8; ++q) si[cur ^ 1][lcr + 2 * q][lr] = pi[q];
#pragma unroll
            for (int q = 0; q < 4; ++q) sj[cur ^ 1][lcc + 4 * q][lc] = pj[q];
        }
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int cc = cj + wc + 16 * a + l4 + 4 * reg;
                const int r = ri + wr + 16 * b + l15;
                if (r < nf && cc < nf && r >= cc) {
                    double* p = &F[r + (size_t)nf * cc];
                    *p = (PRE ? cpre[a][b][reg] : *p) - acc[a][b][reg];
                }
            }
}

// as k_db, but the global loads run two stages ahead of the MFMAs (second register set)
template <bool PRE>
__global__ void __launch_bounds__(256)
k_db2(double* Fall, int nf, int kc0, int kw) {
    constexpr int KC = 16, NQ = KC / 4;
    double* F = Fall + (size_t)blockIdx.z * nf * nf;
    const int col_lo = kc0 + kw;
    const int cj = col_lo + blockIdx.y * TS;
    if (cj >= nf) return;
    const int ri = cj + blockIdx.x * TS;
    if (ri >= nf) return;
    __shared__ double si[2][KC][LSTR];
    __shared__ double sj[2][KC][LSTR];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = (wv & 1) * 32, wc = (wv >> 1) * 32;
    const int l15 = lane & 15, l4 = lane >> 4;
    mfma_d4 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) acc[a][b] = (mfma_d4){0.0, 0.0, 0.0, 0.0};
    const int lr = tid % TS, lc = tid / TS;
    double pa[NQ], pb[NQ], qa[NQ], qb[NQ];
    const bool iok = ri + lr < nf, jok = cj + lr < nf;
#define FETCH(PI, PJ, K0) do { _Pragma("unroll") for (int q = 0; q < NQ; ++q) { const int c = lc + 4 * q; \
        PI[q] = (iok && (K0) + c < kw) ? F[(ri + lr) + (size_t)nf * (kc0 + (K0) + c)] : 0.0; \
        PJ[q] = (jok && (K0) + c < kw) ? F[(cj + lr) + (size_t)nf * (kc0 + (K0) + c)] : 0.0; } } while (0)
#define PUT(BUF, PI, PJ) do { _Pragma("unroll") for (int q = 0; q < NQ; ++q) { si[BUF][lc + 4 * q][lr] = PI[q]; sj[BUF][lc + 4 * q][lr] = PJ[q]; } } while (0)
#define COMPUTE(BUF) do { _Pragma("unroll") for (int kk = 0; kk < KC; kk += 4) { \
        const double a0 = sj[BUF][kk + l4][wc + l15], a1 = sj[BUF][kk + l4][wc + 16 + l15]; \
        const double b0 = si[BUF][kk + l4][wr + l15], b1 = si[BUF][kk + l4][wr + 16 + l15]; \
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0); \
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0); \
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0); \
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0); } } while (0)
    FETCH(pa, pb, 0);
    FETCH(qa, qb, KC);
    double cpre[2][2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int cc = cj + wc + 16 * a + l4 + 4 * reg;
                const int r = ri + wr + 16 * b + l15;
                cpre[a][b][reg] = (PRE && r < nf && cc < nf && r >= cc) ? F[r + (size_t)nf * cc] : 0.0;
            }
    PUT(0, pa, pb);
    __syncthreads();
    // stages come in pairs so that the two register sets alternate without copies
    for (int k0 = 0; k0 < kw; k0 += 2 * KC) {
        // buffer 0 holds stage k0; qa/qb hold stage k0+KC; fetch k0+2KC into pa/pb
        FETCH(pa, pb, k0 + 2 * KC);
        COMPUTE(0);
        PUT(1, qa, qb);
        __syncthreads();
        if (k0 + KC >= kw) break;
        FETCH(qa, qb, k0 + 3 * KC);
        COMPUTE(1);
        PUT(0, pa, pb);
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int cc = cj + wc + 16 * a + l4 + 4 * reg;
                const int r = ri + wr + 16 * b + l15;
                if (r < nf && cc < nf && r >= cc) {
                    double* p = &F[r + (size_t)nf * cc];
                    *p = (PRE ? cpre[a][b][reg] : *p) - acc[a][b][reg];
                }
            }
}

// k_db with 16-byte staging loads: a thread moves two consecutive rows of a factor column per load (needs an even leading
// dimension and even tile origins) -- half the global-load and half the LDS-store instructions per stage
typedef double d2 __attribute__((ext_vector_type(2)));
// VAR 1: later stages are loaded but not written to LDS; VAR 2: every workgroup loads the rows of tile (0, 0) (cache-hot);
// VAR 3: batched epilogue loads (no serial read-modify-write chain)
template <bool PRE, int VAR = 0>
__global__ void __launch_bounds__(256)
k_db16(double* Fall, int nf, int kc0, int kw) {
    constexpr int KC = 16;
    double* F = Fall + (size_t)blockIdx.z * nf * nf;
    const int col_lo = kc0 + kw;
    const int cj = col_lo + blockIdx.y * TS;
    if (cj >= nf) return;
    const int ri = cj + blockIdx.x * TS;
    if (ri >= nf) return;
    __shared__ __attribute__((aligned(16))) double si[2][KC][LSTR];
    __shared__ __attribute__((aligned(16))) double sj[2][KC][LSTR];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = (wv & 1) * 32, wc = (wv >> 1) * 32;
    const int l15 = lane & 15, l4 = lane >> 4;
    mfma_d4 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) acc[a][b] = (mfma_d4){0.0, 0.0, 0.0, 0.0};
    const int rp = tid & 31, cg = tid >> 5;                  // rows 2 rp, 2 rp + 1; columns cg, cg + 8
    d2 pi[2], pj[2];
    const bool iok = ri + 2 * rp + 1 < nf, jok = cj + 2 * rp + 1 < nf;      // (micro: nf even, tiles never cut a pair)
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = cg + 8 * q;
            const d2 z = {0.0, 0.0};
            pi[q] = (iok && k0 + c < kw) ? *reinterpret_cast<const d2*>(&F[((VAR == 2 ? col_lo : ri) + 2 * rp) + (size_t)nf * (kc0 + k0 + c)]) : z;
            pj[q] = (jok && k0 + c < kw) ? *reinterpret_cast<const d2*>(&F[((VAR == 2 ? col_lo : cj) + 2 * rp) + (size_t)nf * (kc0 + k0 + c)]) : z;
        }
    };
    double dummy = 0.0;
    fetch(0);
    double cpre[2][2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int cc = cj + wc + 16 * a + l4 + 4 * reg;
                const int r = ri + wr + 16 * b + l15;
                cpre[a][b][reg] = (PRE && r < nf && cc < nf && r >= cc) ? F[r + (size_t)nf * cc] : 0.0;
            }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        *reinterpret_cast<d2*>(&si[0][cg + 8 * q][2 * rp]) = pi[q];
        *reinterpret_cast<d2*>(&sj[0][cg + 8 * q][2 * rp]) = pj[q];
    }
    __syncthreads();
    int cur = 0;
    for (int k0 = 0; k0 < kw; k0 += KC) {
        const bool more = k0 + KC < kw;
        if (more) fetch(k0 + KC);
#pragma unroll
        for (int kk = 0; kk < KC; kk += 4) {
            const double a0 = sj[cur][kk + l4][wc + l15], a1 = sj[cur][kk + l4][wc + 16 + l15];
            const double b0 = si[cur][kk + l4][wr + l15], b1 = si[cur][kk + l4][wr + 16 + l15];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) {
            if (VAR == 1) {
#pragma unroll
                for (int q = 0; q < 2; ++q) dummy += pi[q].x + pi[q].y + pj[q].x + pj[q].y;
            } else {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                *reinterpret_cast<d2*>(&si[cur ^ 1][cg + 8 * q][2 * rp]) = pi[q];
                *reinterpret_cast<d2*>(&sj[cur ^ 1][cg + 8 * q][2 * rp]) = pj[q];
            }
            }
        }
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int cc = cj + wc + 16 * a + l4 + 4 * reg;
                const int r = ri + wr + 16 * b + l15;
                if (VAR == 3) continue;
                if (r < nf && cc < nf && r >= cc) {
                    double* p = &F[r + (size_t)nf * cc];
                    *p = (PRE ? cpre[a][b][reg] : *p) - acc[a][b][reg];
                }
            }
    if (VAR == 1 && dummy == 1.2345) F[0] = dummy;
    if (VAR == 3) {
        double cv[2][2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int cc = cj + wc + 16 * a + l4 + 4 * reg;
                    const int r = ri + wr + 16 * b + l15;
                    const bool ok = r < nf && cc < nf && r >= cc;
                    cv[a][b][reg] = F[ok ? r + (size_t)nf * cc : 0];          // unconditional load from a safe address
                }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int cc = cj + wc + 16 * a + l4 + 4 * reg;
                    const int r = ri + wr + 16 * b + l15;
                    if (r < nf && cc < nf && r >= cc) F[r + (size_t)nf * cc] = cv[a][b][reg] - acc[a][b][reg];
                }
    }
}

// scalar-base staging: the stage pointer is uniform (SGPR), every thread keeps four constant 32-bit byte offsets, so a
// stage costs four global_load_dwordx4 with saddr and no vector address arithmetic; rows beyond the front are clamped
// (their products are never stored), only the last partial K stage is masked; batched epilogue.
template <int TAILMODE>
__global__ void __launch_bounds__(256, 4)
k_sg(double* Fall, int nf, int kc0, int kw) {
    constexpr int KC = 16;
    double* F = Fall + (size_t)blockIdx.z * nf * nf;
    const int col_lo = kc0 + kw;
    const int cj = col_lo + blockIdx.y * TS;
    if (cj >= nf) return;
    const int ri = cj + blockIdx.x * TS;
    if (ri >= nf) return;
    __shared__ __attribute__((aligned(16))) double si[2][KC][LSTR];
    __shared__ __attribute__((aligned(16))) double sj[2][KC][LSTR];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = (wv & 1) * 32, wc = (wv >> 1) * 32;
    const int l15 = lane & 15, l4 = lane >> 4;
    mfma_d4 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) acc[a][b] = (mfma_d4){0.0, 0.0, 0.0, 0.0};
    const int rp = tid & 31, cg = tid >> 5;
    const int rowi = min(ri + 2 * rp, nf - 2), rowj = min(cj + 2 * rp, nf - 2);
    const unsigned obi0 = 8u * (unsigned)(rowi + nf * cg), obi1 = 8u * (unsigned)(rowi + nf * (cg + 8));
    const unsigned obj0 = 8u * (unsigned)(rowj + nf * cg), obj1 = 8u * (unsigned)(rowj + nf * (cg + 8));
    const char* base = reinterpret_cast<const char*>(F + (size_t)nf * kc0);
    const size_t stage_bytes = (size_t)nf * KC * 8;
    d2 pi[2], pj[2];
    auto fetch = [&](const char* b) {
        pi[0] = *reinterpret_cast<const d2*>(b + obi0); pi[1] = *reinterpret_cast<const d2*>(b + obi1);
        pj[0] = *reinterpret_cast<const d2*>(b + obj0); pj[1] = *reinterpret_cast<const d2*>(b + obj1);
    };
    auto fetch_tail = [&](const char* b, int left) {          // left = columns of this stage that exist (1..15)
        const d2 z = {0.0, 0.0};
        const unsigned c0 = min(cg, left - 1), c1 = min(cg + 8, left - 1);
        const unsigned a0 = 8u * (unsigned)(nf * c0), a1 = 8u * (unsigned)(nf * c1);
        const d2 vi0 = *reinterpret_cast<const d2*>(b + 8u * (unsigned)rowi + a0), vi1 = *reinterpret_cast<const d2*>(b + 8u * (unsigned)rowi + a1);
        const d2 vj0 = *reinterpret_cast<const d2*>(b + 8u * (unsigned)rowj + a0), vj1 = *reinterpret_cast<const d2*>(b + 8u * (unsigned)rowj + a1);
        pi[0] = cg < left ? vi0 : z; pi[1] = cg + 8 < left ? vi1 : z;
        pj[0] = cg < left ? vj0 : z; pj[1] = cg + 8 < left ? vj1 : z;
    };
    if (kw >= KC) fetch(base); else fetch_tail(base, kw);
    base += stage_bytes;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        *reinterpret_cast<d2*>(&si[0][cg + 8 * q][2 * rp]) = pi[q];
        *reinterpret_cast<d2*>(&sj[0][cg + 8 * q][2 * rp]) = pj[q];
    }
    __syncthreads();
    int cur = 0;
    for (int k0 = 0; k0 < kw; k0 += KC) {
        const int left = kw - k0 - KC;                  // columns behind this stage
        if (left > 0) {
            if (left >= KC) fetch(base); else fetch_tail(base, left);
            base += stage_bytes;
        }
#pragma unroll
        for (int kk = 0; kk < KC; kk += 4) {
            const double a0 = sj[cur][kk + l4][wc + l15], a1 = sj[cur][kk + l4][wc + 16 + l15];
            const double b0 = si[cur][kk + l4][wr + l15], b1 = si[cur][kk + l4][wr + 16 + l15];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (left > 0) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                *reinterpret_cast<d2*>(&si[cur ^ 1][cg + 8 * q][2 * rp]) = pi[q];
                *reinterpret_cast<d2*>(&sj[cur ^ 1][cg + 8 * q][2 * rp]) = pj[q];
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    double cv[2][2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int cc = cj + wc + 16 * a + l4 + 4 * reg;
                const int r = ri + wr + 16 * b + l15;
                const bool ok = r < nf && cc < nf && r >= cc;
                cv[a][b][reg] = F[ok ? r + (size_t)nf * cc : 0];
            }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int cc = cj + wc + 16 * a + l4 + 4 * reg;
                const int r = ri + wr + 16 * b + l15;
                if (r < nf && cc < nf && r >= cc) F[r + (size_t)nf * cc] = cv[a][b][reg] - acc[a][b][reg];
            }
}

int main(int argc, char** argv) {
    const int nfr = argc > 1 ? atoi(argv[1]) : 8, nf = argc > 2 ? atoi(argv[2]) : 3200, kw = argc > 3 ? atoi(argv[3]) : 128;
    double* F; CK(hipMalloc(&F, sizeof(double) * (size_t)nfr * nf * nf));
    CK(hipMemset(F, 0, sizeof(double) * (size_t)nfr * nf * nf));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nt = (nf - kw + TS - 1) / TS;
    const double flops = (double)nfr * nt * (nt + 1) / 2 * 64.0 * 64.0 * kw * 2.0;
    for (int mode = 0; mode < 26; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_var<0>, dim3(nt, nt, nfr), dim3(256), 0, 0, F, nf, 0, kw);
            if (mode == 1) hipLaunchKernelGGL(k_var<1>, dim3(nt, nt, nfr), dim3(256), 0, 0, F, nf, 0, kw);
            if (mode == 2) hipLaunchKernelGGL(k_var<2>, dim3(nt, nt, nfr), dim3(256), 0, 0, F, nf, 0, kw);
            if (mode == 3) hipLaunchKernelGGL((k_pipe<false, false, false>), dim3(nt, nt, nfr), dim3(256), 0, 0, F, nf, 0, kw);
            if (mode == 4) hipLaunchKernelGGL((k_pipe<true, false, false>), dim3(nt, nt, nfr), dim3(256), 0, 0, F, nf, 0, kw);
            if (mode == 5) hipLaunchKernelGGL((k_pipe<false, true, false>), dim3(nt, nt, nfr), dim3(256), 0, 0, F, nf, 0, kw);
            if (mode == 6) hipLaunchKernelGGL((k_pipe<false, true, true>), dim3(nt, nt, nfr), dim3(256), 0, 0, F, nf, 0, kw);
            if (mode == 7) hipLaunchKernelGGL((k_pipe<true, true, true>), dim3(nt, nt, nfr), dim3(256), 0, 0, F, nf, 0, kw);
            const int nb = (nf - kw + 127) / 128;
            if (mode == 8) hipLaunchKernelGGL((k_big<false>), dim3(nb, nb, nfr), dim3(256), 0, 0, F, nf, 0, kw);
            if (mode == 10) hipLaunchKernelGGL((k_db<false, 16>), dim3(nt, nt, nfr), dim3(256), 0, 0, F, nf, 0, kw);
            if (mode == 11) hipLaunchKernelGGL((k_db<true, 16>), dim3(nt, nt, nfr), dim3(256), 0, 0, F, nf, 0, kw);
            if (mode == 13) hipLaunchKernelGGL((k_db<false, 16, false>), dim3(nt, nt, nfr), dim3(256), 0, 0, F, nf, 0, kw);
            if (mode == 14) hipLaunchKernelGGL((k_db<false, 32, false>), dim3(nt, nt, nfr), dim3(256), 0, 0, F, nf, 0, kw);
            if (mode == 15) hipLaunchKernelGGL((k_db<false, 8, false>), dim3(nt, nt, nfr), dim3(256), 0, 0, F, nf, 0, kw);
            if (mode == 12) hipLaunchKernelGGL((k_db<false, 32>), dim3(nt, nt, nfr), dim3(256), 0, 0, F, nf, 0, kw);
            if (mode == 16) hipLaunchKernelGGL((k_rect<true>), dim3((nf - kw + 127) / 128, nt, nfr), dim3(256), 0, 0, F, nf, 0, kw);
            if (mode == 17) hipLaunchKernelGGL((k_rect<false>), dim3((nf - kw + 127) / 128, nt, nfr), dim3(256), 0, 0, F, nf, 0, kw);
            if (mode == 18) hipLaunchKernelGGL((k_db2<true>), dim3(nt, nt, nfr), dim3(256), 0, 0, F, nf, 0, kw);
            if (mode == 19) hipLaunchKernelGGL((k_db2<false>), dim3(nt, nt, nfr), dim3(256), 0, 0, F, nf, 0, kw);
            if (mode == 20) hipLaunchKernelGGL((k_db16<true>), dim3(nt, nt, nfr), dim3(256), 0, 0, F, nf, 0, kw);
            if (mode == 21) hipLaunchKernelGGL((k_db16<false>), dim3(nt, nt, nfr), dim3(256), 0, 0, F, nf, 0, kw);
            if (mode == 22) hipLaunchKernelGGL((k_db16<false, 1>), dim3(nt, nt, nfr), dim3(256), 0, 0, F, nf, 0, kw);
            if (mode == 23) hipLaunchKernelGGL((k_db16<false, 2>), dim3(nt, nt, nfr), dim3(256), 0, 0, F, nf, 0, kw);
            if (mode == 24) hipLaunchKernelGGL((k_db16<false, 3>), dim3(nt, nt, nfr), dim3(256), 0, 0, F, nf, 0, kw);
            if (mode == 25) hipLaunchKernelGGL((k_sg<0>), dim3(nt, nt, nfr), dim3(256), 0, 0, F, nf, 0, kw);
            if (mode == 9) hipLaunchKernelGGL((k_big<true>), dim3(nb, nb, nfr), dim3(256), 0, 0, F, nf, 0, kw);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("mode %d: %8.1f us  %6.2f TFLOP/s\n", mode, ms * 1e3, flops / (ms * 1e-3) / 1e12);
        }
    return 0;
}
