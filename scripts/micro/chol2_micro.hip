// micro-benchmark: the 32 x 32 factor + inverse of k_diag_block, one wave -- register elimination over 32 columns
// (chol32_inverse) against two eliminations over 16 columns with MFMA glue (chol32_inverse_v2).  Prints the time per
// block and the largest difference of the two inverses.
#include "../../femo_alpha_amd/csrc/frontal.h"
#include <cstdio>
#include <cmath>
#include <vector>
using namespace femo;

__global__ void __launch_bounds__(64) k_v1(const double* A, double* X, int reps, int wb) {
    const int lane = threadIdx.x;
    __shared__ blk32 D;
    for (int it = 0; it < reps; ++it) {
        for (int idx = lane; idx < NB * NB; idx += 64) D[idx / NB][idx % NB] = A[idx];
        __builtin_amdgcn_wave_barrier();
        double a[NB];
#pragma unroll
        for (int c = 0; c < NB; ++c) a[c] = (lane < NB) ? D[lane][c] : (c == lane - NB ? 1.0 : 0.0);
        chol32_inverse(a, wb, lane);
        if (lane >= NB) {
            const int cl = lane - NB;
#pragma unroll
            for (int r = 0; r < NB; ++r) D[r][cl] = (cl < wb && r < wb && cl <= r) ? a[r] : 0.0;
        }
        __builtin_amdgcn_wave_barrier();
        for (int idx = lane; idx < NB * NB; idx += 64) X[idx] = D[idx / NB][idx % NB];
    }
}

__global__ void __launch_bounds__(64) k_v2(const double* A, double* X, int reps, int wb) {
    const int lane = threadIdx.x;
    __shared__ blk32 D;
    for (int it = 0; it < reps; ++it) {
        for (int idx = lane; idx < NB * NB; idx += 64) D[idx / NB][idx % NB] = A[idx];
        __builtin_amdgcn_wave_barrier();
        chol32_inverse_v2<false, false>(D, wb, lane);
        for (int idx = lane; idx < NB * NB; idx += 64) X[idx] = D[idx / NB][idx % NB];
    }
}

__global__ void __launch_bounds__(64) k_v3(const double* A, double* X, int reps, int wb) {
    const int lane = threadIdx.x;
    __shared__ blk32 D;
    for (int it = 0; it < reps; ++it) {
        for (int idx = lane; idx < NB * NB; idx += 64) D[idx / NB][idx % NB] = A[idx];
        __builtin_amdgcn_wave_barrier();
        double a[NB];
#pragma unroll
        for (int c = 0; c < NB; ++c) a[c] = (lane < NB) ? D[lane][c] : (c == lane - NB ? 1.0 : 0.0);
        double rs;
        ldl32_inverse<false>(a, wb, lane, &rs);
        // the row scaling reads lanes < 32: broadcast OUTSIDE the divergent store (inside it the compiler may fold rs to the
        // value the active lanes hold)
#pragma unroll
        for (int r = 0; r < NB; ++r) a[r] *= rl(rs, r);
        if (lane >= NB) {
            const int cl = lane - NB;
#pragma unroll
            for (int r = 0; r < NB; ++r) D[r][cl] = (cl < wb && r < wb && cl <= r) ? a[r] : 0.0;
        }
        __builtin_amdgcn_wave_barrier();
        for (int idx = lane; idx < NB * NB; idx += 64) X[idx] = D[idx / NB][idx % NB];
    }
}

__global__ void __launch_bounds__(64) k_v4(const double* A, double* X, int reps, int wb) {
    const int lane = threadIdx.x;
    __shared__ blk32 D;
    __shared__ double bc[64];
    for (int it = 0; it < reps; ++it) {
        for (int idx = lane; idx < NB * NB; idx += 64) D[idx / NB][idx % NB] = A[idx];
        __builtin_amdgcn_wave_barrier();
        double a[NB];
#pragma unroll
        for (int c = 0; c < NB; ++c) a[c] = (lane < NB) ? D[lane][c] : (c == lane - NB ? 1.0 : 0.0);
        double rs;
        ldl32_inverse_lds<false>(a, wb, lane, &rs, bc);
#pragma unroll
        for (int r = 0; r < NB; ++r) a[r] *= rl(rs, r);
        if (lane >= NB) {
            const int cl = lane - NB;
#pragma unroll
            for (int r = 0; r < NB; ++r) D[r][cl] = (cl < wb && r < wb && cl <= r) ? a[r] : 0.0;
        }
        __builtin_amdgcn_wave_barrier();
        for (int idx = lane; idx < NB * NB; idx += 64) X[idx] = D[idx / NB][idx % NB];
    }
}

__global__ void __launch_bounds__(64) k_v5(const double* A, double* X, int reps, int wb) {
    const int lane = threadIdx.x;
    __shared__ blk32 D;
    for (int it = 0; it < reps; ++it) {
        for (int idx = lane; idx < NB * NB; idx += 64) D[idx / NB][idx % NB] = A[idx];
        __builtin_amdgcn_wave_barrier();
        chol32_inverse_v2<true, false>(D, wb, lane);
        for (int idx = lane; idx < NB * NB; idx += 64) X[idx] = D[idx / NB][idx % NB];
    }
}

__global__ void __launch_bounds__(64) k_v0(const double* A, double* X, int reps, int wb) {     // load / store only
    const int lane = threadIdx.x;
    __shared__ blk32 D;
    for (int it = 0; it < reps; ++it) {
        for (int idx = lane; idx < NB * NB; idx += 64) D[idx / NB][idx % NB] = A[idx];
        __builtin_amdgcn_wave_barrier();
        for (int idx = lane; idx < NB * NB; idx += 64) X[idx] = D[idx / NB][idx % NB] + it;
    }
}

int main() {
    for (int wb : {32, 20, 9}) {
        std::vector<double> h(NB * NB, 0.0);
        srand(3);
        for (int i = 0; i < NB; ++i)
            for (int j = 0; j <= i; ++j) h[i * NB + j] = (i < wb && j < wb) ? ((i == j) ? 40.0 + i : (rand() / (double)RAND_MAX - 0.5)) : (i == j ? 1.0 : 0.0);
        std::vector<double> x3(NB * NB);
        double *A, *X1, *X2;
        hipMalloc(&A, sizeof(double) * NB * NB); hipMalloc(&X1, sizeof(double) * NB * NB); hipMalloc(&X2, sizeof(double) * NB * NB);
        hipMemcpy(A, h.data(), sizeof(double) * NB * NB, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int reps = 200;
        float t[6] = {0, 0, 0, 0, 0, 0};
        std::vector<double> x5(NB * NB);
        double* X3; hipMalloc(&X3, sizeof(double) * NB * NB);
        std::vector<double> x4(NB * NB);
        for (int v = 0; v < 6; ++v)
            for (int w = 0; w < 2; ++w) {
                hipEventRecord(e0);
                if (v == 0) hipLaunchKernelGGL(k_v1, dim3(1), dim3(64), 0, 0, A, X1, reps, wb);
                else if (v == 1) hipLaunchKernelGGL(k_v2, dim3(1), dim3(64), 0, 0, A, X2, reps, wb);
                else if (v == 2) hipLaunchKernelGGL(k_v3, dim3(1), dim3(64), 0, 0, A, X3, reps, wb);
                else if (v == 3) hipLaunchKernelGGL(k_v0, dim3(1), dim3(64), 0, 0, A, X3, reps, wb);
                else if (v == 4) hipLaunchKernelGGL(k_v4, dim3(1), dim3(64), 0, 0, A, X3, reps, wb);
                else hipLaunchKernelGGL(k_v5, dim3(1), dim3(64), 0, 0, A, X3, reps, wb);
                if (v == 5 && w == 1) { hipDeviceSynchronize(); hipMemcpy(x5.data(), X3, sizeof(double) * NB * NB, hipMemcpyDeviceToHost); }
                if (v == 4 && w == 1) { hipDeviceSynchronize(); hipMemcpy(x4.data(), X3, sizeof(double) * NB * NB, hipMemcpyDeviceToHost); }
                if (v == 2 && w == 1) { hipDeviceSynchronize(); hipMemcpy(x3.data(), X3, sizeof(double) * NB * NB, hipMemcpyDeviceToHost); }
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&t[v], e0, e1);
            }
        std::vector<double> x1(NB * NB), x2(NB * NB);
        hipMemcpy(x1.data(), X1, sizeof(double) * NB * NB, hipMemcpyDeviceToHost);
        hipMemcpy(x2.data(), X2, sizeof(double) * NB * NB, hipMemcpyDeviceToHost);
        double dmax = 0, xmax = 0;
        for (int i = 0; i < NB * NB; ++i) { dmax = fmax(dmax, fabs(x1[i] - x2[i])); xmax = fmax(xmax, fabs(x1[i])); }
        // check X against the matrix: (X^T X) A = I on the leading wb x wb block
        double emax = 0;
        for (int i = 0; i < wb; ++i)
            for (int j = 0; j < wb; ++j) {
                double s = 0;
                for (int k = 0; k < wb; ++k) {
                    double m = 0;                       // (X^T X)[i][k] = sum_r X[r][i] X[r][k]
                    for (int r = 0; r < wb; ++r) m += x2[r * NB + i] * x2[r * NB + k];
                    const double akj = k >= j ? h[k * NB + j] : h[j * NB + k];
                    s += m * akj;
                }
                emax = fmax(emax, fabs(s - (i == j ? 1.0 : 0.0)));
            }
        double d3 = 0;
        for (int i = 0; i < NB * NB; ++i) d3 = fmax(d3, fabs(x1[i] - x3[i]));
        printf("   X1[0][0] %.6f X3[0][0] %.6f | X1[1][0] %.6f X3[1][0] %.6f | X1[1][1] %.6f X3[1][1] %.6f | X1[31][5] %.6e X3[31][5] %.6e\n", x1[0], x3[0], x1[NB], x3[NB], x1[NB + 1], x3[NB + 1], x1[31 * NB + 5], x3[31 * NB + 5]);
        double d5 = 0;
        for (int i = 0; i < NB * NB; ++i) d5 = fmax(d5, fabs(x1[i] - x5[i]));
        printf("wb %2d: LDL, two 16-column eliminations + MFMA glue (v5) %.2f us, |X1 - X5| = %.3e\n", wb, t[5] * 1e3 / reps, d5);
        double d4 = 0;
        for (int i = 0; i < NB * NB; ++i) d4 = fmax(d4, fabs(x1[i] - x4[i]));
        printf("wb %2d: LDL (v3) %.2f us, LDL + LDS broadcast (v4) %.2f us, load/store only %.2f us; max |X1 - X3| = %.3e, |X1 - X4| = %.3e\n", wb, t[2] * 1e3 / reps,
               t[4] * 1e3 / reps, t[3] * 1e3 / reps, d3, d4);
        printf("wb %2d: v1 %.2f us, v2 %.2f us per block (incl. ~0.4 us of load/store); max |X1 - X2| = %.3e (max |X| %.3e); |X2^T X2 A - I| = %.3e\n",
               wb, t[0] * 1e3 / reps, t[1] * 1e3 / reps, dmax, xmax, emax);
    }
    return 0;
}
