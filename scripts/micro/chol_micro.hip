// micro-benchmark: latency of the register Cholesky+inverse (one wave) and of an LDS-parallel variant
#include "../../femo_alpha_amd/csrc/frontal.h"
#include <cstdio>
#include <vector>
using namespace femo;

__global__ void __launch_bounds__(64) k_reg(double* A, double* X, int reps) {
    const int lane = threadIdx.x;
    double a[NB];
    for (int it = 0; it < reps; ++it) {
#pragma unroll
        for (int c = 0; c < NB; ++c) a[c] = (lane < NB) ? A[(lane % NB) * NB + c] : (c == lane - NB ? 1.0 : 0.0);
        chol32_inverse(a, NB, lane);
        if (lane >= NB)
#pragma unroll
            for (int r = 0; r < NB; ++r) X[r * NB + lane - NB] = a[r] + it;
    }
}

// (measured: 14.7 us against 5.5 us for the v_readlane version -- the LDS round trip sits on every step's chain)
// The same elimination with the multipliers of a step broadcast through LDS: every lane stores its entry of column j
// (one ds_write_b64), and the updates of the columns j + 2 .. read L[c][j] back from a wave-uniform address (two columns
// per ds_read2_b64) instead of two v_readlane_b32 each -- 1.5 instead of 3 instructions per (step, column) pair; only
// the update of column j + 1, which feeds the next pivot, keeps the register broadcast.  `bc`: 64 doubles of LDS owned
// by the calling wave.
__device__ __forceinline__ int chol32_inverse_lds(double (&a)[NB], int wb, int lane, double* bc) {
    int bad = 0;
    double d = rl(a[0], 0);
    if (!(d > 0.0)) { if (0 < wb) bad += 1; d = fabs(d) > 1e-300 ? fabs(d) : 1.0; }
    double y = rsqrt_newton(d);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const double l = (lane == j) ? d * y : a[j] * y;
        a[j] = l;
        if (j + 2 < NB) bc[lane] = l;
        if (j + 1 < NB) {
            a[j + 1] -= l * rl(l, j + 1);
            asm volatile("" : "+v"(a[j + 1]));
            d = rl(a[j + 1], j + 1);
            if (j + 1 < wb) {
                if (!(d > 0.0)) { bad += 1; d = fabs(d) > 1e-300 ? fabs(d) : 1.0; }
            } else {
                d = 1.0;
            }
            y = rsqrt_newton(d);
        }
#pragma unroll
        for (int c = j + 2; c < NB; ++c) {
            a[c] -= l * bc[c];
            asm volatile("" : "+v"(a[c]));
        }
    }
    return bad;
}

__global__ void __launch_bounds__(64) k_reg_lds(double* A, double* X, int reps) {
    const int lane = threadIdx.x;
    __shared__ double bc[64];
    double a[NB];
    for (int it = 0; it < reps; ++it) {
#pragma unroll
        for (int c = 0; c < NB; ++c) a[c] = (lane < NB) ? A[(lane % NB) * NB + c] : (c == lane - NB ? 1.0 : 0.0);
        chol32_inverse_lds(a, NB, lane, bc);
        if (lane >= NB)
#pragma unroll
            for (int r = 0; r < NB; ++r) X[r * NB + lane - NB] = a[r] + it;
    }
}

// all 256 threads cooperate on every elimination step, matrix in LDS
__global__ void __launch_bounds__(256) k_lds(double* A, double* X, int reps) {
    __shared__ double a[NB][NB + 1];
    __shared__ double li[NB][NB + 1];
    const int tid = threadIdx.x;
    for (int it = 0; it < reps; ++it) {
        for (int idx = tid; idx < NB * NB; idx += 256) a[idx / NB][idx % NB] = A[idx];
        __syncthreads();
        for (int j = 0; j < NB; ++j) {
            const double d = a[j][j];
            double y = __builtin_amdgcn_rsq(d);
            double e = 1.0 - d * y * y; y = y + 0.5 * y * e;
            e = 1.0 - d * y * y; y = y + 0.5 * y * e;
            __syncthreads();
            if (tid < NB) { if (tid == j) a[j][j] = d * y; else if (tid > j) a[tid][j] *= y; }
            __syncthreads();
            const int r = tid % NB, c0 = tid / NB;    // 8 column groups
            if (r > j) for (int c = j + 1 + c0; c <= r; c += 8) a[r][c] -= a[r][j] * a[c][j];
            __syncthreads();
        }
        // inverse: thread c < 32 column c
        if (tid < NB) {
            const int c = tid;
            for (int r = 0; r < NB; ++r) {
                double s = (r == c) ? 1.0 : 0.0;
                for (int mm = c; mm < r; ++mm) s -= a[r][mm] * li[mm][c];
                li[r][c] = (r >= c) ? s / a[r][r] : 0.0;
            }
        }
        __syncthreads();
        for (int idx = tid; idx < NB * NB; idx += 256) X[idx] = li[idx / NB][idx % NB] + it;
    }
}

int main() {
    std::vector<double> h(NB * NB, 0.0);
    for (int i = 0; i < NB; ++i) for (int j = 0; j <= i; ++j) h[i * NB + j] = (i == j) ? 40.0 + i : 1.0 / (1 + i + j);
    for (int i = 0; i < NB; ++i) for (int j = i + 1; j < NB; ++j) h[i * NB + j] = h[j * NB + i];
    double *A, *X;
    hipMalloc(&A, sizeof(double) * NB * NB); hipMalloc(&X, sizeof(double) * NB * NB);
    hipMemcpy(A, h.data(), sizeof(double) * NB * NB, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 200;
    for (int variant = 0; variant < 3; ++variant) {
        for (int w = 0; w < 2; ++w) {
            hipEventRecord(e0);
            if (variant == 0) hipLaunchKernelGGL(k_reg, dim3(1), dim3(64), 0, 0, A, X, reps);
            else if (variant == 1) hipLaunchKernelGGL(k_lds, dim3(1), dim3(256), 0, 0, A, X, reps);
            else hipLaunchKernelGGL(k_reg_lds, dim3(1), dim3(64), 0, 0, A, X, reps);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (w) printf("variant %d: %.2f us per factor+inverse\n", variant, ms * 1e3 / reps);
        }
        std::vector<double> x(NB * NB);
        hipMemcpy(x.data(), X, sizeof(double) * NB * NB, hipMemcpyDeviceToHost);
        printf("  X[0][0]=%.6f X[5][2]=%.6f X[31][31]=%.6f\n", x[0] - (reps - 1), x[5 * NB + 2] - (reps - 1), x[31 * NB + 31] - (reps - 1));
    }
    return 0;
}
