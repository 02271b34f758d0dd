// latency of dependent fp64 instructions on one wave (gfx950): v_fma_f64 chain, v_rcp_f64 chain, v_readlane -> v_fma chain
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_fma(double* o, int n, double x) {
    double a = x;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 64; ++k) a = __builtin_fma(a, 1.0000001, 1e-9);
    }
    o[threadIdx.x] = a;
}
__global__ void k_rcp(double* o, int n, double x) {
    double a = x;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 64; ++k) a = __builtin_amdgcn_rcp(a);
    }
    o[threadIdx.x] = a;
}
__global__ void k_rl(double* o, int n, double x) {
    double a = x + threadIdx.x;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            int lo = __double2loint(a), hi = __double2hiint(a);
            lo = __builtin_amdgcn_readlane(lo, 5); hi = __builtin_amdgcn_readlane(hi, 5);
            a = __builtin_fma(a, 0.5, __hiloint2double(hi, lo));
        }
    }
    o[threadIdx.x] = a;
}
__global__ void k_fma_indep(double* o, int n, double x) {      // 8 independent chains: issue rate
    double a[8];
    for (int k = 0; k < 8; ++k) a[k] = x + k;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 64; ++k) a[k & 7] = __builtin_fma(a[k & 7], 1.0000001, 1e-9);
    }
    double s = 0; for (int k = 0; k < 8; ++k) s += a[k];
    o[threadIdx.x] = s;
}
int main() {
    double* o; hipMalloc(&o, 64 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 2000;
    const char* names[] = {"dependent v_fma_f64", "dependent v_rcp_f64", "readlane x2 + fma (dependent)", "independent v_fma_f64 (8 chains)"};
    for (int v = 0; v < 4; ++v)
        for (int w = 0; w < 2; ++w) {
            hipEventRecord(e0);
            if (v == 0) hipLaunchKernelGGL(k_fma, dim3(1), dim3(64), 0, 0, o, n, 1.0);
            else if (v == 1) hipLaunchKernelGGL(k_rcp, dim3(1), dim3(64), 0, 0, o, n, 1.3);
            else if (v == 2) hipLaunchKernelGGL(k_rl, dim3(1), dim3(64), 0, 0, o, n, 1.0);
            else hipLaunchKernelGGL(k_fma_indep, dim3(1), dim3(64), 0, 0, o, n, 1.0);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (w) printf("%-36s %.1f ns per step  (%.0f cycles at 2.4 GHz)\n", names[v], ms * 1e6 / (n * 64.0), ms * 1e6 / (n * 64.0) * 2.4);
        }
    return 0;
}
