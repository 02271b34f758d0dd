// micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 and of v_fma_f64 (registers only)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_mfma(double* out, int iters) {
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_fma(double* out, int iters) {
    double acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = i;
    double a = threadIdx.x * 1e-3, b = 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = fma(a, acc[i], b);
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    double* out; hipMalloc(&out, sizeof(double) * 256 * 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int wpc : {1, 2, 4}) {            // workgroups per CU
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_mfma, dim3(256 * wpc), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double fl = 256.0 * wpc * 4 * iters * 16 * 2048.0;
            if (rep) printf("mfma f64 16x16x4, %d WG/CU: %.1f TFLOP/s\n", wpc, fl / (ms * 1e-3) / 1e12);
        }
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_fma, dim3(256 * wpc), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double fl = 256.0 * wpc * 256 * iters * 16 * 2.0;
            if (rep) printf("valu fma f64, %d WG/CU: %.1f TFLOP/s\n", wpc, fl / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
