// Check of wave_sum_cols<N> (femo_alpha_amd/csrc/shell_device.h) on the device: every lane holds N values, the function must
// return, in every lane, the sum over the 64 lanes of the column it reports.   hipcc --offload-arch=gfx950 -O3 -I../../femo_alpha_amd/csrc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "shell_device.h"
using namespace femo;
template <int N>
__global__ void k(const double* in, double* out, int* cols) {
    const int lane = threadIdx.x;
    double p[N];
    for (int i = 0; i < N; ++i) p[i] = in[lane * 32 + i];
    int col;
    const double v = wave_sum_cols<N>(p, lane, col);
    out[lane] = v; cols[lane] = col;
}
template <int N>
int run(const std::vector<double>& h, const double* din, double* dout, int* dcols) {
    hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, din, dout, dcols);
    std::vector<double> o(64); std::vector<int> c(64);
    hipMemcpy(o.data(), dout, 64 * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), dcols, 64 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    std::vector<int> seen(N, 0);
    for (int l = 0; l < 64; ++l) {
        if (c[l] < 0 || c[l] >= N) { ++bad; continue; }
        double s = 0; for (int j = 0; j < 64; ++j) s += h[j * 32 + c[l]];
        if (s != o[l]) ++bad;
        seen[c[l]]++;
    }
    for (int i = 0; i < N; ++i) if (seen[i] != 64 / N) ++bad;
    printf("N=%d: %s\n", N, bad ? "FAIL" : "ok");
    return bad;
}
int main() {
    std::vector<double> h(64 * 32);
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 32; ++i) h[l * 32 + i] = (double)((l * 131 + i * 17 + l * i) % 1000);   // integers: sums are exact
    double *din, *dout; int* dcols;
    hipMalloc(&din, h.size() * 8); hipMalloc(&dout, 64 * 8); hipMalloc(&dcols, 64 * 4);
    hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    int bad = run<4>(h, din, dout, dcols) + run<8>(h, din, dout, dcols) + run<16>(h, din, dout, dcols) + run<32>(h, din, dout, dcols);
    printf(bad ? "FAILED\n" : "PASSED\n");
    return bad != 0;
}
