// CSR pattern of the CG2 x CG1 stiffness matrix and the destination-sorted contribution map of k_csr_segmented, built on
// the device (once per mesh): the nel * ld^2 (row, column) keys are sorted with their contribution index (radix sort,
// hipCUB -- a plain library sort in a set-up step), run heads become CSR entries.  Replaces the host numpy build
// (femo_alpha_amd/csr.py: 3.5 s at 1 M DOF; kept as the cross-check of tests/test_gpu_parity.py::test_csr_assembly).
// Role in the reference: dolfinx create_matrix builds the sparsity pattern once per form (fea/utils_dolfinx.py:200-206).
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

namespace femo {

// cr_nn >= 0 (CG2CR1): the rotation of local slot k lives on the cell's edge midpoint k = P2 node nvc + k, numbered from cr_nn (= nn)
__device__ __forceinline__ int csr_cell_dof(int e, int i, int nel, int npc, int ndof_u, const int* __restrict__ cellp2, const int* __restrict__ cells,
                                            int cr_nn) {
    if (i < 3 * npc) return 3 * cellp2[(size_t)(i / 3) * nel + e] + i % 3;
    const int k = i - 3 * npc;
    if (cr_nn >= 0) return ndof_u + 3 * (cellp2[(size_t)(3 + k / 3) * nel + e] - cr_nn) + k % 3;
    return ndof_u + 3 * cells[(size_t)(k / 3) * nel + e] + k % 3;
}

__global__ void k_csr_keys(long long ncontrib, int nel, int ld, int npc, int ndof_u, long long ndof, const int* __restrict__ cellp2,
                           const int* __restrict__ cells, long long* __restrict__ keys, int* __restrict__ vals, int cr_nn) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ncontrib) return;
    const int e = (int)(k / (ld * ld)), r = (int)(k % (ld * ld));
    const int i = r / ld, j = r % ld;
    keys[k] = (long long)csr_cell_dof(e, i, nel, npc, ndof_u, cellp2, cells, cr_nn) * ndof + csr_cell_dof(e, j, nel, npc, ndof_u, cellp2, cells, cr_nn);
    vals[k] = (int)k;
}

__global__ void k_csr_heads(long long n, const long long* __restrict__ keys, int* __restrict__ flags) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) flags[t] = (t == 0 || keys[t] != keys[t - 1]) ? 1 : 0;
}

// dest = (inclusive scan of the head flags) - 1 in place; heads write their column and count their row
__global__ void k_csr_pattern(long long n, const long long* __restrict__ keys, const int* __restrict__ flags, int* __restrict__ dest,
                              long long ndof, int* __restrict__ colidx, int* __restrict__ rowcount) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int d = dest[t] - 1;
    dest[t] = d;
    if (flags[t]) {
        colidx[d] = (int)(keys[t] % ndof);
        atomicAdd(&rowcount[keys[t] / ndof + 1], 1);
    }
}

}  // namespace femo
