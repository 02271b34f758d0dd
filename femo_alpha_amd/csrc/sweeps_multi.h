// Triangular sweeps with several right-hand sides at once (round 6).
//
// An application of the factor streams every factor byte once (2 x 2.7 GB at 1 M DOF, 1.5 ms) for ONE vector; the reference
// registers four and more outputs on `disp_solid` (compliance, elastic_energy, pnorm_stress, per-tag aggregates:
// rm_shell_model.py:221-253) and solves their adjoints one at a time (state_operation.py:188-220).  The kernels below are the sweep
// kernels of frontal.h (same tiling, same schedule, same index maps) with NR = 2 or 4 vectors INTERLEAVED -- entry d of vector r at
// [d * NR + r] -- so that a factor entry is loaded once and multiplied into NR accumulators, a gathered index fetches NR adjacent
// doubles, and the atomics of a tile go to adjacent addresses.  The plain kernels stay what they are (NR = 1 is not routed here).
#pragma once

namespace femo {

// forward, one workgroup per front (levels of many small fronts): y_p = L11^-1 v_p -> yv ; v_B -= L21 y_p
template <int NR>
__global__ void __launch_bounds__(256, 3)
k_front_fwd_small_m(FrontDev fd, const int* __restrict__ level_nodes, double* __restrict__ v, double* __restrict__ yv) {
    const int t = level_nodes[blockIdx.x];
    const int np = fd.npiv[t], nf = fd.nf[t];
    if (np == 0) return;
    const double* F = fd.P + fd.poff[t];
    const int ldp = ldp_of(nf);
    const int* gd = fd.dofs + fd.doff[t];
    extern __shared__ double sh[];
    double* y = sh;                          // np * NR
    double* part = sh + (size_t)np * NR;     // SMALL_PART * NR
    const int tid = threadIdx.x;
    for (int i = tid; i < np * NR; i += 256) y[i] = v[(size_t)gd[i / NR] * NR + i % NR];
    __syncthreads();
    const int npan = (np + NB - 1) / NB;
    for (int k = 0; k < npan; ++k) {
        const int c0 = k * NB, wb = min(NB, np - c0);
        const double* Li = fd.Linv + fd.linvoff[t] + (size_t)k * NB * NB;
        {
            const int r = tid & 31, g = tid >> 5;
            double a[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int mm = g + 8 * q; a[q] = (mm <= r && r < wb) ? Li[r + NB * mm] : 0.0; }
            double s[NR];
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) s[rr] = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = c0 + min(g + 8 * q, wb - 1);
#pragma unroll
                for (int rr = 0; rr < NR; ++rr) s[rr] += a[q] * y[col * NR + rr];
            }
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) part[(32 * g + r) * NR + rr] = s[rr];
        }
        __syncthreads();
        if (tid < 32) {
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                double s = 0.0;
#pragma unroll
                for (int g = 0; g < 8; ++g) s += part[(32 * g + tid) * NR + rr];
                part[(256 + tid) * NR + rr] = s;
                if (tid < wb) y[(c0 + tid) * NR + rr] = s;
            }
        }
        __syncthreads();
        const int r1 = c0 + wb;
        if (r1 < np) {
            const int h = tid >> 7;
            for (int rb = r1; rb < np; rb += 128) {
                const int r = rb + (tid & 127);
                double a[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) { const int mm = 16 * h + q; a[q] = (r < np && mm < wb) ? F[r + (size_t)ldp * (c0 + mm)] : 0.0; }
                double s[NR];
#pragma unroll
                for (int rr = 0; rr < NR; ++rr) s[rr] = 0.0;
#pragma unroll
                for (int q = 0; q < 16; ++q)
#pragma unroll
                    for (int rr = 0; rr < NR; ++rr) s[rr] += a[q] * part[(256 + 16 * h + q) * NR + rr];
                if (h)
#pragma unroll
                    for (int rr = 0; rr < NR; ++rr) part[(tid & 127) * NR + rr] = s[rr];
                __syncthreads();
                if (!h && r < np)
#pragma unroll
                    for (int rr = 0; rr < NR; ++rr) y[r * NR + rr] -= s[rr] + part[(tid & 127) * NR + rr];
                __syncthreads();
            }
        }
    }
    for (int i = tid; i < np * NR; i += 256) yv[(size_t)gd[i / NR] * NR + i % NR] = y[i];
    const int nb = nf - np;
    if (nb > 0) {
        const int slots = nb <= 64 ? 64 : nb <= 128 ? 128 : 256, G = 256 / slots;
        const int rs = tid % slots, g = tid / slots;
        const int cper = (np + G - 1) / G, cbeg = g * cper, cend = min(np, cbeg + cper);
        for (int rb = 0; rb < nb; rb += slots) {
            const int r = rb + rs;
            const double* row = F + np + min(r, nb - 1);
            double s[NR];
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) s[rr] = 0.0;
            for (int cb = cbeg; cb < cend; cb += 32) {
                double a[32];
#pragma unroll
                for (int q = 0; q < 32; ++q) a[q] = cb + q < cend ? row[(size_t)ldp * (cb + q)] : 0.0;
#pragma unroll
                for (int q = 0; q < 32; ++q) {
                    const int col = min(cb + q, np - 1);
#pragma unroll
                    for (int rr = 0; rr < NR; ++rr) s[rr] += a[q] * y[col * NR + rr];
                    // the LDS reads of the vectors stay with their group of eight columns: hoisted all at once they cost 32 x NR doubles
                    // of registers beside the 32 loads in flight
                }
            }
            if (G > 1) {
                if (g)
#pragma unroll
                    for (int rr = 0; rr < NR; ++rr) part[(slots * (g - 1) + rs) * NR + rr] = s[rr];
                __syncthreads();
                if (!g)
                    for (int gg = 1; gg < G; ++gg)
#pragma unroll
                        for (int rr = 0; rr < NR; ++rr) s[rr] += part[(slots * (gg - 1) + rs) * NR + rr];
            }
            if (!g && r < nb) {
                double* dst = v + (size_t)gd[np + r] * NR;
#pragma unroll
                for (int rr = 0; rr < NR; ++rr) atomicAdd(dst + rr, -s[rr]);
            }
            if (G > 1) __syncthreads();
        }
    }
}

// backward, one workgroup per front: x_p = L11^-T (y_p - L21^T x_B)
template <int NR>
__global__ void __launch_bounds__(256, 3)
k_front_bwd_small_m(FrontDev fd, const int* __restrict__ level_nodes, const double* __restrict__ sv, double* __restrict__ xv) {
    const int t = level_nodes[blockIdx.x];
    const int np = fd.npiv[t], nf = fd.nf[t];
    if (np == 0) return;
    const double* F = fd.P + fd.poff[t];
    const int ldp = ldp_of(nf);
    const int* gd = fd.dofs + fd.doff[t];
    extern __shared__ double sh[];
    double* x = sh;                          // nf * NR: s_p (then x_p) in [0, np), x_B behind
    double* part = sh + (size_t)nf * NR;     // SMALL_PART * NR
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < nf * NR; i += 256) {
        const int p = i / NR;
        x[i] = p < np ? sv[(size_t)gd[p] * NR + i % NR] : xv[(size_t)gd[p] * NR + i % NR];
    }
    __syncthreads();
    const int nb = nf - np;
    // a wave's pass over eight columns (lanes along the rows): NR x 8 sums, each set of eight in one butterfly
    auto columns8 = [&](const double* base, int nrows, int xoff, int cb, int ncol, int c00) {
        double s[8][NR];
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) s[q][rr] = 0.0;
        for (int rb = 0; rb < nrows; rb += 256) {
            double a[8][4];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const double* col = base + (size_t)ldp * (c00 + min(cb + q, ncol - 1));
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int r = rb + lane + 64 * u; a[q][u] = r < nrows ? col[r] : 0.0; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = rb + lane + 64 * u;
                const double* xr = x + (size_t)(xoff + min(r, nrows - 1)) * NR;
#pragma unroll
                for (int rr = 0; rr < NR; ++rr) {
                    const double xv_ = xr[rr];
#pragma unroll
                    for (int q = 0; q < 8; ++q) s[q][rr] += a[q][u] * xv_;
                }
            }
        }
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) {
            double p8[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) p8[q] = s[q][rr];
            int q;
            const double tot = wave_sum_cols<8>(p8, lane, q);
            if (!(lane & 7) && cb + q < ncol) x[(size_t)(c00 + cb + q) * NR + rr] -= tot;     // a column belongs to one wave
        }
    };
    if (nb > 0) {
        for (int cb = 8 * wid; cb < np; cb += 32) columns8(F + np, nb, np, cb, np, 0);
        __syncthreads();
    }
    const int npan = (np + NB - 1) / NB;
    for (int k = npan - 1; k >= 0; --k) {
        const int c0 = k * NB, wb = min(NB, np - c0);
        const double* Li = fd.Linv + fd.linvoff[t] + (size_t)k * NB * NB;
        const int r1 = c0 + wb, nr = np - r1;
        if (nr > 0) {
            columns8(F + r1, nr, r1, 8 * wid, wb, c0);
            __syncthreads();
        }
        {
            const int c = tid & 31, g = tid >> 5;
            double a[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int r = c + g + 8 * q; a[q] = (r < wb && c < wb) ? Li[r + NB * c] : 0.0; }
            double s[NR];
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) s[rr] = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = c0 + min(c + g + 8 * q, wb - 1);
#pragma unroll
                for (int rr = 0; rr < NR; ++rr) s[rr] += a[q] * x[(size_t)r * NR + rr];
            }
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) part[(32 * g + c) * NR + rr] = s[rr];
        }
        __syncthreads();
        if (tid < wb) {
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                double s = 0.0;
#pragma unroll
                for (int g = 0; g < 8; ++g) s += part[(32 * g + tid) * NR + rr];
                x[(size_t)(c0 + tid) * NR + rr] = s;
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < np * NR; i += 256) xv[(size_t)gd[i / NR] * NR + i % NR] = x[i];
}

// wide levels, plain products: 128 x 128 tile of X (TRI) or L21
template <bool TRI, int NR>
__global__ void __launch_bounds__(256, 2)
k_sweep_gemv_n_m(FrontDev fd, const int* __restrict__ level_nodes, int first, const double* __restrict__ in, double* __restrict__ out) {
    const int t = level_nodes[first + blockIdx.y], bx = (int)blockIdx.x;
    const int np = fd.npiv[t], nf = fd.nf[t];
    if (np == 0) return;
    const int nct = (np + 127) / 128;
    int ti, tj;
    if (TRI) {
        const int lin = bx;
        if (lin >= nct * (nct + 1) / 2) return;
        ti = (int)((sqrt(8.0 * lin + 1.0) - 1.0) * 0.5);
        while ((ti + 1) * (ti + 2) / 2 <= lin) ++ti;
        while (ti * (ti + 1) / 2 > lin) --ti;
        tj = lin - ti * (ti + 1) / 2;
    } else {
        const int nrt = (nf - np + 127) / 128;
        if (bx >= nrt * nct) return;
        ti = bx / nct; tj = bx % nct;
    }
    const int* gd = fd.dofs + fd.doff[t];
    const int ld = TRI ? ldx_of(np) : ldp_of(nf);
    const double* M = TRI ? fd.X + fd.xoff[t] : fd.P + fd.poff[t] + np;
    const int nrows = TRI ? np : nf - np;
    const int cw = TRI ? 128 : ((np + nct - 1) / nct + 1) & ~1, hw = cw / 2;
    const int r0 = 128 * ti, c0 = cw * tj;
    __shared__ double xs[128 * NR];
    __shared__ double part[128 * NR];
    const int tid = threadIdx.x, lr = tid & 127, ch = tid >> 7;
    for (int i = tid; i < 128 * NR; i += 256) {
        const int p = i / NR;
        xs[i] = (p < cw && c0 + p < np) ? in[(size_t)gd[c0 + p] * NR + i % NR] : 0.0;
    }
    const int r = r0 + lr;
    const double* row = M + r + (size_t)ld * (c0 + hw * ch);
    const int cmax = min(np - c0 - hw * ch, hw);
    const int clim = (TRI && ti == tj) ? min(cmax, lr - hw * ch + 1) : cmax;
    double a[32];
    double s[NR];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) s[rr] = 0.0;
    __syncthreads();
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int k = 0; k < 32; ++k) a[k] = (r < nrows && 32 * h + k < clim) ? row[(size_t)ld * (32 * h + k)] : 0.0;
        int xoff = (hw * ch + 32 * h) * NR;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            // The address of this group's LDS reads is tied to the previous group's sums: left to itself the compiler issues the reads of
            // all 32 columns x NR vectors at once (256 registers at NR = 4, beside the 32 loads in flight: 394 in all, one workgroup per CU)
            if (NR > 1) asm volatile("" : "+v"(xoff), "+v"(s[0]));
#pragma unroll
            for (int k = 8 * kb; k < 8 * kb + 8; ++k)
#pragma unroll
                for (int rr = 0; rr < NR; ++rr) s[rr] += a[k] * xs[xoff + k * NR + rr];
        }
    }
    if (ch)
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) part[lr * NR + rr] = s[rr];
    __syncthreads();
    if (!ch && r < nrows) {
        double* dst = out + (size_t)gd[TRI ? r : np + r] * NR;
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) atomicAdd(dst + rr, TRI ? s[rr] + part[lr * NR + rr] : -(s[rr] + part[lr * NR + rr]));
    }
}

// wide levels, transposed products: tile of 128 rows x 128 columns, wave w owns columns 32 w .. 32 w + 31
template <bool TRI, int NR>
__global__ void __launch_bounds__(256)
k_sweep_gemv_t_m(FrontDev fd, const int* __restrict__ level_nodes, int first, const double* __restrict__ in, double* __restrict__ out) {
    const int t = level_nodes[first + blockIdx.y], bx = (int)blockIdx.x;
    const int np = fd.npiv[t], nf = fd.nf[t];
    if (np == 0) return;
    const int nct = (np + 127) / 128;
    int ti, tj;
    if (TRI) {
        const int lin = bx;
        if (lin >= nct * (nct + 1) / 2) return;
        ti = (int)((sqrt(8.0 * lin + 1.0) - 1.0) * 0.5);
        while ((ti + 1) * (ti + 2) / 2 <= lin) ++ti;
        while (ti * (ti + 1) / 2 > lin) --ti;
        tj = lin - ti * (ti + 1) / 2;
    } else {
        const int nrt = (nf - np + 127) / 128;
        if (bx >= nrt * nct) return;
        ti = bx / nct; tj = bx % nct;
    }
    const int* gd = fd.dofs + fd.doff[t];
    const int ld = TRI ? ldx_of(np) : ldp_of(nf);
    const double* M = TRI ? fd.X + fd.xoff[t] : fd.P + fd.poff[t] + np;
    const int nrows = TRI ? np : nf - np;
    const int rbase = TRI ? 0 : np;
    const int r0 = 128 * ti, c0 = 128 * tj;
    __shared__ double xs[128 * NR];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 128 * NR; i += 256) {
        const int p = i / NR;
        xs[i] = (r0 + p < nrows) ? in[(size_t)gd[rbase + r0 + p] * NR + i % NR] : 0.0;
    }
    const int ra = r0 + lane, rb = r0 + lane + 64;
    double a0[32], a1[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const int c = c0 + 32 * wv + k;
        const double* col = M + (size_t)ld * c;
        const bool ca = c < np && ra < nrows && (!TRI || ra >= c);
        const bool cb = c < np && rb < nrows && (!TRI || rb >= c);
        a0[k] = ca ? col[ra] : 0.0;
        a1[k] = cb ? col[rb] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        const double x0 = xs[lane * NR + rr], x1 = xs[(lane + 64) * NR + rr];
        double p[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) p[k] = a0[k] * x0 + a1[k] * x1;
        int col;
        const double sum = wave_sum_cols<32>(p, lane, col);
        const int c = c0 + 32 * wv + col;
        if (!(lane & 1) && c < np) atomicAdd(&out[(size_t)gd[c] * NR + rr], TRI ? sum : -sum);
    }
}

// s_p = y_p - L21^T x_B with one workgroup per 16 pivot columns and all boundary rows of the front (no atomics)
template <int NR>
__global__ void __launch_bounds__(256)
k_sweep_bnd_cols_m(FrontDev fd, const int* __restrict__ level_nodes, int first, double* __restrict__ sv, const double* __restrict__ xv) {
    extern __shared__ double xs[];                       // nb * NR
    const int t = level_nodes[first + blockIdx.y], bx = (int)blockIdx.x;
    const int np = fd.npiv[t], nf = fd.nf[t];
    const int nb = nf - np;
    const int c0 = bx * BB_COLS;
    if (c0 >= np || nb == 0) return;
    const int* gd = fd.dofs + fd.doff[t];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < nb * NR; i += 256) xs[i] = xv[(size_t)gd[np + i / NR] * NR + i % NR];
    __syncthreads();
    const int ldp = ldp_of(nf);
    const double* L21 = fd.P + fd.poff[t] + np;
    const int cb = c0 + 4 * wv;
    const double* col[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) col[k] = L21 + (size_t)ldp * min(cb + k, np - 1);
    double s[4][NR];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) s[k][rr] = 0.0;
    for (int rb = 0; rb < nb; rb += 512) {
        double a[4][8];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int r = rb + lane + 64 * u; a[k][u] = r < nb ? col[k][r] : 0.0; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double* xr = xs + (size_t)min(rb + lane + 64 * u, nb - 1) * NR;
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                const double xv_ = xr[rr];
#pragma unroll
                for (int k = 0; k < 4; ++k) s[k][rr] += a[k][u] * xv_;
            }
        }
    }
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        double p4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) p4[k] = s[k][rr];
        int k;
        const double tot = wave_sum_cols<4>(p4, lane, k);
        if (!(lane & 15) && cb + k < np) sv[(size_t)gd[cb + k] * NR + rr] -= tot;
    }
}

// NR separate vectors <-> one interleaved buffer
struct VecPtrs { double* p[4]; };
template <int NR>
__global__ void k_interleave(VecPtrs src, double* __restrict__ dst, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) dst[i * NR + rr] = src.p[rr][i];
}
template <int NR>
__global__ void k_deinterleave(const double* __restrict__ src, VecPtrs dst, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) dst.p[rr][i] = src[i * NR + rr];
}

}  // namespace femo
