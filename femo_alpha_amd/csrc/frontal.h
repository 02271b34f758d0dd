// Multifrontal Cholesky of the shell operator on gfx950 (fp64): numeric factorisation and the
// triangular solves used as the PCG preconditioner.
//
// Role in the reference: PETSc KSP 'preonly' + PC 'lu' with MUMPS
// (reference femo_alpha/fea/utils_dolfinx.py:466,495-531) -- itself a multifrontal method.
// Here the elimination tree comes from a geometric nested dissection of the elements
// (femo_alpha_amd/solver/symbolic.py); element matrices are summed straight into the leaf fronts;
// every level of the tree is factorised by batched, blocked partial Cholesky kernels whose grids span
// all fronts of the level: parents are gathered from their children's Schur complements, then per
// outer panel of 128 columns the diagonal block is factorised and inverted by one workgroup per front
// (32x32 register Cholesky + fp64 MFMA), the rows below become factor rows by one GEMM against that
// inverse, and 64x64 tiles receive the rank-k updates on the matrix cores (right-looking with
// look-ahead at the top of the tree, left-looking on the levels with many fronts).
//
// Storage: a front is a dense nf x nf matrix of which only the lower triangle is maintained, kept in two places.  Its
// first npiv columns -- [L11; L21] after factorisation, i.e. the factor itself -- live for good in the panel store P
// (column-major, leading dimension nf, at P + poff[t]).  The trailing nb x nb block, the Schur complement that only the
// parent's extend-add consumes, lives in an arena S (leading dimension nb, at S + soff[t]) whose regions are reused
// as the factorisation climbs the tree (one region per parent level, placed at plan time so that regions alive together
// never overlap): 5.4 GB instead of 14.9 GB of square fronts at 1 M DOF.  Kernels address both through FrontView.  Linv keeps the inverses of the NB x NB diagonal
// blocks of L11, Sinv those of the 128 x 128 diagonal blocks (levels with few fronts), so that every
// triangular solve becomes a GEMV.
#pragma once
#include "shell_device.h"

namespace femo {

constexpr int NB = 32;     // panel width
constexpr int TS = 64;     // trailing-update tile
constexpr int NBO = 128;   // outer panel: the wide trailing update applies this many factor columns per pass
constexpr int SPD = 128;   // leading dimension of the stored diagonal-block inverses (== SP of the solve kernels)

struct FrontDev {
    int ntree;
    const int* nf;
    const int* npiv;
    const long long* poff;      // doubles, [ntree+1]: pivot columns of front t in P (nf x npiv, leading dimension nf)
    const long long* soff;      // doubles, [ntree]: Schur complement of front t in S (nb x nb, leading dimension nb)
    const long long* doff;      // ints, [ntree+1]
    const int* dofs;            // global DOF of every front row
    const int* upmap;           // row of the parent front (boundary rows only)
    const int* parent;
    const int* child[2];        // left / right
    const int* cinv[2];         // per front row: the row of the left / right child's front that lands there, or -1
    const long long* linvoff;   // doubles, [ntree+1]
    const long long* xoff;      // doubles, [ntree+1]: where the front's X = L11^-1 starts (fronts of the wide levels)
    double* P;
    double* S;
    double* Linv;
    double* X;                  // L11^-1, lower triangle, column-major with leading dimension ldx_of(npiv)
    double* Xtmp;               // scratch of the same shape (the products C XA of the recursive inversion)
};

// entries of one front by FRONT row and column (r >= c): pivot columns in the panel store, the rest in the Schur arena
// leading dimension of the pivot columns in the panel store: nf rounded up to even, so that every column starts on a
// 16-byte boundary and the rank-k update can stage two rows of a factor column per load (the pad row stays zero)
__device__ __host__ inline int ldp_of(int nf) { return (nf + 1) & ~1; }

struct FrontView {
    double* P;
    double* S;
    int nf, np;
    __device__ __forceinline__ double* col(int c) const {
        return c < np ? P + (size_t)ldp_of(nf) * c : S + (size_t)(nf - np) * (c - np) - np;     // then [r], r >= c
    }
};
__device__ __forceinline__ FrontView front_view(const FrontDev& fd, int t) {
    FrontView v;
    v.nf = fd.nf[t]; v.np = fd.npiv[t];
    v.P = fd.P + fd.poff[t];
    v.S = fd.S + fd.soff[t];
    return v;
}

// leading dimension of a front's X (a multiple of the 128-column outer panel, so that k_diag_block can write the
// inverse of every diagonal block straight into place)
__device__ __host__ inline int ldx_of(int np) { return (np + SPD - 1) / SPD * SPD; }

// ------------------------------------------------------------------------------------------ assembly
// one wave per element; lane j evaluates column j of K_e (operator applied to e_j) and adds its
// lower-triangle entries into the element's leaf front.  Masked (strong-BC) rows/columns are skipped.
template <int NPC, int NVC, bool QUAD, bool UHAT, bool MASS>
__global__ void __launch_bounds__(64)          // 239 registers, two waves per SIMD; capped at 167 (three waves, 52 B of scratch): 1.33 against 1.07 ms
k_front_assemble(MeshDev m, FieldsDev f, const Tables* __restrict__ tab, double aK, double aM, FrontDev fd,
                 const int* __restrict__ elem_front, const int* __restrict__ elem_map, const unsigned char* __restrict__ mask,
                 const double* __restrict__ eq) {
    constexpr int LD = 3 * NPC + 3 * NVC;
    const int e = blockIdx.x;
    const int j = threadIdx.x;
    if (e >= m.nel) return;
    Elem<NPC, NVC> el;
    load_elem<NPC, NVC, UHAT>(m, f, e, el);
    // the quadrature-point data all lanes share: computed once, by lane q for point q, and parked in LDS (stage_qpoints)
    extern __shared__ double sq_raw[];
    QPoint<NPC, NVC>* sq = reinterpret_cast<QPoint<NPC, NVC>*>(sq_raw);
    stage_qpoints<NPC, NVC, QUAD, UHAT>(tab, el, aK, j, 64, sq);
    if (j >= LD) return;
    double ye[LD];
#pragma unroll
    for (int i = 0; i < LD; ++i) ye[i] = 0.0;
    // the lane's unit vector e_j: displacement component cj of P2 node aj, or rotation component cj of vertex aj
    const bool is_u = j < 3 * NPC;
    const int aj = is_u ? j / 3 : (j - 3 * NPC) / 3;
    const int cj = j - 3 * (is_u ? aj : NPC + aj);
    const int nq = tab->nq;
    for (int q = 0; q < nq; ++q) {
        const QPoint<NPC, NVC>& p = sq[q];
        // strains of e_j without the 39-entry reduction: the reduced vectors are scalar multiples of one unit vector;
        // the lane reads its own node's table row (a per-lane index into d[][] would send the array to scratch)
        const double r0 = is_u ? tab->dN2[q][aj][0] : tab->dNR[q][aj][0], r1 = is_u ? tab->dN2[q][aj][1] : tab->dNR[q][aj][1];
        const double dk0 = r0 * p.g.Q[0][0] + r1 * p.g.Q[1][0], dk1 = r0 * p.g.Q[0][1] + r1 * p.g.Q[1][1];
        const double Mj = is_u ? 0.0 : tab->NR[q][aj];
        double G0[3], G1[3], th[3], T0[3], T1[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double ec = (c == cj) ? 1.0 : 0.0;
            G0[c] = is_u ? dk0 * ec : 0.0;
            G1[c] = is_u ? dk1 * ec : 0.0;
            th[c] = Mj * ec;
            T0[c] = is_u ? 0.0 : dk0 * ec;
            T1[c] = is_u ? 0.0 : dk1 * ec;
        }
        const Gen s = strains_reduced(p.g, G0, G1, th, T0, T1);
        const Gen t = stress_of(s, p.mat);
        strains_T<NPC, NVC>(p.g, p.d, p.mm, tab->NR[q], t, ye);
    }
    if (MASS) {
        // The inertia term in a loop of its own (compiled out of the static operator): inside the stiffness loop its live values pushed
        // the kernel from 239 registers past 256 and to ONE wave per SIMD (front assembly of config 5: 1.95 against 1.09 ms).  Column j
        // of the element mass matrix, written out for the unit vector e_j: entries only where the component matches, N_i N_j for the
        // displacement, h_K^2 N_i N_j for the rotation (mass_qp)
        double rhon[NVC];
#pragma unroll
        for (int b = 0; b < NVC; ++b) rhon[b] = f.rho[f.ewm ? e : el.vid[b]];
        const double hk2 = is_u ? 1.0 : el.hK * el.hK;
        for (int q = 0; q < nq; ++q) {
            const QPoint<NPC, NVC>& p = sq[q];
            double rq = 0.0;
#pragma unroll
            for (int b = 0; b < NVC; ++b) rq += tab->N1[q][b] * rhon[b];
            const double cmj = aM * rq * p.hq * tab->w[q] * p.g.det * p.g.Ju * hk2 * (is_u ? tab->N2[q][aj] : tab->NR[q][aj]);
#pragma unroll
            for (int a = 0; a < NPC; ++a)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) ye[3 * a + cc] += (is_u && cc == cj) ? cmj * tab->N2[q][a] : 0.0;
#pragma unroll
            for (int b = 0; b < NVC; ++b)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) ye[3 * NPC + 3 * b + cc] += (!is_u && cc == cj) ? cmj * tab->NR[q][b] : 0.0;
        }
    }
    const int t = elem_front[e];
    const FrontView fv = front_view(fd, t);
    const int* map = elem_map + (size_t)e * LD;
    const int* gd = fd.dofs + fd.doff[t];
    // K_e is symmetric: the lane's column j is also row j.  Adding it as a ROW makes the 39 lanes of one atomic
    // instruction hit one column of the front (a few cache lines) instead of 39 different columns.
    const int pj = map[j];
    if (mask && mask[gd[pj]]) return;
    // eq (option "equilibrate", an experiment): the matrix factorised is D K D with D = diag(eq)
    const double sj = eq ? eq[gd[pj]] : 1.0;
#pragma unroll
    for (int i = 0; i < LD; ++i) {
        const int pi = map[i];
        if (pj >= pi && !(mask && mask[gd[pi]])) atomicAdd(fv.col(pi) + pj, eq ? ye[i] * sj * eq[gd[pi]] : ye[i]);
    }
}

// ---- front-centric assembly (r5, option "assemble_fc"): ONE workgroup per leaf front.
// k_front_assemble is bound by the memory side: float atomics never execute in L2 on this chip (MI355X_MICROARCH.md, "Global float
// atomics": every wave-instruction leaves L2 as one uncached 64-byte request per segment it touches, ~20 G requests/s chip-wide), and a
// column of K_e lands on ~10 segments of its front -- 26 M requests, 1.3 of the kernel's 1.43 ms at 1 M DOF, whatever the arithmetic
// does (threads sharing the quadrature points of a column changed nothing).  Here a front belongs to one workgroup for the whole
// kernel: it zeroes the front (no k_zero_fronts launch), then takes the front's elements ONE AFTER THE OTHER -- thread (j, part)
// evaluates column j of K_e over every PARTS-th quadrature point (39 x 3 of 128 threads on quadrilaterals, 27 x 2 of 64 on triangles:
// one wave per element used 39 / 27 of its 64 lanes), the partial columns meet in LDS, and the sums are added to the front with
// plain loads and stores: no other workgroup touches this front, and the elements of one front are separated by workgroup barriers.
// The front is written in L2 and leaves it once.
__device__ __host__ constexpr int assemble_block(int ld) { return ld > 32 ? 128 : 64; }
__device__ __host__ constexpr size_t assemble_lds(int ld, size_t qpoint_bytes) {   // the staged points, then the partial columns in the same place
    const size_t part = (size_t)(assemble_block(ld) / ld) * ld * (ld | 1) * sizeof(double);
    return part > qpoint_bytes ? part : qpoint_bytes;
}
template <int NPC, int NVC, bool QUAD, bool UHAT, bool MASS>
__global__ void __launch_bounds__(assemble_block(3 * NPC + 3 * NVC))
k_front_assemble_fc(MeshDev m, FieldsDev f, const Tables* __restrict__ tab, double aK, double aM, FrontDev fd,
                    const int* __restrict__ level_nodes, const int* __restrict__ fel_off, const int* __restrict__ fel,
                    const int* __restrict__ elem_map, const unsigned char* __restrict__ mask, const double* __restrict__ eq) {
    constexpr int LD = 3 * NPC + 3 * NVC;
    constexpr int BLK = assemble_block(LD), PARTS = BLK / LD, PS = LD | 1, NK = (LD + PARTS - 1) / PARTS;
    const int slot = blockIdx.x;
    const int t = level_nodes[slot];
    const int tid = threadIdx.x;
    const int part = tid / LD, j = tid - part * LD;           // part == PARTS: the lanes left over (they zero and stage, nothing else)
    const bool work = part < PARTS;
    const FrontView fv = front_view(fd, t);
    const int nf = fv.nf, np = fv.np;
    // the front starts from zero: its pivot columns (rows c.. of column c; the pad row of an odd front stays as it is) and its Schur block
    {
        const int ldp = ldp_of(nf);
        for (int c = 0; c < np; ++c) {
            double* col = fv.P + (size_t)ldp * c;
            for (int r = c + tid; r < nf; r += BLK) col[r] = 0.0;
        }
        const int nb = nf - np;
        for (int c = 0; c < nb; ++c) {
            double* col = fv.S + (size_t)nb * c;
            for (int r = c + tid; r < nb; r += BLK) col[r] = 0.0;
        }
    }
    extern __shared__ double sq_raw[];
    QPoint<NPC, NVC>* sq = reinterpret_cast<QPoint<NPC, NVC>*>(sq_raw);
    double* pb = sq_raw;                                      // partial columns: [part][j][i], rows of PS doubles (odd: conflict-free)
    const int* gd = fd.dofs + fd.doff[t];
    const bool is_u = j < 3 * NPC;
    const int aj = is_u ? j / 3 : (j - 3 * NPC) / 3;
    const int cj = j - 3 * (is_u ? aj : NPC + aj);
    const int e0 = fel_off[slot], e1 = fel_off[slot + 1];
    for (int ei = e0; ei < e1; ++ei) {
        const int e = fel[ei];
        Elem<NPC, NVC> el;
        load_elem<NPC, NVC, UHAT>(m, f, e, el);
        __syncthreads();                                      // the previous element's partial columns have been read (and the zero fill is done)
        stage_qpoints<NPC, NVC, QUAD, UHAT>(tab, el, aK, tid, BLK, sq);
        double ye[LD];
#pragma unroll
        for (int i = 0; i < LD; ++i) ye[i] = 0.0;
        const int nq = work ? tab->nq : 0;
        for (int q = part; q < nq; q += PARTS) {
            const QPoint<NPC, NVC>& p = sq[q];
            const double r0 = is_u ? tab->dN2[q][aj][0] : tab->dNR[q][aj][0], r1 = is_u ? tab->dN2[q][aj][1] : tab->dNR[q][aj][1];
            const double dk0 = r0 * p.g.Q[0][0] + r1 * p.g.Q[1][0], dk1 = r0 * p.g.Q[0][1] + r1 * p.g.Q[1][1];
            const double Mj = is_u ? 0.0 : tab->NR[q][aj];
            double G0[3], G1[3], th[3], T0[3], T1[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double ec = (c == cj) ? 1.0 : 0.0;
                G0[c] = is_u ? dk0 * ec : 0.0;
                G1[c] = is_u ? dk1 * ec : 0.0;
                th[c] = Mj * ec;
                T0[c] = is_u ? 0.0 : dk0 * ec;
                T1[c] = is_u ? 0.0 : dk1 * ec;
            }
            const Gen s = strains_reduced(p.g, G0, G1, th, T0, T1);
            const Gen tt = stress_of(s, p.mat);
            strains_T<NPC, NVC>(p.g, p.d, p.mm, tab->NR[q], tt, ye);
        }
        if (MASS) {
            double rhon[NVC];
#pragma unroll
            for (int b = 0; b < NVC; ++b) rhon[b] = f.rho[f.ewm ? e : el.vid[b]];
            const double hk2 = is_u ? 1.0 : el.hK * el.hK;
            for (int q = part; q < nq; q += PARTS) {
                const QPoint<NPC, NVC>& p = sq[q];
                double rq = 0.0;
#pragma unroll
                for (int b = 0; b < NVC; ++b) rq += tab->N1[q][b] * rhon[b];
                const double cmj = aM * rq * p.hq * tab->w[q] * p.g.det * p.g.Ju * hk2 * (is_u ? tab->N2[q][aj] : tab->NR[q][aj]);
#pragma unroll
                for (int a = 0; a < NPC; ++a)
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) ye[3 * a + cc] += (is_u && cc == cj) ? cmj * tab->N2[q][a] : 0.0;
#pragma unroll
                for (int b = 0; b < NVC; ++b)
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) ye[3 * NPC + 3 * b + cc] += (!is_u && cc == cj) ? cmj * tab->NR[q][b] : 0.0;
            }
        }
        __syncthreads();                                      // every thread is through with the staged points
        if (work) {
#pragma unroll
            for (int i = 0; i < LD; ++i) pb[((size_t)part * LD + j) * PS + i] = ye[i];
        }
        __syncthreads();
        if (work) {
            // thread (j, part) adds the rows i = part, part + PARTS, ... of column j (K_e is symmetric: as ROW j of the front's column i,
            // so that the threads of a wave write along a column)
            const int* map = elem_map + (size_t)e * LD;
            const int pj = map[j];
            const bool mj = mask && mask[gd[pj]];
            const double sj = eq ? eq[gd[pj]] : 1.0;
            double* dst[NK];
            bool ok[NK];
            double v[NK], old[NK];
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int i = min(part + PARTS * k, LD - 1);
                const int pi = map[i];
                ok[k] = part + PARTS * k < LD && pj >= pi && !mj && !(mask && mask[gd[pi]]);
                double x = 0.0;
#pragma unroll
                for (int pp = 0; pp < PARTS; ++pp) x += pb[((size_t)pp * LD + j) * PS + i];
                v[k] = eq ? x * sj * eq[gd[pi]] : x;
                dst[k] = ok[k] ? fv.col(pi) + pj : fd.P;         // (unconditional loads from a safe address: all of them in flight at once)
            }
#pragma unroll
            for (int k = 0; k < NK; ++k) old[k] = *dst[k];
#pragma unroll
            for (int k = 0; k < NK; ++k) if (ok[k]) *dst[k] = old[k] + v[k];
        }
    }
}

// penalty facet blocks into the leaf front of the facet's element
__global__ void k_front_penalty(FacetDev pf, FrontDev fd, const int* __restrict__ elem_front, const int* __restrict__ elem_map,
                                int ld, int npc, int nvc, const unsigned char* __restrict__ mask, const double* __restrict__ eq) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pf.nf) return;
    const int e = pf.cell[i], k = pf.ledge[i];
    const int t = elem_front[e];
    const FrontView fv = front_view(fd, t);
    const int* map = elem_map + (size_t)e * ld;
    const int* gd = fd.dofs + fd.doff[t];
    const int kb = (k + 1) % nvc;
    const int un[3] = {k, npc == nvc ? k : nvc + k, kb};          // element-local P2 nodes (a, mid, b); CG1CG1: no mid node, its slot of M2 is empty
    const int vn[2] = {k, kb};                   // element-local vertices
    for (int c = 0; c < 3; ++c) {
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                const int pa = map[3 * un[a] + c], pb = map[3 * un[b] + c];
                if (pa >= pb && !(mask && (mask[gd[pa]] || mask[gd[pb]])))
                    atomicAdd(fv.col(pb) + pa, pf.M2[9 * i + 3 * a + b] * (eq ? eq[gd[pa]] * eq[gd[pb]] : 1.0));
            }
        if (pf.MR) {
            // CG2CR1: the 3 x 3 rotation block over the cell's three edge midpoints (element-local rotation slots 0, 1, 2)
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) {
                    const int pa = map[3 * npc + 3 * a + c], pb = map[3 * npc + 3 * b + c];
                    if (pa >= pb && !(mask && (mask[gd[pa]] || mask[gd[pb]])))
                        atomicAdd(fv.col(pb) + pa, pf.MR[9 * i + 3 * a + b] * (eq ? eq[gd[pa]] * eq[gd[pb]] : 1.0));
                }
        } else
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b) {
                const int pa = map[3 * npc + 3 * vn[a] + c], pb = map[3 * npc + 3 * vn[b] + c];
                if (pa >= pb && !(mask && (mask[gd[pa]] || mask[gd[pb]])))
                    atomicAdd(fv.col(pb) + pa, pf.M1[4 * i + 2 * a + b] * (eq ? eq[gd[pa]] * eq[gd[pb]] : 1.0));
            }
    }
}

// unit diagonal for masked pivots (their rows / columns were skipped during assembly)
__global__ void k_front_mask_diag(FrontDev fd, const unsigned char* __restrict__ mask) {
    const int t = blockIdx.x;
    const int np = fd.npiv[t], nf = fd.nf[t];
    double* P = fd.P + fd.poff[t];
    const int* gd = fd.dofs + fd.doff[t];
    for (int p = threadIdx.x; p < np; p += blockDim.x)
        if (mask[gd[p]]) P[p + (size_t)ldp_of(nf) * p] = 1.0;
}

// lower triangle of every front of a level := 0 (the leaf fronts before the element matrices are added)
__global__ void __launch_bounds__(256)
k_zero_fronts(FrontDev fd, const int* __restrict__ level_nodes, int first) {
    const int t = level_nodes[first + blockIdx.y];
    const int nf = fd.nf[t];
    const int nt = (nf + TS - 1) / TS;
    const int lin = blockIdx.x;
    if (lin >= nt * (nt + 1) / 2) return;
    int ti = (int)((sqrt(8.0 * lin + 1.0) - 1.0) * 0.5);
    while ((ti + 1) * (ti + 2) / 2 <= lin) ++ti;
    while (ti * (ti + 1) / 2 > lin) --ti;
    const int tj = lin - ti * (ti + 1) / 2;
    const FrontView fv = front_view(fd, t);
    const int r0 = ti * TS, c0 = tj * TS;
    for (int idx = threadIdx.x; idx < TS * TS; idx += blockDim.x) {
        const int r = r0 + idx % TS, cc = c0 + idx / TS;
        if (r < nf && cc <= r) fv.col(cc)[r] = 0.0;
    }
}

// extend-add as a gather: every entry of the parent's lower triangle is the sum of the entries of its children's
// Schur complements that land there (row maps cinv) -- written once, never read: no zero fill of the parent, no
// read-modify-write, one launch per level for both children.  Masked (strong-BC) pivots get their unit diagonal here.
__global__ void __launch_bounds__(256)
k_extend_gather(FrontDev fd, const int* __restrict__ level_nodes, int first, const unsigned char* __restrict__ mask, int skip_schur) {
    const int p = level_nodes[first + blockIdx.y];
    const int ch0 = fd.child[0][p], ch1 = fd.child[1][p];
    if (ch0 < 0 && ch1 < 0) return;                      // nothing below: the front keeps what it was given
    const int nfp = fd.nf[p], npp = fd.npiv[p];
    const int nt = (nfp + TS - 1) / TS;
    const int lin = blockIdx.x;
    if (lin >= nt * (nt + 1) / 2) return;
    int ti = (int)((sqrt(8.0 * lin + 1.0) - 1.0) * 0.5);
    while ((ti + 1) * (ti + 2) / 2 <= lin) ++ti;
    while (ti * (ti + 1) / 2 > lin) --ti;
    const int tj = lin - ti * (ti + 1) / 2;
    const int r0 = ti * TS, c0 = tj * TS;
    // skip_schur: only the first outer panel's columns are filled here -- every other column of this level's fronts is
    // gathered from the children by the rank-k update that touches it first (k_trailing_mfma<true>); fronts without
    // pivots have no such update
    const int c_end = (skip_schur && npp > 0) ? min(npp, NBO) : nfp;
    if (c0 >= c_end) return;
    __shared__ int rmap[2][TS], cmap[2][TS];
    const long long dp = fd.doff[p];
    for (int i = threadIdx.x; i < 4 * TS; i += blockDim.x) {
        const int side = (i / TS) & 1, isc = i / (2 * TS), k = i % TS;
        const int g = (isc ? c0 : r0) + k;
        const int v = (g < nfp && (side ? ch1 : ch0) >= 0) ? fd.cinv[side][dp + g] : -1;
        if (isc) cmap[side][k] = v; else rmap[side][k] = v;
    }
    __syncthreads();
    const FrontView fp = front_view(fd, p);
    // the children are read in their Schur complements only (rows and columns beyond their pivots)
    const FrontView f0 = front_view(fd, ch0 >= 0 ? ch0 : p), f1 = front_view(fd, ch1 >= 0 ? ch1 : p);
    const int* gd = fd.dofs + dp;
    // a thread owns row lr of the tile and 16 of its columns: all child reads are issued before the first store
    const int lr = threadIdx.x % TS, lc0 = threadIdx.x / TS;
    const int r = r0 + lr;
    const int ra0 = rmap[0][lr], ra1 = rmap[1][lr];
    double v[TS / 4];
#pragma unroll
    for (int k = 0; k < TS / 4; ++k) {
        const int lc = lc0 + 4 * k, cc = c0 + lc;
        double x = 0.0;
        if (r < nfp && cc <= r && cc < c_end) {
            const int b0 = cmap[0][lc], b1 = cmap[1][lc];
            if (ra0 >= 0 && b0 >= 0) x += f0.col(min(ra0, b0))[max(ra0, b0)];
            if (ra1 >= 0 && b1 >= 0) x += f1.col(min(ra1, b1))[max(ra1, b1)];
        }
        v[k] = x;
    }
#pragma unroll
    for (int k = 0; k < TS / 4; ++k) {
        const int cc = c0 + lc0 + 4 * k;
        if (r < nfp && cc <= r && cc < c_end) {
            double x = v[k];
            if (mask && r == cc && r < npp && mask[gd[r]]) x = 1.0;
            fp.col(cc)[r] = x;
        }
    }
}

// ------------------------------------------------------------------------------------------ factorisation
// Cholesky of a 32x32 block and the inverse of its factor, in the registers of one wave.
// Lanes 0..31 own the rows of the block, lanes 32..63 the columns of an identity: running the same
// elimination on both halves (A = L L^T on the left, L X = I on the right) yields the factor and its inverse
// from one instruction stream -- the multipliers L[c][j] are the same for both.
// On entry a[] = row `lane` of the block (lanes < 32) or column `lane - 32` of the identity (lanes >= 32);
// on return lane r < 32 holds row r of L, lane 32 + c holds column c of L^-1 (a[r] = Linv[r][c]).
// Pivots and multipliers travel by v_readlane (static lane ids).
// Returns the number of non-positive pivots that had to be repaired.
// broadcast of a double from a compile-time lane: two v_readlane_b32 (a 64-wide __shfl here sends hipcc's
// optimiser into a compile that does not finish within minutes once the loops below are unrolled)
__device__ __forceinline__ double rl(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double rsqrt_newton(double d) {
    // 1/sqrt(d): hardware estimate + two Newton steps (no fp64 sqrt / divide expansion on the critical path)
    double y = __builtin_amdgcn_rsq(d);
    double e = 1.0 - d * y * y;
    y = y + 0.5 * y * e;
    e = 1.0 - d * y * y;
    y = y + 0.5 * y * e;
    return y;
}

// W = width of the block (32: lanes 0..31 rows, 32..63 identity columns; 16: lanes 0..15 rows, 16..31 identity columns, and
// whatever rows lanes 32.. carry receive the same column operations, i.e. become rows of L21 = A21 L11^-T for free).
template <int W>
__device__ __forceinline__ int chol_inverse_w(double (&a)[W], int wb, int lane) {
    int bad = 0;
    // pivot of step 0
    double d = rl(a[0], 0);
    if (!(d > 0.0)) { if (0 < wb) bad += 1; d = fabs(d) > 1e-300 ? fabs(d) : 1.0; }
    double y = rsqrt_newton(d);
#pragma unroll
    for (int j = 0; j < W; ++j) {
        // left half: column j of L for this lane's row (valid for lane >= j); right half: row j of L^-1
        const double l = (lane == j) ? d * y : a[j] * y;
        a[j] = l;
        if (j + 1 < W) {
            // the next pivot is final after the first update: start its reciprocal square root now, so that this
            // dependent chain (~150 cycles) runs beside the remaining updates of this step instead of after them
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
        for (int c = j + 2; c < W; c += 2) {
            // two broadcasts ahead of two updates: no wait states between a v_readlane and the FMA that consumes it
            const double lc0 = rl(l, c);                      // L[c][j]
            const double lc1 = (c + 1 < W) ? rl(l, c + 1) : 0.0;
            a[c] -= l * lc0;                                  // left half: only entries with c <= row are meaningful
            if (c + 1 < W) a[c + 1] -= l * lc1;
            // pin the updates to this step: a[c] is not consumed before step c, and left to itself the compiler
            // defers the FMAs until then, keeping every broadcast alive in SGPRs (1500 of them spilled)
            asm volatile("" : "+v"(a[c]));
            if (c + 1 < W) asm volatile("" : "+v"(a[c + 1]));
        }
    }
    return bad;
}

__device__ __forceinline__ int chol32_inverse(double (&a)[NB], int wb, int lane) { return chol_inverse_w<NB>(a, wb, lane); }

// The same factor + inverse without a square root on the critical path: A = L~ D L~^T (unit lower L~), one pivot per step.
// The register elimination above spends ~500 cycles per step, most of them waiting: pivot -> rsq + two Newton steps -> scale
// the column -> broadcast -> update the next diagonal entry -> broadcast it -> next pivot, a chain of ~20 dependent
// instructions.  Here the next pivot is formed on the scalar side,  d' = A[j+1][j+1] - A[j+1][j]^2 / d,  from two broadcasts
// that do not wait for this step (both entries were final one step earlier): the chain is  d -> rcp + two Newton steps -> one
// FMA -> d',  and everything else -- the multipliers m = a[j] / d, their broadcasts, the updates a[c] -= a[j] m_c -- hangs off
// it instead of sitting on it.  Left lanes (rows of A) and right lanes (columns of the identity, i.e. the forward
// substitution z = L~^-1 e_k) run the same instruction stream, as above.
// On return: lane 32 + k holds z (a[r] = (L~^-1)[r][k]); *rs = 1 / sqrt(d_lane) for lanes < 32, so that the inverse of the
// Cholesky factor L = L~ D^1/2 is  X[r][k] = rs_r a[r].  REPAIR: non-positive pivots are replaced on the chain (option
// "allow_pivot_repair"); otherwise they are only counted -- the factorisation fails anyway.
__device__ __forceinline__ double rcp_newton(double d) {
    double y = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    return y;
}

// Width W as in chol_inverse_w: W = 16 leaves lanes 32.. free to carry rows below the block (they become L~21 D).
template <int W, bool REPAIR>
__device__ __forceinline__ int ldl_inverse_w(double (&a)[W], int wb, int lane, double* rs) {
    int bad = 0;
    double dvec = 1.0;                              // lane j keeps the pivot d_j
    double d = rl(a[0], 0);
    if (REPAIR && !(d > 0.0)) { if (0 < wb) bad += 1; d = fabs(d) > 1e-300 ? fabs(d) : 1.0; }
#pragma unroll
    for (int j = 0; j < W; ++j) {
        dvec = (lane == j) ? d : dvec;
        const double r = rcp_newton(d);
        if (j + 1 < W) {
            const double s = rl(a[j], j + 1), s2 = rl(a[j + 1], j + 1);
            d = __builtin_fma(-(s * s), r, s2);
            if (REPAIR && !(d > 0.0)) { if (j + 1 < wb) bad += 1; d = fabs(d) > 1e-300 ? fabs(d) : 1.0; }
        }
        const double m = a[j] * r;                  // left lanes: L~[row][j]
#pragma unroll
        for (int c = j + 1; c < W; c += 2) {
            const double m0 = rl(m, c);
            const double m1 = (c + 1 < W) ? rl(m, c + 1) : 0.0;
            a[c] -= a[j] * m0;
            if (c + 1 < W) a[c + 1] -= a[j] * m1;
            asm volatile("" : "+v"(a[c]));
            if (c + 1 < W) asm volatile("" : "+v"(a[c + 1]));
        }
    }
    if (!REPAIR) bad = __popcll(__ballot(lane < wb && !(dvec > 0.0)));
    *rs = rsqrt_newton(lane < W ? dvec : 1.0);
    return bad;
}

template <bool REPAIR>
__device__ __forceinline__ int ldl32_inverse(double (&a)[NB], int wb, int lane, double* rs) { return ldl_inverse_w<NB, REPAIR>(a, wb, lane, rs); }

// ldl32_inverse with the multipliers of a step broadcast through LDS: a wave issues one instruction per ~4 cycles whatever its
// kind, and the register broadcast costs two v_readlane_b32 per (step, column) on top of the FMA.  Here every lane stores its
// multiplier once (ds_write_b64) and the updates read m_c back from wave-uniform addresses, two columns per ds_read -- 1.5
// instead of 3 instructions per pair.  The round trip through LDS does not touch the pivot chain (see above), and the one
// update that feeds the NEXT step's multipliers, column j + 1, keeps the register broadcast.  All reads of a step are issued
// before its first FMA (one pin per step, not per column: a pin per column made every read wait for its own latency --
// the round-2 LDS variant of the Cholesky elimination was 2.7 x slower than the register one for that reason).
// `bc`: 64 doubles of LDS owned by the calling wave.
template <bool REPAIR>
__device__ __forceinline__ int ldl32_inverse_lds(double (&a)[NB], int wb, int lane, double* rs, double* bc) {
    int bad = 0;
    double dvec = 1.0;
    double d = rl(a[0], 0);
    if (REPAIR && !(d > 0.0)) { if (0 < wb) bad += 1; d = fabs(d) > 1e-300 ? fabs(d) : 1.0; }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        dvec = (lane == j) ? d : dvec;
        const double r = rcp_newton(d);
        if (j + 1 < NB) {
            const double s = rl(a[j], j + 1), s2 = rl(a[j + 1], j + 1);
            d = __builtin_fma(-(s * s), r, s2);
            if (REPAIR && !(d > 0.0)) { if (j + 1 < wb) bad += 1; d = fabs(d) > 1e-300 ? fabs(d) : 1.0; }
        }
        const double m = a[j] * r;
        if (j + 2 < NB) bc[lane] = m;
        if (j + 1 < NB) a[j + 1] -= a[j] * rl(m, j + 1);
        if (j + 2 < NB) {
            double mc[NB];
#pragma unroll
            for (int c = j + 2; c < NB; ++c) mc[c] = bc[c];
#pragma unroll
            for (int c = j + 2; c < NB; ++c) a[c] -= a[j] * mc[c];
        }
        // pin the whole step: left to itself the compiler defers updates to the step that consumes them
#pragma unroll
        for (int c = j + 1; c < NB; ++c) asm volatile("" : "+v"(a[c]));
    }
    if (!REPAIR) bad = __popcll(__ballot(lane < wb && !(dvec > 0.0)));
    *rs = rsqrt_newton(lane < NB ? dvec : 1.0);
    return bad;
}

// ---- outer panel of <= 128 columns in two launches ----------------------------------------------------------
// 16x16 MFMA sub-block product helpers: wave w of a 4-wave workgroup owns sub-block (w & 1, w >> 1) of a 32x32
// result.  v_mfma_f64_16x16x4_f64: a lane supplies A[i = lane & 15][k = lane >> 4] and B[k = lane >> 4][j = lane & 15]
// and holds D[i = (lane >> 4) + 4 reg][j = lane & 15].
typedef double mfma_d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
typedef double blk32[NB][NB + 1];

// acc += A(32 x 32) * B(32 x 32) restricted to this wave's sub-block; TA / TB: the operand is stored transposed
template <bool TA, bool TB>
__device__ __forceinline__ void mfma_blk(mfma_d4& acc, const blk32& A, const blk32& B, int si, int sj, int l15, int l4) {
#pragma unroll
    for (int kk = 0; kk < NB; kk += 4) {
        const double a = TA ? A[kk + l4][16 * si + l15] : A[16 * si + l15][kk + l4];
        const double b = TB ? B[16 * sj + l15][kk + l4] : B[kk + l4][16 * sj + l15];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
}

// One wave: acc += M1 (16 x 16) M2 (16 x 16), both row-major in LDS with leading dimension LDB = NB + 1 (quadrants of a blk32).
// NEG1: the first operand enters negated.  TB: M2 is stored transposed (the product is M1 M2^T).
template <bool NEG1, bool TB>
__device__ __forceinline__ void mm16(mfma_d4& acc, const double* M1, const double* M2, int l15, int l4) {
#pragma unroll
    for (int kk = 0; kk < 16; kk += 4) {
        const double a = M1[l15 * (NB + 1) + kk + l4];
        const double b = TB ? M2[l15 * (NB + 1) + kk + l4] : M2[(kk + l4) * (NB + 1) + l15];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(NEG1 ? -a : a, b, acc, 0, 0, 0);
    }
}

// The inverse of the Cholesky factor of the 32 x 32 block B (lower triangle valid, in LDS), by ONE wave, in place: on return
// B holds X = L^-1 (lower triangle, zeros above).  Two register factorisations of 16 columns instead of one of 32 -- a step of
// the register elimination costs its pivot chain (~110 cycles) plus three instructions per remaining column, and half of the
// columns halve that -- with the 16 x 16 glue on the matrix cores, staged through the free quadrants of B itself:
//   A  lanes 0..15 rows of B11, 16..31 identity, 32..47 rows of B21:  L11, X11 = L11^-1, L21 = B21 L11^-T     (registers)
//   B  B22 -= L21 L21^T                                                                                       (MFMA)
//   C  lanes 0..15 rows of B22, 16..31 identity:  L22, X22                                                    (registers)
//   D  X21 = -X22 (L21 X11)                                                                                   (MFMA)
// LDS operations of one wave execute in order, so the phases need no barrier.  Returns the number of repaired pivots.
template <bool LDL, bool REPAIR>
__device__ __forceinline__ int chol32_inverse_v2(blk32& B, int wb, int lane) {
    constexpr int H = NB / 2, LDB = NB + 1;
    const int l15 = lane & 15, l4 = lane >> 4;
    double* b00 = &B[0][0];
    double* b01 = b00 + H;                  // upper-right quadrant: scratch for L21 (zeroed at the end)
    double* b10 = b00 + H * LDB;            // B21, then T = L21 X11, then X21
    double* b11 = b10 + H;                  // B22, then X22
    double a[H];
    {
        const int grp = lane >> 4;          // 0 rows of B11, 1 identity, 2 rows of B21, 3 unused
        const double* src = grp == 2 ? b10 + l15 * LDB : b00 + l15 * LDB;
#pragma unroll
        for (int c = 0; c < H; ++c) {
            const double v = src[c];
            a[c] = (grp == 1) ? (c == l15 ? 1.0 : 0.0) : (grp == 3 ? 0.0 : v);
        }
    }
    int bad;
    if (LDL) {
        // LDL^T elimination (no square root on the pivot chain), then L = L~ D^1/2: columns of the left rows (lanes 0..15 and
        // 32..47: a[k] = L~[.][k] d_k) and rows of the inverse (lanes 16..31: a[r] = (L~^-1)[r][.]) take the same factor rs_i
        double rs;
        bad = ldl_inverse_w<H, REPAIR>(a, min(wb, H), lane, &rs);
#pragma unroll
        for (int i = 0; i < H; ++i) a[i] *= rl(rs, i);
    } else {
        bad = chol_inverse_w<H>(a, min(wb, H), lane);
    }
    // X11 into place (lane 16 + c holds its column c), L21 into the scratch quadrant (lane 32 + i holds its row i)
    if (l4 == 1) {
#pragma unroll
        for (int r = 0; r < H; ++r) b00[r * LDB + l15] = (r >= l15 && r < wb && l15 < wb) ? a[r] : 0.0;
    } else if (l4 == 2) {
#pragma unroll
        for (int k = 0; k < H; ++k) b01[l15 * LDB + k] = a[k];
    }
    __builtin_amdgcn_wave_barrier();
    {   // B22 -= L21 L21^T (upper triangle: garbage in, garbage out -- never read)
        mfma_d4 acc;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) acc[reg] = b11[(l4 + 4 * reg) * LDB + l15];
        mm16<true, true>(acc, b01, b01, l15, l4);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) b11[(l4 + 4 * reg) * LDB + l15] = acc[reg];
    }
    __builtin_amdgcn_wave_barrier();
    {
        const double* src = b11 + l15 * LDB;
#pragma unroll
        for (int c = 0; c < H; ++c) {
            const double v = src[c];
            a[c] = (l4 == 1) ? (c == l15 ? 1.0 : 0.0) : (l4 == 0 ? v : 0.0);
        }
    }
    const int wb2 = max(wb - H, 0);
    if (LDL) {
        double rs;
        bad += ldl_inverse_w<H, REPAIR>(a, wb2, lane, &rs);
#pragma unroll
        for (int i = 0; i < H; ++i) a[i] *= rl(rs, i);
    } else {
        bad += chol_inverse_w<H>(a, wb2, lane);
    }
    if (l4 == 1) {
#pragma unroll
        for (int r = 0; r < H; ++r) b11[r * LDB + l15] = (r >= l15 && r < wb2 && l15 < wb2) ? a[r] : 0.0;
    }
    __builtin_amdgcn_wave_barrier();
    {   // T = L21 X11 -> b10;  X21 = -X22 T -> b10
        mfma_d4 t = (mfma_d4){0.0, 0.0, 0.0, 0.0};
        mm16<false, false>(t, b01, b00, l15, l4);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) b10[(l4 + 4 * reg) * LDB + l15] = t[reg];
        __builtin_amdgcn_wave_barrier();
        mfma_d4 x = (mfma_d4){0.0, 0.0, 0.0, 0.0};
        mm16<true, false>(x, b11, b10, l15, l4);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            b10[(l4 + 4 * reg) * LDB + l15] = x[reg];
            b01[(l4 + 4 * reg) * LDB + l15] = 0.0;
        }
    }
    __builtin_amdgcn_wave_barrier();
    return bad;
}

// PA: the diagonal block [C0, C0+kw) x [C0, C0+kw), kw <= 128, of every front of the level -- one workgroup per front.
// The block is held in LDS as 32x32 sub-blocks (lower block triangle).  For each block column j: wave 0 factorises
// the diagonal sub-block and inverts its factor in registers (chol32_inverse); the four waves then form
// L_ij = D_ij Linv_j^T and D_ik -= L_ij L_kj^T on the matrix cores.  Afterwards the inverse S of the whole kw x kw
// factor is accumulated block-wise (S_jj = Linv_j, S_ij = -Linv_i sum_{k=j}^{i-1} L_ik S_kj) and stored
// column-major with leading dimension 128: k_panel_rows turns the rows below the block into factor rows with one
// GEMM against S^T, and the wide triangular solves use the same S.
// Outputs: off-diagonal sub-blocks of L into F, the 32x32 inverses into Linv, S into Sout.
// The diagonal sub-blocks of F are left as they were (nothing reads them afterwards).
#ifdef FEMO_PANEL_STAMPS
__device__ long long g_stamps[32];
#define STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_stamps[i] = wall_clock64(); } while (0)
#else
#define STAMP(i)
#endif

// LDS is sized by the launch: `nblk` = the largest number of 32-column sub-blocks any front of the launch has in this
// panel (the host sorts the fronts of a level by pivot count and launches them in classes), so that the many small
// fronts at the bottom of the tree run two or three workgroups per CU instead of one.
// r3: the inverse needs no LDS of its own -- column j of S and the scratch block of a product live in the sub-blocks of L's
// column j, each of which is dead once its row's sum has been formed (112 -> 80 KB for four sub-blocks: two workgroups per CU).
__device__ __host__ inline int diag_block_lds_blocks(int nblk) { return nblk * (nblk + 1) / 2; }

// fuse_rows (fronts of ONE panel on levels that do not keep S): the rows below the block are turned into factor rows here, with
// S read from LDS where the inverse phase leaves it -- S never goes to memory and k_panel_rows is not launched.  On the leaves
// this kernel is HBM-bound (block in, factor sub-blocks, the 32 x 32 inverses and the 128 x 128 S out: 2.4 GB at 1 M DOF), and S is
// the largest item; k_panel_rows then read it back.
__global__ void __launch_bounds__(256)
k_diag_block(FrontDev fd, const int* __restrict__ level_nodes, int first, int nblk, int C0, double* __restrict__ Swork,
             int* __restrict__ info, int fuse_rows) {
    STAMP(31);
    const int slot = first + blockIdx.x;                       // position of the front in its level
    const int t = level_nodes[slot];
    const int np = fd.npiv[t];
    if (C0 >= np) return;
    const int kw = min(NBO, np - C0);
    const int nkb = (kw + NB - 1) / NB;
    const int ldp = ldp_of(fd.nf[t]);
    double* F = fd.P + fd.poff[t];                             // pivot columns only
    // the inverse of this diagonal block: scratch (levels solved with the one-workgroup-per-front kernels) or the diagonal
    // block of the front's X (wide levels)
    const int lds_ = Swork ? SPD : ldx_of(np);
    double* Sout = Swork ? Swork + (size_t)slot * SPD * SPD : fd.X + fd.xoff[t] + C0 + (size_t)lds_ * C0;
    extern __shared__ double lds_raw[];
    blk32* D = reinterpret_cast<blk32*>(lds_raw);              // sub-block (i, j), i >= j, at i (i + 1) / 2 + j
    const int nD = nblk * (nblk + 1) / 2;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int si = wv & 1, sj = wv >> 1, l15 = lane & 15, l4 = lane >> 4;
    // load (identity padding beyond kw; only the lower triangle of the front is maintained): all 40 loads of a thread
    // are issued before the first one is consumed -- a loop over the sub-blocks would pay the memory latency ten times
    {
        double v[10][4];
#pragma unroll
        for (int b = 0; b < 10; ++b) {
            const int bi = b < 1 ? 0 : b < 3 ? 1 : b < 6 ? 2 : 3;
            const int bj = b - bi * (bi + 1) / 2;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = tid + 256 * q;
                const int r = idx % NB, c = idx / NB;
                const int gr = NB * bi + r, gc = NB * bj + c;
                v[b][q] = (gr == gc) ? 1.0 : 0.0;
                if (b < nD && bi < nkb && gr < kw && gc < kw && gc <= gr) v[b][q] = F[(C0 + gr) + (size_t)ldp * (C0 + gc)];
            }
        }
#pragma unroll
        for (int b = 0; b < 10; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = tid + 256 * q;
                if (b < nD) D[b][idx % NB][idx / NB] = v[b][q];
            }
    }
    // (no zero fill of S: every sub-block on or below the block diagonal is written in full below, and nothing above it
    //  or beyond the kw columns is ever read)
    STAMP(0);
    __syncthreads();
    STAMP(1);
    for (int j = 0; j < nkb; ++j) {
        blk32& Djj = D[j * (j + 1) / 2 + j];
        const int wb = min(NB, kw - NB * j);
        STAMP(2 + 4 * j);
        if (wv == 0) {
            double a[NB];
#pragma unroll
            for (int c = 0; c < NB; ++c) a[c] = (lane < NB) ? Djj[lane][c] : (c == lane - NB ? 1.0 : 0.0);
            const int bad = chol32_inverse(a, wb, lane);
            if (lane >= NB) {
                const int cl = lane - NB;               // this lane holds column cl of the inverse
                double* Li = fd.Linv + fd.linvoff[t] + (size_t)(C0 / NB + j) * NB * NB;
#pragma unroll
                for (int r = 0; r < NB; ++r) {
                    const double v = (cl < wb && r < wb && cl <= r) ? a[r] : 0.0;
                    Djj[r][cl] = v;                     // the diagonal sub-block now holds Linv_j
                    Li[r + NB * cl] = v;
                }
                if (cl == 0 && bad) atomicAdd(info, bad);
            }
        }
        __syncthreads();
        STAMP(3 + 4 * j);
        if (j + 1 < nkb) {
            // L_ij = D_ij Linv_j^T for the sub-blocks below
            mfma_d4 acc[3];
#pragma unroll
            for (int i = 1; i < 4; ++i) {
                acc[i - 1] = (mfma_d4){0.0, 0.0, 0.0, 0.0};
                if (j + i < nkb) mfma_blk<false, true>(acc[i - 1], D[(j + i) * (j + i + 1) / 2 + j], Djj, si, sj, l15, l4);
            }
            __syncthreads();
#pragma unroll
            for (int i = 1; i < 4; ++i) {
                if (j + i >= nkb) continue;
                blk32& Dij = D[(j + i) * (j + i + 1) / 2 + j];
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int r = 16 * si + l4 + 4 * reg, c = 16 * sj + l15;
                    Dij[r][c] = acc[i - 1][reg];
                    const int gr = NB * (j + i) + r, gc = NB * j + c;
                    if (gr < kw && gc < kw) F[(C0 + gr) + (size_t)ldp * (C0 + gc)] = acc[i - 1][reg];
                }
            }
            __syncthreads();
            STAMP(4 + 4 * j);
            // D_ik -= L_ij L_kj^T, i >= k > j (each wave updates its own sub-block of every D_ik)
            for (int i = j + 1; i < nkb; ++i)
                for (int k = j + 1; k <= i; ++k) {
                    mfma_d4 u = (mfma_d4){0.0, 0.0, 0.0, 0.0};
                    mfma_blk<false, true>(u, D[i * (i + 1) / 2 + j], D[k * (k + 1) / 2 + j], si, sj, l15, l4);
                    blk32& Dik = D[i * (i + 1) / 2 + k];
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) Dik[16 * si + l4 + 4 * reg][16 * sj + l15] -= u[reg];
                }
            __syncthreads();
        }
    }
    // inverse of the kw x kw factor, block column by block column
    STAMP(20);
    for (int j = 0; j < nkb; ++j) {
        const blk32& Sjj = D[j * (j + 1) / 2 + j];
        if (!fuse_rows)
        for (int idx = tid; idx < NB * NB; idx += 256) {
            const int r = idx % NB, c = idx / NB;
            Sout[(NB * j + r) + (size_t)lds_ * (NB * j + c)] = Sjj[r][c];
        }
        for (int i = j + 1; i < nkb; ++i) {
            // W = sum_{k=j}^{i-1} L_ik S_kj: S_kj (j < k < i) already sits where L_kj was; L_ij is read here for the last time
            blk32& Dij = D[i * (i + 1) / 2 + j];
            mfma_d4 w = (mfma_d4){0.0, 0.0, 0.0, 0.0};
            for (int k = j; k < i; ++k)
                mfma_blk<false, false>(w, D[i * (i + 1) / 2 + k], D[k * (k + 1) / 2 + j], si, sj, l15, l4);
            __syncthreads();                                   // every wave has read L_ij: its sub-block takes W
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) Dij[16 * si + l4 + 4 * reg][16 * sj + l15] = w[reg];
            __syncthreads();
            // S_ij = -Linv_i W
            mfma_d4 x = (mfma_d4){0.0, 0.0, 0.0, 0.0};
            mfma_blk<false, false>(x, D[i * (i + 1) / 2 + i], Dij, si, sj, l15, l4);
            __syncthreads();                                   // every wave has read W: the sub-block takes S_ij
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = 16 * si + l4 + 4 * reg, c = 16 * sj + l15;
                Dij[r][c] = -x[reg];
                if (!fuse_rows) Sout[(NB * i + r) + (size_t)lds_ * (NB * j + c)] = -x[reg];
            }
            __syncthreads();
        }
    }
    STAMP(21);
    if (fuse_rows) {
        // L[r][C0 + c] = sum_{k <= c} A[r][C0 + k] S[c][k] for the rows below the block (k_panel_rows' product): a wave takes 16 rows
        // at a time, their kw entries in registers as the MFMA B operand, S[c][k] from the sub-blocks in LDS as the A operand
        __syncthreads();
        const int nf = fd.nf[t];
        for (int row0 = C0 + kw + 16 * wv; row0 < nf; row0 += 64) {
            const int row = row0 + l15;
            const bool rok = row < nf;
            double a[NBO / 4];
#pragma unroll
            for (int kk = 0; kk < NBO / 4; ++kk) {
                const int k = 4 * kk + l4;
                a[kk] = (rok && k < kw) ? F[row + (size_t)ldp * (C0 + k)] : 0.0;
            }
#pragma unroll
            for (int cb = 0; cb < NBO / 16; ++cb) {
                if (16 * cb >= kw) break;
                const int bi = cb >> 1;
                mfma_d4 acc = (mfma_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 4 * cb + 4; ++kk)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(D[bi * (bi + 1) / 2 + (kk >> 3)][16 * (cb & 1) + l15][4 * (kk & 7) + l4], a[kk], acc, 0, 0, 0);
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int c = 16 * cb + l4 + 4 * reg;
                    if (rok && c < kw) F[row + (size_t)ldp * (C0 + c)] = acc[reg];
                }
            }
        }
    }
}

// k_diag_block for the classes of at most NBLK = 1, 2 or 3 sub-blocks (round 6): the same phases with every loop bounded at compile
// time, so that a front of 72 pivots does not carry the registers of a 128-column block (188 VGPRs: two workgroups per CU whatever
// the LDS footprint) -- 168 registers, three workgroups per CU for one or two sub-blocks (capped at 128 the elimination spills
// 370 bytes) -- and with the LDL^T elimination of k_diag_block2 on
// wave 0.  The levels of many small fronts are bound by workgroups per CU times the latency of one (8192 leaf fronts / (256 CUs x 2)
// rounds of ~45 us), not by bytes or flops.
template <int NBLK, bool REPAIR, bool LDL>
__global__ void __launch_bounds__(256, NBLK <= 2 ? 3 : 2)
k_diag_block_t(FrontDev fd, const int* __restrict__ level_nodes, int first, int C0, double* __restrict__ Swork,
               int* __restrict__ info, int fuse_rows) {
    constexpr int nblk = NBLK, ND = NBLK * (NBLK + 1) / 2;
    STAMP(31);
    const int slot = first + blockIdx.x;                       // position of the front in its level
    const int t = level_nodes[slot];
    const int np = fd.npiv[t];
    if (C0 >= np) return;
    const int kw = min(NBO, np - C0);
    const int nkb = (kw + NB - 1) / NB;
    const int ldp = ldp_of(fd.nf[t]);
    double* F = fd.P + fd.poff[t];                             // pivot columns only
    // the inverse of this diagonal block: scratch (levels solved with the one-workgroup-per-front kernels) or the diagonal
    // block of the front's X (wide levels)
    const int lds_ = Swork ? SPD : ldx_of(np);
    double* Sout = Swork ? Swork + (size_t)slot * SPD * SPD : fd.X + fd.xoff[t] + C0 + (size_t)lds_ * C0;
    extern __shared__ double lds_raw[];
    blk32* D = reinterpret_cast<blk32*>(lds_raw);              // sub-block (i, j), i >= j, at i (i + 1) / 2 + j
    constexpr int nD = ND;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int si = wv & 1, sj = wv >> 1, l15 = lane & 15, l4 = lane >> 4;
    // load (identity padding beyond kw; only the lower triangle of the front is maintained): all 40 loads of a thread
    // are issued before the first one is consumed -- a loop over the sub-blocks would pay the memory latency ten times
    {
        double v[ND][4];
#pragma unroll
        for (int b = 0; b < ND; ++b) {
            const int bi = b < 1 ? 0 : b < 3 ? 1 : b < 6 ? 2 : 3;
            const int bj = b - bi * (bi + 1) / 2;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = tid + 256 * q;
                const int r = idx % NB, c = idx / NB;
                const int gr = NB * bi + r, gc = NB * bj + c;
                v[b][q] = (gr == gc) ? 1.0 : 0.0;
                if (b < nD && bi < nkb && gr < kw && gc < kw && gc <= gr) v[b][q] = F[(C0 + gr) + (size_t)ldp * (C0 + gc)];
            }
        }
#pragma unroll
        for (int b = 0; b < ND; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = tid + 256 * q;
                if (b < nD) D[b][idx % NB][idx / NB] = v[b][q];
            }
    }
    // (no zero fill of S: every sub-block on or below the block diagonal is written in full below, and nothing above it
    //  or beyond the kw columns is ever read)
    STAMP(0);
    __syncthreads();
    STAMP(1);
    for (int j = 0; j < nkb; ++j) {
        blk32& Djj = D[j * (j + 1) / 2 + j];
        const int wb = min(NB, kw - NB * j);
        STAMP(2 + 4 * j);
        if (wv == 0) {
            double a[NB];
#pragma unroll
            for (int c = 0; c < NB; ++c) a[c] = (lane < NB) ? Djj[lane][c] : (c == lane - NB ? 1.0 : 0.0);
            // LDL^T elimination: no square root on the pivot chain (4.9 against ~6.5 us per block, scripts/micro/lat_micro.hip); the rows of
            // the inverse take 1 / sqrt(d_row) afterwards (the broadcast reads lanes < 32 and stays outside the divergent store)
            int bad;
            if (LDL) {
                double rs;
                bad = ldl32_inverse<REPAIR>(a, wb, lane, &rs);
#pragma unroll
                for (int r = 0; r < NB; ++r) a[r] *= rl(rs, r);
            } else {
                bad = chol32_inverse(a, wb, lane);
            }
            if (lane >= NB) {
                const int cl = lane - NB;               // this lane holds column cl of the inverse
                double* Li = fd.Linv + fd.linvoff[t] + (size_t)(C0 / NB + j) * NB * NB;
#pragma unroll
                for (int r = 0; r < NB; ++r) {
                    const double v = (cl < wb && r < wb && cl <= r) ? a[r] : 0.0;
                    Djj[r][cl] = v;                     // the diagonal sub-block now holds Linv_j
                    Li[r + NB * cl] = v;
                }
                if (cl == 0 && bad) atomicAdd(info, bad);
            }
        }
        __syncthreads();
        STAMP(3 + 4 * j);
        if (j + 1 < nkb) {
            // L_ij = D_ij Linv_j^T for the sub-blocks below
            mfma_d4 acc[NBLK > 1 ? NBLK - 1 : 1];
#pragma unroll
            for (int i = 1; i < NBLK; ++i) {
                acc[i - 1] = (mfma_d4){0.0, 0.0, 0.0, 0.0};
                if (j + i < nkb) mfma_blk<false, true>(acc[i - 1], D[(j + i) * (j + i + 1) / 2 + j], Djj, si, sj, l15, l4);
            }
            __syncthreads();
#pragma unroll
            for (int i = 1; i < NBLK; ++i) {
                if (j + i >= nkb) continue;
                blk32& Dij = D[(j + i) * (j + i + 1) / 2 + j];
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int r = 16 * si + l4 + 4 * reg, c = 16 * sj + l15;
                    Dij[r][c] = acc[i - 1][reg];
                    const int gr = NB * (j + i) + r, gc = NB * j + c;
                    if (gr < kw && gc < kw) F[(C0 + gr) + (size_t)ldp * (C0 + gc)] = acc[i - 1][reg];
                }
            }
            __syncthreads();
            STAMP(4 + 4 * j);
            // D_ik -= L_ij L_kj^T, i >= k > j (each wave updates its own sub-block of every D_ik)
            for (int i = j + 1; i < nkb; ++i)
                for (int k = j + 1; k <= i; ++k) {
                    mfma_d4 u = (mfma_d4){0.0, 0.0, 0.0, 0.0};
                    mfma_blk<false, true>(u, D[i * (i + 1) / 2 + j], D[k * (k + 1) / 2 + j], si, sj, l15, l4);
                    blk32& Dik = D[i * (i + 1) / 2 + k];
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) Dik[16 * si + l4 + 4 * reg][16 * sj + l15] -= u[reg];
                }
            __syncthreads();
        }
    }
    // inverse of the kw x kw factor, block column by block column
    STAMP(20);
    for (int j = 0; j < nkb; ++j) {
        const blk32& Sjj = D[j * (j + 1) / 2 + j];
        if (!fuse_rows)
        for (int idx = tid; idx < NB * NB; idx += 256) {
            const int r = idx % NB, c = idx / NB;
            Sout[(NB * j + r) + (size_t)lds_ * (NB * j + c)] = Sjj[r][c];
        }
        for (int i = j + 1; i < nkb; ++i) {
            // W = sum_{k=j}^{i-1} L_ik S_kj: S_kj (j < k < i) already sits where L_kj was; L_ij is read here for the last time
            blk32& Dij = D[i * (i + 1) / 2 + j];
            mfma_d4 w = (mfma_d4){0.0, 0.0, 0.0, 0.0};
            for (int k = j; k < i; ++k)
                mfma_blk<false, false>(w, D[i * (i + 1) / 2 + k], D[k * (k + 1) / 2 + j], si, sj, l15, l4);
            __syncthreads();                                   // every wave has read L_ij: its sub-block takes W
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) Dij[16 * si + l4 + 4 * reg][16 * sj + l15] = w[reg];
            __syncthreads();
            // S_ij = -Linv_i W
            mfma_d4 x = (mfma_d4){0.0, 0.0, 0.0, 0.0};
            mfma_blk<false, false>(x, D[i * (i + 1) / 2 + i], Dij, si, sj, l15, l4);
            __syncthreads();                                   // every wave has read W: the sub-block takes S_ij
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = 16 * si + l4 + 4 * reg, c = 16 * sj + l15;
                Dij[r][c] = -x[reg];
                if (!fuse_rows) Sout[(NB * i + r) + (size_t)lds_ * (NB * j + c)] = -x[reg];
            }
            __syncthreads();
        }
    }
    STAMP(21);
    if (fuse_rows) {
        // L[r][C0 + c] = sum_{k <= c} A[r][C0 + k] S[c][k] for the rows below the block (k_panel_rows' product): a wave takes 16 rows
        // at a time, their kw entries in registers as the MFMA B operand, S[c][k] from the sub-blocks in LDS as the A operand
        __syncthreads();
        const int nf = fd.nf[t];
        for (int row0 = C0 + kw + 16 * wv; row0 < nf; row0 += 64) {
            const int row = row0 + l15;
            const bool rok = row < nf;
            double a[NBLK * 8];
#pragma unroll
            for (int kk = 0; kk < NBLK * 8; ++kk) {
                const int k = 4 * kk + l4;
                a[kk] = (rok && k < kw) ? F[row + (size_t)ldp * (C0 + k)] : 0.0;
            }
#pragma unroll
            for (int cb = 0; cb < NBLK * 2; ++cb) {
                if (16 * cb >= kw) break;
                const int bi = cb >> 1;
                mfma_d4 acc = (mfma_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 4 * cb + 4; ++kk)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(D[bi * (bi + 1) / 2 + (kk >> 3)][16 * (cb & 1) + l15][4 * (kk & 7) + l4], a[kk], acc, 0, 0, 0);
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int c = 16 * cb + l4 + 4 * reg;
                    if (rok && c < kw) F[row + (size_t)ldp * (C0 + c)] = acc[reg];
                }
            }
        }
    }
}


// ---- k_diag_block2: the same block, the same outputs, scheduled for latency.
// k_diag_block runs its phases one after the other on ONE workgroup: four register factorisations on wave 0 (three waves
// idle), then the MFMA phases with wave 0 idle in between, then the inverse S -- 45 us per block, and the chain of these
// blocks IS the factorisation time at the top of the tree (0.018 of fp64 peak).  Here:
//   * the 32 x 32 factor + inverse is the LDL^T elimination (ldl32_inverse: no square root on the pivot chain);
//   * while wave 0 factorises D_jj, waves 1..3 work through whole 32 x 32 x 32 block products that do not depend on it:
//     the rank-32 updates of the previous block column that the NEXT factorisation does not need, and the rows of the inverse
//     S that are already determined;
//   * between two factorisations only what the next one needs runs on all four waves (by 16 x 16 quadrant): the block column
//     L_ij = D_ij Linv_j^T and the update of D_(j+1)(j+1);
//   * after the last factorisation one product per remaining block of S.
// LDS: D (lower block triangle), the off-diagonal blocks of S, one scratch block per wave 1..3.
__device__ __host__ inline int diag_block2_lds_blocks(int nblk) { return nblk * (nblk + 1) / 2 + nblk * (nblk - 1) / 2 + (nblk > 1 ? 3 : 0); }

// acc (2 x 2 quadrants, D layout) += A B for whole 32 x 32 blocks, one wave
template <bool TA, bool TB>
__device__ __forceinline__ void wave_mm(mfma_d4 (&acc)[2][2], const blk32& A, const blk32& B, int l15, int l4) {
#pragma unroll
    for (int kk = 0; kk < NB; kk += 4) {
        const double a0 = TA ? A[kk + l4][l15] : A[l15][kk + l4], a1 = TA ? A[kk + l4][16 + l15] : A[16 + l15][kk + l4];
        const double b0 = TB ? B[l15][kk + l4] : B[kk + l4][l15], b1 = TB ? B[16 + l15][kk + l4] : B[kk + l4][16 + l15];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
}
__device__ __forceinline__ void wave_zero(mfma_d4 (&acc)[2][2]) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (mfma_d4){0.0, 0.0, 0.0, 0.0};
}
// block (LDS) = sign * acc, and optionally the same values to global memory (column-major, leading dimension ldg)
template <bool NEG, bool SUB>
__device__ __forceinline__ void wave_store(blk32& dst, const mfma_d4 (&acc)[2][2], int l15, int l4, double* g = nullptr, size_t ldg = 0) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = 16 * a + l4 + 4 * reg, c = 16 * b + l15;
                const double v = NEG ? -acc[a][b][reg] : acc[a][b][reg];
                if (SUB) dst[r][c] -= v; else dst[r][c] = v;
                if (g) g[r + ldg * c] = v;
            }
}

template <bool REPAIR>
__global__ void __launch_bounds__(256)
k_diag_block2(FrontDev fd, const int* __restrict__ level_nodes, int first, int nblk, int C0, double* __restrict__ Swork,
              int* __restrict__ info) {
    STAMP(31);
    const int slot = first + blockIdx.x;
    const int t = level_nodes[slot];
    const int np = fd.npiv[t];
    if (C0 >= np) return;
    const int kw = min(NBO, np - C0);
    const int nkb = (kw + NB - 1) / NB;
    const int ldp = ldp_of(fd.nf[t]);
    double* F = fd.P + fd.poff[t];
    const int lds_ = Swork ? SPD : ldx_of(np);
    double* Sout = Swork ? Swork + (size_t)slot * SPD * SPD : fd.X + fd.xoff[t] + C0 + (size_t)lds_ * C0;
    extern __shared__ double lds_raw[];
    const int nD = nblk * (nblk + 1) / 2;
    blk32* D = reinterpret_cast<blk32*>(lds_raw);              // D_ij, i >= j, at i (i + 1) / 2 + j; D_jj becomes Linv_j = S_jj
    blk32* Sb = D + nD;                                         // S_ij, i > j, at i (i - 1) / 2 + j
    blk32* Sc = Sb + nblk * (nblk - 1) / 2;                      // scratch block of wave w at Sc[w - 1]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int si = wv & 1, sj = wv >> 1, l15 = lane & 15, l4 = lane >> 4;
#define DB(i, j) D[(i) * ((i) + 1) / 2 + (j)]
#define SB(i, j) Sb[(i) * ((i) - 1) / 2 + (j)]
#define SOUT(i, j) (Sout + NB * (i) + (size_t)lds_ * (NB * (j)))
    // Load, identity padding beyond kw.  D_00 is needed by the first factorisation only: wave 0 reads its rows straight
    // into registers and factorises them while the other blocks are still on their way to LDS (nobody else touches D[0]).
    double a0[NB];
    if (wv == 0) {
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            a0[c] = (lane < NB) ? (lane == c ? 1.0 : 0.0) : (c == lane - NB ? 1.0 : 0.0);
            if (lane < NB && c <= lane && lane < kw) a0[c] = F[(C0 + lane) + (size_t)ldp * (C0 + c)];
        }
    }
    if (wv > 0) {
        // the other nine blocks: waves 1..3, 192 threads, all loads of a thread in flight before the first store
        constexpr int NL = (9 * NB * NB + 191) / 192;           // 48
        const int t3 = tid - 64;
        double v[NL];
#pragma unroll
        for (int q = 0; q < NL; ++q) {
            const int e = t3 + 192 * q;
            const int b = 1 + e / (NB * NB), idx = e % (NB * NB);
            const int bi = b < 3 ? 1 : b < 6 ? 2 : 3;
            const int bj = b - bi * (bi + 1) / 2;
            const int r = idx % NB, c = idx / NB;
            const int gr = NB * bi + r, gc = NB * bj + c;
            v[q] = (gr == gc) ? 1.0 : 0.0;
            if (b < nD && bi < nkb && gr < kw && gc < kw && gc <= gr) v[q] = F[(C0 + gr) + (size_t)ldp * (C0 + gc)];
        }
#pragma unroll
        for (int q = 0; q < NL; ++q) {
            const int e = t3 + 192 * q;
            const int b = 1 + e / (NB * NB), idx = e % (NB * NB);
            if (b < nD) D[b][idx % NB][idx / NB] = v[q];
        }
    }
    STAMP(0);
    // block tasks of one wave.  L_ij (in place of D_ij) also goes to the front's factor columns.
    auto factor_rows = [&](int i, int j) {                       // T(i, j): D_ij <- L_ij = D_ij Linv_j^T
        mfma_d4 acc[2][2];
        wave_zero(acc);
        wave_mm<false, true>(acc, DB(i, j), DB(j, j), l15, l4);
        blk32& Dij = DB(i, j);
#pragma unroll
        for (int qa = 0; qa < 2; ++qa)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int r = 16 * qa + l4 + 4 * reg, c = 16 * qb + l15;
                    Dij[r][c] = acc[qa][qb][reg];
                    const int gr = NB * i + r, gc = NB * j + c;
                    if (gr < kw && gc < kw) F[(C0 + gr) + (size_t)ldp * (C0 + gc)] = acc[qa][qb][reg];
                }
        __builtin_amdgcn_wave_barrier();
    };
    auto update = [&](int i, int k, int j) {                    // U(i, k; j): D_ik -= L_ij L_kj^T
        mfma_d4 acc[2][2];
        wave_zero(acc);
        wave_mm<false, true>(acc, DB(i, j), DB(k, j), l15, l4);
        wave_store<false, true>(DB(i, k), acc, l15, l4);
        __builtin_amdgcn_wave_barrier();
    };
    auto s_block = [&](int i, int j, blk32& W) {                // S_ij = -Linv_i W  -> LDS and global
        mfma_d4 acc[2][2];
        wave_zero(acc);
        wave_mm<false, false>(acc, DB(i, i), W, l15, l4);
        wave_store<true, false>(SB(i, j), acc, l15, l4, SOUT(i, j), (size_t)lds_);
        __builtin_amdgcn_wave_barrier();
    };
    auto s_of = [&](int k, int j) -> const blk32& { return k == j ? DB(j, j) : SB(k, j); };
    auto w_block = [&](int i, int j, blk32& W) {                // W = sum_{k=j}^{i-1} L_ik S_kj
        mfma_d4 acc[2][2];
        wave_zero(acc);
        for (int k = j; k < i; ++k) wave_mm<false, false>(acc, DB(i, k), s_of(k, j), l15, l4);
        wave_store<false, false>(W, acc, l15, l4);
        __builtin_amdgcn_wave_barrier();
    };
    const int last = nkb - 1;
    for (int j = 0; j < nkb; ++j) {
        STAMP(2 + 4 * j);
        if (wv == 0) {
            // ---- F(j): factor + inverse of D_jj in registers; Linv_j replaces D_jj and goes to Linv
            blk32& Djj = DB(j, j);
            const int wb = min(NB, kw - NB * j);
            double a[NB];
#pragma unroll
            for (int c = 0; c < NB; ++c) a[c] = j == 0 ? a0[c] : ((lane < NB) ? Djj[lane][c] : (c == lane - NB ? 1.0 : 0.0));
            double rs;
            const int bad = ldl32_inverse<REPAIR>(a, wb, lane, &rs);
            // rows of the inverse take 1 / sqrt(d_row): the broadcast reads lanes < 32 and must stay outside the divergent store
#pragma unroll
            for (int r = 0; r < NB; ++r) a[r] *= rl(rs, r);
            if (lane >= NB) {
                const int cl = lane - NB;
                double* Li = fd.Linv + fd.linvoff[t] + (size_t)(C0 / NB + j) * NB * NB;
#pragma unroll
                for (int r = 0; r < NB; ++r) {
                    const double v = (cl < wb && r < wb && cl <= r) ? a[r] : 0.0;
                    Djj[r][cl] = v;
                    Li[r + NB * cl] = v;
                }
                if (cl == 0 && bad) atomicAdd(info, bad);
            }
        } else if (j == 1) {
            // ---- beside F(1): block column 0 below row 1 -- its factor rows and the updates they feed.  U(3, 2; 0) needs the
            // rows of both waves and waits for the next stage (D_32 is first read between F(2) and F(3)).
            if (wv == 1 && nkb > 2) { factor_rows(2, 0); update(2, 1, 0); update(2, 2, 0); }
            if (wv == 3 && nkb > 3) { factor_rows(3, 0); update(3, 1, 0); update(3, 3, 0); }
            if (nkb == 2 && wv == 2) w_block(1, 0, Sc[1]);      // last row: W_10 = L_10 S_00
        } else if (j == 2) {
            if (wv == 1 && nkb > 3) { factor_rows(3, 1); update(3, 2, 0); update(3, 2, 1); update(3, 3, 1); }
            if (wv == 3) {                                      // row 1 of S; and the last row's W if this is the last stage
                w_block(1, 0, Sc[2]);
                s_block(1, 0, Sc[2]);
                if (nkb == 3) w_block(2, 0, Sc[2]);             // W_20 = L_20 S_00 + L_21 S_10
            }
            if (nkb == 3 && wv == 2) w_block(2, 1, Sc[1]);      // W_21 = L_21 S_11
        } else if (j == 3) {
            if (wv == 1) { w_block(2, 0, Sc[0]); s_block(2, 0, Sc[0]); w_block(3, 0, Sc[0]); }
            if (wv == 2) { w_block(2, 1, Sc[1]); s_block(2, 1, Sc[1]); w_block(3, 1, Sc[1]); }
            if (wv == 3) { w_block(3, 2, Sc[2]); }
        }
        __syncthreads();
        STAMP(3 + 4 * j);
        {   // S_jj = Linv_j to its place (all threads; stores only)
            const blk32& Sjj = DB(j, j);
            for (int idx = tid; idx < NB * NB; idx += 256) {
                const int r = idx % NB, c = idx / NB;
                SOUT(j, j)[r + (size_t)lds_ * c] = Sjj[r][c];
            }
        }
        if (j + 1 < nkb) {
            // ---- between two factorisations, on all four waves by quadrant: only what F(j + 1) needs --
            // L_(j+1)j = D_(j+1)j Linv_j^T, then D_(j+1)(j+1) -= L_(j+1)j L_(j+1)j^T
            mfma_d4 acc = (mfma_d4){0.0, 0.0, 0.0, 0.0};
            mfma_blk<false, true>(acc, DB(j + 1, j), DB(j, j), si, sj, l15, l4);
            __syncthreads();
            {
                blk32& Dij = DB(j + 1, j);
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int r = 16 * si + l4 + 4 * reg, c = 16 * sj + l15;
                    Dij[r][c] = acc[reg];
                    const int gr = NB * (j + 1) + r, gc = NB * j + c;
                    if (gr < kw && gc < kw) F[(C0 + gr) + (size_t)ldp * (C0 + gc)] = acc[reg];
                }
            }
            __syncthreads();
            mfma_d4 u = (mfma_d4){0.0, 0.0, 0.0, 0.0};
            mfma_blk<false, true>(u, DB(j + 1, j), DB(j + 1, j), si, sj, l15, l4);
            blk32& Dn = DB(j + 1, j + 1);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) Dn[16 * si + l4 + 4 * reg][16 * sj + l15] -= u[reg];
            __syncthreads();
            STAMP(4 + 4 * j);
        }
    }
    STAMP(20);
    // ---- after the last factorisation: S_(last)(j) = -Linv_last W_(last)(j), by quadrant on all four waves
    int wsrc[3] = {0, 0, 0};                                    // scratch block that holds W_(last)(j)
    if (nkb == 2) wsrc[0] = 1;
    if (nkb == 3) { wsrc[0] = 2; wsrc[1] = 1; }
    if (nkb == 4) { wsrc[0] = 0; wsrc[1] = 1; wsrc[2] = 2; }
    for (int j = 0; j < last; ++j) {
        mfma_d4 x = (mfma_d4){0.0, 0.0, 0.0, 0.0};
        mfma_blk<false, false>(x, DB(last, last), Sc[wsrc[j]], si, sj, l15, l4);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int r = 16 * si + l4 + 4 * reg, c = 16 * sj + l15;
            SOUT(last, j)[r + (size_t)lds_ * c] = -x[reg];
        }
    }
    STAMP(21);
#undef DB
#undef SB
#undef SOUT
}

// PB: rows below the diagonal block of the outer panel: L[r][C0 + c] = sum_{k <= c} A[r][C0 + k] S[c][k] -- one GEMM
// per 64-row tile against the inverse S of the diagonal block's factor (k_diag_block).  A wave owns 16 rows: their
// kw <= 128 entries sit in registers as the MFMA B operand (read once, so the result can overwrite them in place),
// S streams from L2 as the A operand, and the product is formed transposed so that the stores run along the
// columns of the column-major front.
__global__ void __launch_bounds__(256, 4)      // four waves per SIMD (measured: -15 % against three, 20 B of scratch)
k_panel_rows(FrontDev fd, const int* __restrict__ level_nodes, int first, int C0, const double* __restrict__ Swork, int tile_first) {
    const int slot = first + blockIdx.y;                       // position of the front in its level
    const int t = level_nodes[slot];
    const int np = fd.npiv[t];
    if (C0 >= np) return;
    const int kw = min(NBO, np - C0);
    const int nf = fd.nf[t];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int tile0 = C0 + kw + (blockIdx.x + tile_first) * TS;     // tile_first: a launch may take the row tiles from there on only
    if (tile0 >= nf) return;
    const int row0 = tile0 + 16 * wv;
    double* F = fd.P + fd.poff[t];                             // pivot columns only
    const int ldp = ldp_of(nf);
    const int lds_ = Swork ? SPD : ldx_of(np);
    const double* S = Swork ? Swork + (size_t)slot * SPD * SPD : fd.X + fd.xoff[t] + C0 + (size_t)lds_ * C0;
    // S is shared by the four waves: the 16 rows of S that produce output columns [16 cb, 16 cb + 16) are staged in
    // LDS (k-major, so that the MFMA A operand S[c][k] is a conflict-free read), double-buffered over cb
    __shared__ double sb[2][NBO][16];
    const int row = row0 + l15;
    const bool rok = row < nf;
    double a[NBO / 4];
#pragma unroll
    for (int kk = 0; kk < NBO / 4; ++kk) {
        const int k = 4 * kk + l4;
        a[kk] = (rok && k < kw) ? F[row + (size_t)ldp * (C0 + k)] : 0.0;
    }
    const int sc = tid & 15, sk0 = tid >> 4;              // staging: column sc of the block, rows sk0, sk0 + 16, ...
    double pre[NBO / 16];
    // stage block 0
#pragma unroll
    for (int q = 0; q < 1; ++q) pre[q] = S[sc + (size_t)lds_ * (sk0 + 16 * q)];
    sb[0][sk0][sc] = pre[0];
    __syncthreads();
#pragma unroll
    for (int cb = 0; cb < NBO / 16; ++cb) {
        if (16 * cb >= kw) break;
        const int cur = cb & 1;
        const bool more = cb + 1 < NBO / 16 && 16 * (cb + 1) < kw;
        if (more) {
#pragma unroll
            for (int q = 0; q < cb + 2; ++q) pre[q] = S[(16 * (cb + 1) + sc) + (size_t)lds_ * (sk0 + 16 * q)];
        }
        mfma_d4 acc = (mfma_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4 * cb + 4; ++kk)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sb[cur][4 * kk + l4][l15], a[kk], acc, 0, 0, 0);   // D[i = column][j = row]
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int c = 16 * cb + l4 + 4 * reg;
            if (rok && c < kw) F[row + (size_t)ldp * (C0 + c)] = acc[reg];
        }
        if (more) {
#pragma unroll
            for (int q = 0; q < cb + 2; ++q) sb[cur ^ 1][sk0 + 16 * q][sc] = pre[q];
        }
        __syncthreads();
    }
}

// The same product for levels of few fronts, where a launch lasts as long as ONE workgroup does: k_panel_rows exposes the
// latency of a load of S from L2 eight times (once per 16 output columns; the products between two loads are too short to
// cover it).  Here all 36 lower-triangular 16 x 16 blocks of S travel to LDS in one burst (72 KB, two workgroups per CU),
// and the 144 MFMAs of a wave follow without a barrier or a global load between them.
constexpr int PANEL_ROWS_PRELOAD_LDS = (NBO / 16) * (NBO / 16 + 1) / 2 * 256 * (int)sizeof(double);
__global__ void __launch_bounds__(256, 2)
k_panel_rows_preload(FrontDev fd, const int* __restrict__ level_nodes, int first, int C0, const double* __restrict__ Swork, int tile_first) {
    extern __shared__ double sall[];                           // block (cb, kb <= cb) at cb (cb + 1) / 2 + kb: [k][c], 16 x 16
    const int slot = first + blockIdx.y;
    const int t = level_nodes[slot];
    const int np = fd.npiv[t];
    if (C0 >= np) return;
    const int kw = min(NBO, np - C0);
    const int nf = fd.nf[t];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int tile0 = C0 + kw + (blockIdx.x + tile_first) * TS;
    if (tile0 >= nf) return;
    const int row0 = tile0 + 16 * wv;
    double* F = fd.P + fd.poff[t];
    const int ldp = ldp_of(nf);
    const int lds_ = Swork ? SPD : ldx_of(np);
    const double* S = Swork ? Swork + (size_t)slot * SPD * SPD : fd.X + fd.xoff[t] + C0 + (size_t)lds_ * C0;
    const int sc = tid & 15, sk = tid >> 4;                    // this thread's entry of every block: column sc, row sk
    constexpr int NCB = NBO / 16, NBLK = NCB * (NCB + 1) / 2;
    const int ncb = (kw + 15) / 16;                            // block rows this panel has (the same S entries k_panel_rows reads)
    double pre[NBLK];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int kb = 0; kb <= cb; ++kb)
            pre[cb * (cb + 1) / 2 + kb] = cb < ncb ? S[(16 * cb + sc) + (size_t)lds_ * (16 * kb + sk)] : 0.0;
    const int row = row0 + l15;
    const bool rok = row < nf;
    double a[NBO / 4];
#pragma unroll
    for (int kk = 0; kk < NBO / 4; ++kk) {
        const int k = 4 * kk + l4;
        a[kk] = (rok && k < kw) ? F[row + (size_t)ldp * (C0 + k)] : 0.0;
    }
#pragma unroll
    for (int b = 0; b < NBLK; ++b) sall[b * 256 + sk * 16 + sc] = pre[b];
    __syncthreads();
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        if (cb >= ncb) break;
        mfma_d4 acc = (mfma_d4){0.0, 0.0, 0.0, 0.0};
        const double* blk = sall + cb * (cb + 1) / 2 * 256;
#pragma unroll
        for (int kk = 0; kk < 4 * cb + 4; ++kk)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(blk[(kk >> 2) * 256 + (4 * (kk & 3) + l4) * 16 + l15], a[kk], acc, 0, 0, 0);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int c = 16 * cb + l4 + 4 * reg;
            if (rok && c < kw) F[row + (size_t)ldp * (C0 + c)] = acc[reg];
        }
    }
}

// The same product cut finer, for launches that do not fill the chip (the top of the tree: a launch lasts as long as ONE workgroup
// does, and k_panel_rows gives a wave 144 dependent MFMAs behind eight staged loads of S).  A workgroup takes 16 rows; wave w
// forms their output column blocks w and 7 - w (4 w + 4 and 32 - 4 w MFMAs: 36 per wave, a quarter of k_panel_rows' chain).  No
// LDS: the MFMA A operand S[c][k] (16 consecutive c per k: one 128-byte segment) and the rows travel straight to registers, all
// loads of a wave in flight at once.  The product overwrites the rows in place and the four waves share them, so every wave
// has ALL its loads back before any wave stores (the barrier).
__global__ void __launch_bounds__(256, 2)
k_panel_rows_fine(FrontDev fd, const int* __restrict__ level_nodes, int first, int C0, const double* __restrict__ Swork, int tile_first) {
    const int slot = first + blockIdx.y;
    const int t = level_nodes[slot];
    const int np = fd.npiv[t];
    if (C0 >= np) return;                                     // (uniform per workgroup: no barrier is skipped by part of it)
    const int kw = min(NBO, np - C0);
    const int nf = fd.nf[t];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int row0 = C0 + kw + (blockIdx.x + 4 * tile_first) * 16;     // tile_first counts 64-row tiles, as in k_panel_rows
    if (row0 >= nf) return;
    double* F = fd.P + fd.poff[t];
    const int ldp = ldp_of(nf);
    const int lds_ = Swork ? SPD : ldx_of(np);
    const double* S = Swork ? Swork + (size_t)slot * SPD * SPD : fd.X + fd.xoff[t] + C0 + (size_t)lds_ * C0;
    constexpr int NCB = NBO / 16;
    const int ncb = (kw + 15) / 16;
    const int cbs[2] = {wv, NCB - 1 - wv};                    // this wave's two column blocks
    const int row = row0 + l15;
    const bool rok = row < nf;
    double a[NBO / 4];
#pragma unroll
    for (int kk = 0; kk < NBO / 4; ++kk) {
        const int k = 4 * kk + l4;
        a[kk] = (rok && k < kw) ? F[row + (size_t)ldp * (C0 + k)] : 0.0;
    }
    // S operands of both blocks: block cb needs k < 16 cb + 16; the second block (cb >= 4) the longer chain
    double s0[4 * (NCB / 2)], s1[4 * NCB];
#pragma unroll
    for (int kk = 0; kk < 4 * (NCB / 2); ++kk)
        s0[kk] = (cbs[0] < ncb && kk < 4 * cbs[0] + 4) ? S[(16 * cbs[0] + l15) + (size_t)lds_ * (4 * kk + l4)] : 0.0;
#pragma unroll
    for (int kk = 0; kk < 4 * NCB; ++kk)
        s1[kk] = (cbs[1] < ncb && kk < 4 * cbs[1] + 4) ? S[(16 * cbs[1] + l15) + (size_t)lds_ * (4 * kk + l4)] : 0.0;
    mfma_d4 acc0 = (mfma_d4){0.0, 0.0, 0.0, 0.0}, acc1 = (mfma_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 4 * (NCB / 2); ++kk)
        if (kk < 4 * cbs[0] + 4) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(s0[kk], a[kk], acc0, 0, 0, 0);      // D[i = column][j = row]
#pragma unroll
    for (int kk = 0; kk < 4 * NCB; ++kk)
        if (kk < 4 * cbs[1] + 4) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(s1[kk], a[kk], acc1, 0, 0, 0);
    // every load of this workgroup has been consumed by an MFMA above: the rows may be overwritten once all waves are here
    __syncthreads();
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int c0 = 16 * cbs[0] + l4 + 4 * reg, c1 = 16 * cbs[1] + l4 + 4 * reg;
        if (rok && c0 < kw) F[row + (size_t)ldp * (C0 + c0)] = acc0[reg];
        if (rok && c1 < kw) F[row + (size_t)ldp * (C0 + c1)] = acc1[reg];
    }
}

// P2: rank-k update  C[r][c] -= sum_m L[r][m] L[c][m]  of lower-triangle 64x64 tiles with v_mfma_f64_16x16x4_f64.
// Workgroup = 4 waves = one 64x64 tile of the front; wave w owns the 32x32 quarter (w&1 rows, w>>1 columns) as
// 2x2 MFMA blocks.  To make the stores run along rows of the column-major front (coalesced), the product is
// formed transposed: A[i][k] = L[c0+i][k] (tile columns), B[k][j] = L[r0+j][k] (tile rows), so D[i][j] = C[r0+j][c0+i];
// per the CDNA4 f64 layout a lane holds A[lane&15][lane>>4], B[lane>>4][lane&15] and D[(lane>>4) + 4*reg][lane&15].
// LDS rows are padded to 80 doubles so that the four 16-lane groups of a ds_read_b64 hit disjoint banks.
constexpr int LSTR = TS + 16;

// The K range and the updated columns of one front in one k_trailing_mfma launch (shared with the host's flop count).
//   schur 0  left-looking: the 128 pivot columns at C0 receive the updates of the factor columns [K0, C0) right
//            before they are factorised (K0 = 0: all earlier columns; K0 = start of the super-panel: see below);
//   schur 1  the Schur complement is updated once, after the last panel, with K = npiv.  Every entry of the front is
//            then read and written once per factorisation instead of once per 128 factor columns;
//   schur 2  right-looking: everything behind the factor columns [C0, C0 + KW) is updated with those columns
//            (KW = 128: after every outer panel; KW = 256, 512: after a super-panel whose own panels were updated
//            with schur 0 / K0 = C0 -- the Schur complement then moves through HBM once per KW columns);
//   schur 3, 4  the look-ahead split of 2: only the next 128 columns | everything behind those;
//   schur 5  as 2, but the pivot columns of the NEXT super-panel are left out (columns from min(C0 + 2 KW, npiv)): those
//            get this super-panel's columns through their own schur 0 updates (K0 = C0), so that this launch can run on
//            a second stream beside the next super-panel's chain of panels.
struct TrailRange { int kc0, kw, col_lo, col_hi; };
__device__ __host__ inline TrailRange trail_range(int schur, int C0, int K0, int KW, int np, int nf) {
    TrailRange r{0, 0, 0, 0};
    if (schur != 1 && C0 >= np) return r;
    r.kc0 = schur >= 2 ? C0 : (schur ? 0 : K0);
    r.kw = schur >= 2 ? (KW < np - C0 ? KW : np - C0) : (schur ? np : C0 - K0);
    // the look-ahead split (3 | 4) sits 128 columns behind the even anchor of the first updated column
    const int split = ((C0 + r.kw) & ~1) + NBO;
    r.col_lo = schur >= 2 ? (schur == 4 ? split : schur == 5 ? (C0 + 2 * KW < np ? C0 + 2 * KW : np) : C0 + r.kw) : (schur ? np : C0);
    r.col_hi = schur == 0 ? (C0 + NBO < np ? C0 + NBO : np) : schur == 3 ? (split < nf ? split : nf) : nf;
    return r;
}

// GATHER (levels above the leaves, the launch that touches its columns first): the tile of C does not exist yet -- it is
// the sum of the children's Schur-complement entries that land there (the extend-add of these columns, left out of
// k_extend_gather), gathered here and written once.  `mask`: strong-BC pivots get their unit diagonal, as in the extend-add.
template <bool GATHER>
__global__ void __launch_bounds__(256, 4)      // 128 registers: four waves per SIMD (measured: -6 % against three)
k_trailing_mfma(FrontDev fd, const int* __restrict__ level_nodes, int first, int C0, int schur, int K0, int KW,
                const unsigned char* __restrict__ mask, int bx_first, int tile_map) {
    // Which (front, tile) this workgroup takes.  Workgroups are dealt round-robin over the 8 XCDs (each with its own L2) in
    // launch order, x fastest; the two or three tiles of a small front read the same rows of its factor panel.  On levels
    // of many fronts (a multiple of 8, at least 256) XCD x therefore takes the fronts x, x + 8, ... whole, tile after tile:
    // the panel leaves HBM once instead of once per tile (leaves 604 -> 552 us, levels 1-5 -2..-5 %).  The map is a
    // bijection of the grid whatever the placement really is.  Levels of fewer, larger fronts keep the launch order: their
    // tiles must spread over all eight XCDs (64 fronts: +8 % with the map; 8 unequal fronts: +75 %).
    int bxv = blockIdx.x, zv = blockIdx.z;
    {
        const int G = gridDim.x, n = gridDim.z;
        if (n >= 256 && (n & 7) == 0) {
            const int flat = bxv + G * zv, x = flat & 7, q = flat >> 3;
            zv = (q / G) * 8 + x; bxv = q % G;
        }
    }
    bxv += bx_first;               // a launch may take the tiles from linear index bx_first on only (diagonal look-ahead)
    const int t = level_nodes[first + zv];
    const int np = fd.npiv[t];
    const int nf = fd.nf[t];
    const TrailRange tr = trail_range(schur, C0, K0, KW, np, nf);
    int kc0 = tr.kc0, kw = tr.kw;
    const int col_lo = tr.col_lo, col_hi = tr.col_hi;
    if (kw <= 0) return;
    // gridDim.y > 1 (never with GATHER): the K range is cut into that many slices of whole 16-column stages, one workgroup per
    // slice, and the products are ADDED to C with atomics -- for launches of few tiles at the top of the tree, where the time
    // of the launch is the time one workgroup needs to walk through its K range
    const bool ksplit = !GATHER && gridDim.y > 1;
    if (ksplit) {
        const int per = (kw + 16 * (int)gridDim.y - 1) / (16 * (int)gridDim.y) * 16;
        const int lo = (int)blockIdx.y * per;
        if (lo >= kw) return;
        kc0 += lo; kw = min(per, kw - lo);
    }
    // tile from the linear block index: consecutive workgroups go to different XCDs, so a (row tile, column tile)
    // grid whose x extent is a multiple of 8 would pin every row-tile offset to one XCD -- and the lower triangle has
    // 8x more tiles at offset 0 than at offset 7.  The linear order spreads them evenly (measured: up to 2.2x).
    int bx, by;
    if (schur == 0 || schur == 3) {
        by = bxv & 1; bx = bxv >> 1;                               // two column tiles per panel
    } else if (tile_map == 1) {
        // Few, large fronts (MFMA-bound levels): what the linear order costs there is L2 misses -- the tiles one XCD gets share
        // a row block at best, and every tile pulls its 2 x 64 rows of the K panel through that XCD's L2 (measured 2.2-3 x the
        // compulsory bytes).  Here an XCD takes 4 x 4 super-tiles whole: its workgroups 16 q .. 16 q + 15 (the grid's x extent
        // is a multiple of 128, so workgroup b of any front lands on XCD b % 8) are the tiles of super-tile 8 q + xcd, which
        // walk through K side by side and read 8 row blocks of the panel between them instead of 32.
        const int x = bxv & 7, slot = bxv >> 3;
        const int st = (slot >> 4) * 8 + x, within = slot & 15;
        int I = (int)((sqrt(8.0 * st + 1.0) - 1.0) * 0.5);
        while ((I + 1) * (I + 2) / 2 <= st) ++I;
        while (I * (I + 1) / 2 > st) --I;
        const int J = st - I * (I + 1) / 2;
        const int rt = 4 * I + (within & 3), ct = 4 * J + (within >> 2);
        if (rt < ct) return;
        by = ct; bx = rt - ct;
    } else {
        const int lin = bxv;
        int ti = (int)((sqrt(8.0 * lin + 1.0) - 1.0) * 0.5);
        while ((ti + 1) * (ti + 2) / 2 <= lin) ++ti;
        while (ti * (ti + 1) / 2 > lin) --ti;
        by = lin - ti * (ti + 1) / 2; bx = ti - by;                 // column tile by, row tile by + bx
    }
    // tiles are anchored at an even column: with the even leading dimension of the panel store every pair of rows
    // (2 rp, 2 rp + 1) of a tile is then one aligned 16-byte load
    const int cj = (col_lo & ~1) + by * TS;
    if (cj >= col_hi) return;
    const int ri = cj + bx * TS;                   // row tiles start at the column tile (lower triangle)
    if (ri >= nf) return;
    const FrontView fv = front_view(fd, t);
    // 16 factor columns per stage, two LDS buffers (20 KB each): the next stage travels global -> registers while the
    // matrix cores work on the current one, and one barrier per stage suffices
    constexpr int KC = 16;
    __shared__ __attribute__((aligned(16))) double si[2][KC][LSTR];   // rows of the tile:    si[.][k][r] = L[ri + r][kc0 + k0 + k]
    __shared__ __attribute__((aligned(16))) double sj[2][KC][LSTR];   // columns of the tile: sj[.][k][c] = L[cj + c][kc0 + k0 + k]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = (wv & 1) * 32, wc = (wv >> 1) * 32;     // this wave's quarter: rows wr.., columns wc..
    const int l15 = lane & 15, l4 = lane >> 4;
    // Staging with as little vector arithmetic as possible -- the vector ALU shares its issue slots with the matrix
    // cores, and address / mask arithmetic per load costs this kernel a fifth of its rate (micro-benchmark: 49 -> 60
    // TFLOP/s at K = 512, 32 -> 41 at K = 128).  The stage pointer is uniform; a thread keeps four constant 32-bit byte
    // offsets (rows 2 rp, 2 rp + 1 of the tile's row and column blocks, factor columns cg and cg + 8).  Rows past the
    // front are clamped to its last row pair (their products are never stored); only the last, partial stage is masked.
    const int rp = tid & 31, cg = tid >> 5;
    const int ldp = ldp_of(nf);
    const int rowi = min(ri + 2 * rp, ldp - 2), rowj = min(cj + 2 * rp, ldp - 2);
    const unsigned obi0 = 8u * (unsigned)(rowi + ldp * cg), obi1 = 8u * (unsigned)(rowi + ldp * (cg + 8));
    const unsigned obj0 = 8u * (unsigned)(rowj + ldp * cg), obj1 = 8u * (unsigned)(rowj + ldp * (cg + 8));
    const char* base = reinterpret_cast<const char*>(fv.P + (size_t)ldp * kc0);      // the K panel: always pivot columns
    const size_t stage_bytes = (size_t)ldp * KC * 8;
    d2 pi[2], pj[2];
    auto fetch = [&](const char* b) {
        pi[0] = *reinterpret_cast<const d2*>(b + obi0); pi[1] = *reinterpret_cast<const d2*>(b + obi1);
        pj[0] = *reinterpret_cast<const d2*>(b + obj0); pj[1] = *reinterpret_cast<const d2*>(b + obj1);
    };
    auto fetch_tail = [&](const char* b, int left) {       // left = factor columns this stage still has (1..15)
        const d2 z = {0.0, 0.0};
        const unsigned a0 = 8u * (unsigned)(ldp * min(cg, left - 1)), a1 = 8u * (unsigned)(ldp * min(cg + 8, left - 1));
        const d2 vi0 = *reinterpret_cast<const d2*>(b + 8u * (unsigned)rowi + a0), vi1 = *reinterpret_cast<const d2*>(b + 8u * (unsigned)rowi + a1);
        const d2 vj0 = *reinterpret_cast<const d2*>(b + 8u * (unsigned)rowj + a0), vj1 = *reinterpret_cast<const d2*>(b + 8u * (unsigned)rowj + a1);
        pi[0] = cg < left ? vi0 : z; pi[1] = cg + 8 < left ? vi1 : z;
        pj[0] = cg < left ? vj0 : z; pj[1] = cg + 8 < left ? vj1 : z;
    };
    if (kw >= KC) fetch(base); else fetch_tail(base, kw);
    base += stage_bytes;
    mfma_d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (mfma_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        *reinterpret_cast<d2*>(&si[0][cg + 8 * q][2 * rp]) = pi[q];
        *reinterpret_cast<d2*>(&sj[0][cg + 8 * q][2 * rp]) = pj[q];
    }
    __syncthreads();
    int cur = 0;
    // a tile on the diagonal of the front (row tile == column tile): the wave whose quarter lies entirely ABOVE the diagonal (rows 0..31,
    // columns 32..63) has nothing to store -- it only helps with the staging.  On the levels of small fronts a third to a half of the
    // tiles are diagonal ones (a Schur block of 174 rows: 3 of 6), so this is 8-12 % of their MFMA work, which is what those levels
    // spend about half of their time on (the tile-quantised work is ~2.5 x the algorithmic flops there).
    const bool idle_quarter = ri == cj && wr + 31 < wc;
    for (int k0 = 0; k0 < kw; k0 += KC) {
        const int left = kw - k0 - KC;             // factor columns behind this stage
        if (left > 0) {
            if (left >= KC) fetch(base); else fetch_tail(base, left);
            base += stage_bytes;
        }
        if (!idle_quarter) {
#pragma unroll
        for (int kk = 0; kk < KC; kk += 4) {
            // A: tile columns (index i), B: tile rows (index j); k = kk + (lane >> 4)
            const double a0 = sj[cur][kk + l4][wc + l15], a1 = sj[cur][kk + l4][wc + 16 + l15];
            const double b0 = si[cur][kk + l4][wr + l15], b1 = si[cur][kk + l4][wr + 16 + l15];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
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
    if (idle_quarter) return;                      // (behind the last barrier of the K loop: nothing of this quarter is stored)
    // C -= D with all of the tile's loads in flight at once (entries outside the tile's part of the lower triangle load
    // from a safe address and are not stored).  D[i][j]: i = l4 + 4*reg -> tile column, j = l15 -> tile row
    bool cok[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int cc = cj + wc + 16 * a + l4 + 4 * reg;
            cok[a][reg] = cc >= col_lo && cc < col_hi;
        }
    auto col_ptr = [&](int a, int reg) {                       // safe: [r] below stays inside the panel store
        const int cc = cj + wc + 16 * a + l4 + 4 * reg;
        return cok[a][reg] ? fv.col(cc) : fv.P - (nf - 1);
    };
    double cv[2][2][4];
    if (ksplit) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int cc = cj + wc + 16 * a + l4 + 4 * reg;
                    const int r = ri + wr + 16 * b + l15;
                    if (cok[a][reg] && r < nf && r >= cc) atomicAdd(col_ptr(a, reg) + r, -acc[a][b][reg]);
                }
        return;
    }
    if (!GATHER) {
        double* cp[2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) cp[a][reg] = col_ptr(a, reg);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int cc = cj + wc + 16 * a + l4 + 4 * reg;
                    const int r = ri + wr + 16 * b + l15;
                    const bool ok = cok[a][reg] && r < nf && r >= cc;
                    cv[a][b][reg] = cp[a][reg][ok ? r : nf - 1];
                }
    } else {
        // rows of the two children's fronts that land on this lane's two rows and eight columns (-1: none): boundary rows
        // of the children (>= their pivot count), so every entry sits in a child's Schur block S_c[(hi - np_c) + nb_c (lo - np_c)]
        const long long dp = fd.doff[t];
        int rr[2][2], cr[2][2][4];
        const double* Sc[2];
        int npc[2], nbc[2];
#pragma unroll
        for (int sd = 0; sd < 2; ++sd) {
            const int ch = fd.child[sd][t];
            const int chs = ch >= 0 ? ch : t;
            npc[sd] = fd.npiv[chs]; nbc[sd] = fd.nf[chs] - npc[sd];
            Sc[sd] = fd.S + fd.soff[chs];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int r = ri + wr + 16 * b + l15;
                rr[sd][b] = fd.cinv[sd][dp + min(r, nf - 1)];
                if (ch < 0 || r >= nf) rr[sd][b] = -1;
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int cc = cj + wc + 16 * a + l4 + 4 * reg;
                    cr[sd][a][reg] = fd.cinv[sd][dp + min(cc, nf - 1)];
                    if (ch < 0 || !cok[a][reg]) cr[sd][a][reg] = -1;
                }
        }
        double g0[2][2][4], g1[2][2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int cc = cj + wc + 16 * a + l4 + 4 * reg;
                    const int r = ri + wr + 16 * b + l15;
                    const bool tri = r >= cc;
                    {
                        const int x = rr[0][b], y = cr[0][a][reg];
                        const bool ok = tri && x >= 0 && y >= 0;
                        const int lo = min(x, y) - npc[0], hi = max(x, y) - npc[0];
                        g0[a][b][reg] = Sc[0][ok ? hi + (size_t)nbc[0] * lo : 0];
                        if (!ok) g0[a][b][reg] = 0.0;
                    }
                    {
                        const int x = rr[1][b], y = cr[1][a][reg];
                        const bool ok = tri && x >= 0 && y >= 0;
                        const int lo = min(x, y) - npc[1], hi = max(x, y) - npc[1];
                        g1[a][b][reg] = Sc[1][ok ? hi + (size_t)nbc[1] * lo : 0];
                        if (!ok) g1[a][b][reg] = 0.0;
                    }
                }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) cv[a][b][reg] = g0[a][b][reg] + g1[a][b][reg];
        if (mask && bx == 0) {                                // diagonal entries exist only in the tiles on the diagonal
            const int* gd = fd.dofs + dp;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int cc = cj + wc + 16 * a + l4 + 4 * reg;
                        const int r = ri + wr + 16 * b + l15;
                        if (r == cc && cok[a][reg] && cc < np && mask[gd[r]]) cv[a][b][reg] = 1.0;
                    }
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int cc = cj + wc + 16 * a + l4 + 4 * reg;
                const int r = ri + wr + 16 * b + l15;
                if (cok[a][reg] && r < nf && r >= cc) col_ptr(a, reg)[r] = cv[a][b][reg] - acc[a][b][reg];
            }
}

// ---- the narrow (schur 0) update cut finer, for launches that do not fill the chip.  At the top of the tree a narrow update is a
// few dozen 64 x 64 tiles whose waves each walk K MFMAs behind K / 16 staged loads: the launch lasts as long as that one chain
// (12 us at K = 128, 30 at K = 512, whatever the number of tiles).  Here a workgroup takes a 32 x 32 tile and a wave ONE 16 x 16
// block: K / 4 MFMAs, a quarter of the chain, four times the workgroups.  No LDS: both MFMA operands are 16 consecutive rows of a
// factor column per k (one 128-byte segment), loaded straight to registers in batches of 32 factor columns (two batches, 32 loads,
// in flight per lane), the next batch requested before the products of the current one.  K range, column range and the gathering epilogue
// are those of k_trailing_mfma with schur == 0; the host takes this kernel when the K range is a multiple of 64.
template <bool GATHER>
__global__ void __launch_bounds__(256, 2)
k_trailing_fine(FrontDev fd, const int* __restrict__ level_nodes, int first, int C0, int K0, const unsigned char* __restrict__ mask, int bx_first) {
    const int t = level_nodes[first + blockIdx.z];
    const int np = fd.npiv[t];
    const int nf = fd.nf[t];
    const TrailRange tr = trail_range(0, C0, K0, NBO, np, nf);
    const int kc0 = tr.kc0, kw = tr.kw, col_lo = tr.col_lo, col_hi = tr.col_hi;
    if (kw <= 0) return;
    constexpr int FT = 32;                                       // tile edge
    const int bxv = (int)blockIdx.x + bx_first;
    const int by = bxv & 3, bx = bxv >> 2;                       // four column tiles per panel
    const int cj = (col_lo & ~1) + by * FT;
    if (cj >= col_hi) return;
    const int ri = cj + bx * FT;                                 // row tiles start at the column tile (lower triangle)
    if (ri >= nf) return;
    const FrontView fv = front_view(fd, t);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = (wv & 1) * 16, wc = (wv >> 1) * 16;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int ldp = ldp_of(nf);
    // operand rows: A = tile columns (D index i), B = tile rows (D index j); rows past the front are clamped (never stored)
    const double* pa = fv.P + min(cj + wc + l15, nf - 1) + (size_t)ldp * (kc0 + l4);
    const double* pb = fv.P + min(ri + wr + l15, nf - 1) + (size_t)ldp * (kc0 + l4);
    const size_t kstep = (size_t)ldp * 4;
    constexpr int KB = 32, NS = KB / 4;                          // factor columns per batch, k-steps per batch (two batches in flight)
    double a0[NS], b0[NS], a1[NS], b1[NS];
    mfma_d4 acc = (mfma_d4){0.0, 0.0, 0.0, 0.0};
    const int nbatch = kw / KB;                                  // the host guarantees kw % KB == 0
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) { a0[s_] = pa[kstep * s_]; b0[s_] = pb[kstep * s_]; }
    for (int bt = 0; bt < nbatch; bt += 2) {
        const bool more1 = bt + 1 < nbatch, more2 = bt + 2 < nbatch;
        if (more1) {
#pragma unroll
            for (int s_ = 0; s_ < NS; ++s_) { a1[s_] = pa[kstep * (NS * (bt + 1) + s_)]; b1[s_] = pb[kstep * (NS * (bt + 1) + s_)]; }
        }
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[s_], b0[s_], acc, 0, 0, 0);
        if (more2) {
#pragma unroll
            for (int s_ = 0; s_ < NS; ++s_) { a0[s_] = pa[kstep * (NS * (bt + 2) + s_)]; b0[s_] = pb[kstep * (NS * (bt + 2) + s_)]; }
        }
        if (more1) {
#pragma unroll
            for (int s_ = 0; s_ < NS; ++s_) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[s_], b1[s_], acc, 0, 0, 0);
        }
    }
    // epilogue: D[i = l4 + 4 reg -> column][j = l15 -> row]
    const int r = ri + wr + l15;
    bool cok[4];
    double* cp[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int cc = cj + wc + l4 + 4 * reg;
        cok[reg] = cc >= col_lo && cc < col_hi && r < nf && r >= cc;
        cp[reg] = cok[reg] ? fv.col(cc) + r : fv.P;              // safe address for the lanes that store nothing
    }
    double cv[4];
    if (!GATHER) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) cv[reg] = *cp[reg];
    } else {
        const long long dp = fd.doff[t];
        double g[2][4];
#pragma unroll
        for (int sd = 0; sd < 2; ++sd) {
            const int ch = fd.child[sd][t];
            const int chs = ch >= 0 ? ch : t;
            const int npc = fd.npiv[chs], nbc = fd.nf[chs] - npc;
            const double* Sc = fd.S + fd.soff[chs];
            const int x = (ch >= 0 && r < nf) ? fd.cinv[sd][dp + r] : -1;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int cc = cj + wc + l4 + 4 * reg;
                const int y = (ch >= 0 && cok[reg]) ? fd.cinv[sd][dp + cc] : -1;
                const bool ok = cok[reg] && x >= 0 && y >= 0;
                const int lo = min(x, y) - npc, hi = max(x, y) - npc;
                g[sd][reg] = Sc[ok ? hi + (size_t)nbc * lo : 0];
                if (!ok) g[sd][reg] = 0.0;
            }
        }
        const int* gd = fd.dofs + dp;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int cc = cj + wc + l4 + 4 * reg;
            cv[reg] = g[0][reg] + g[1][reg];
            if (mask && r == cc && cok[reg] && cc < np && mask[gd[r]]) cv[reg] = 1.0;
        }
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg)
        if (cok[reg]) *cp[reg] = cv[reg] - acc[reg];
}

// ---- the rank-k update of the levels of many small fronts (K <= 160: the leaves and the four or five levels above them), where
// the update moves its compulsory bytes only and is bound by HBM, not by the matrix cores.  k_trailing_mfma runs those levels at
// 0.31 of the HBM peak: a 64 x 64 tile there is a chain of dependent round trips -- front metadata, two to nine staged K steps with
// one step of look-ahead, the index maps of the gather, the children's entries, the store -- about 13 us with three workgroups
// per CU, and every tile reads its own 2 x 64 rows of the K panel and its own maps.  Here a workgroup takes a 64-ROW STRIP of a
// front's update (all updated columns up to the diagonal, in chunks of at most STRIP_CH blocks of 16 columns):
//   * front metadata, the row maps of the gather (one register per child and lane) and the strip's own 64 rows of the K panel
//     (the MFMA B operand: 16 rows x K per wave, in registers, k_panel_rows' layout) are loaded ONCE per strip;
//   * the column maps of the chunk go to LDS once (2 x 256 entries);
//   * per block of 16 columns the other operand (16 rows of the panel x K: 16 KB at K = 128) is staged through LDS, double
//     buffered, and while the matrix cores work on block n the panel rows AND the children's entries of block n + 1 are already in
//     flight: a stage is one barrier, K / 4 MFMAs per wave and one epilogue of four entries per lane whose loads were issued a stage
//     earlier -- no dependent round trip inside the loop.
// The product is formed as in k_panel_rows (D[i = column][j = row]: a lane stores along the rows of the column-major front).
// K range, column range and the gather are those of k_trailing_mfma (trail_range); KQ = K steps of 4 the B operand has room for.
constexpr int STRIP_CH = 16;           // blocks of 16 columns per workgroup at most
constexpr int STRIP_KMAX = 160;        // the widest K range the kernel is instantiated for

// workgroups the update of one front needs: rows and updated columns counted from the even column anchor
__device__ __host__ inline int strip_wgs(int nrows, int ncols) {
    const int nstrips = (nrows + 63) / 64, ncb = (ncols + 15) / 16;
    int n = 0;
    for (int s = 0; s < nstrips; ++s) {
        const int nbk = 4 * (s + 1) < ncb ? 4 * (s + 1) : ncb;
        n += (nbk + STRIP_CH - 1) / STRIP_CH;
    }
    return n;
}

template <bool GATHER, int KQ, int DEPTH>
__global__ void __launch_bounds__(256, KQ > 32 ? 2 : 3)
k_schur_strip(FrontDev fd, const int* __restrict__ level_nodes, int first, int C0, int schur, int K0, int KW,
              const unsigned char* __restrict__ mask) {
    // whole fronts per XCD on levels of many fronts (the strips of a front read the same panel rows): k_trailing_mfma's map
    int bxv = blockIdx.x, zv = blockIdx.z;
    {
        const int G = gridDim.x, n = gridDim.z;
        if (n >= 256 && (n & 7) == 0) {
            const int flat = bxv + G * zv, x = flat & 7, q = flat >> 3;
            zv = (q / G) * 8 + x; bxv = q % G;
        }
    }
    const int t = level_nodes[first + zv];
    const int np = fd.npiv[t], nf = fd.nf[t];
    const TrailRange tr = trail_range(schur, C0, K0, KW, np, nf);
    const int kc0 = tr.kc0, kw = tr.kw, col_lo = tr.col_lo, col_hi = tr.col_hi;
    if (kw <= 0 || col_lo >= col_hi) return;
    const int anchor = col_lo & ~1;
    const int nstrips = (nf - anchor + 63) / 64, ncb = (col_hi - anchor + 15) / 16;
    // strip s (rows anchor + 64 s ..) and chunk of its column blocks; the longest strips come first in launch order
    int s = nstrips - 1, chunk = -1;
    for (int lin = bxv; s >= 0; --s) {
        const int nbk = min(4 * (s + 1), ncb), n = (nbk + STRIP_CH - 1) / STRIP_CH;
        if (lin < n) { chunk = lin; break; }
        lin -= n;
    }
    if (chunk < 0) return;
    const int cb_lo = STRIP_CH * chunk, cb_hi = min(min(4 * (s + 1), ncb), cb_lo + STRIP_CH);
    __shared__ __attribute__((aligned(16))) double sA[2][4 * KQ][16];      // sA[.][k][c] = L[anchor + 16 cb + c][kc0 + k]
    __shared__ int cmap[2][16 * STRIP_CH];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const FrontView fv = front_view(fd, t);
    const int ldp = ldp_of(nf);
    const int rw = anchor + 64 * s + 16 * wv;                  // this wave's 16 rows
    const int r = rw + l15;
    const bool rok = r < nf;
    const double* Lk = fv.P + (size_t)ldp * kc0;               // the K panel: always pivot columns
    // B operand: the wave's rows, every k, in registers.  Addresses are clamped and NOTHING is masked: rows past the front give
    // products that are never stored, and the K range is cut off by the other operand (zero beyond kw).  (A select on the loaded
    // value makes the compiler sink every load into a branch of its own with a full wait behind it: 32 dependent round trips.)
    double b[KQ];
    {
        const double* rowp = Lk + min(r, nf - 1);
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk) b[kk] = rowp[(size_t)ldp * min(4 * kk + l4, kw - 1)];
    }
    // staging of the other operand: thread (row pair cp, k slot kq) loads rows 2 cp, 2 cp + 1 of the block at k = kq + 32 q
    const int cp = tid & 7, kq = tid >> 3;
    constexpr int NQ = (4 * KQ + 31) / 32;
    constexpr int NSET = DEPTH + 1;
    // A wave's vector-memory operations retire in ISSUE order: waiting for the panel rows of the next block waits for every load
    // issued before them.  So the panel rows (L2 hits) are requested one block FURTHER ahead than the children's entries (HBM) and
    // always before them -- DEPTH + 1 register sets -- or the entries would have one stage to arrive whatever their look-ahead.
    d2 pa[NSET][NQ];
    auto fetchA = [&](int cb, d2 (&dst)[NQ]) {
        const double* p = Lk + min(anchor + 16 * cb + 2 * cp, ldp - 2);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int k = kq + 32 * q;
            const d2 v = *reinterpret_cast<const d2*>(p + (size_t)ldp * min(k, kw - 1));
            const d2 z = {0.0, 0.0};
            dst[q] = k < kw ? v : z;
        }
    };
    auto stashA = [&](int buf, const d2 (&src)[NQ]) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int k = kq + 32 * q;
            if (k < 4 * KQ) *reinterpret_cast<d2*>(&sA[buf][k][2 * cp]) = src[q];
        }
    };
    // the gather: rows of the children's fronts that land on this lane's row (registers) and on the chunk's columns (LDS)
    const long long dp = fd.doff[t];
    int xrow[2] = {-1, -1}, npc[2] = {0, 0}, nbc[2] = {0, 0};
    const double* Sc[2] = {fd.S, fd.S};
    if (GATHER) {
        int chd[2];
#pragma unroll
        for (int sd = 0; sd < 2; ++sd) {
            const int ch = fd.child[sd][t];
            const int chs = ch >= 0 ? ch : t;
            chd[sd] = ch;
            npc[sd] = fd.npiv[chs]; nbc[sd] = fd.nf[chs] - npc[sd];
            Sc[sd] = fd.S + fd.soff[chs];
            const int x = fd.cinv[sd][dp + min(r, nf - 1)];
            xrow[sd] = (ch >= 0 && rok) ? x : -1;
        }
        for (int i = tid; i < 2 * 16 * STRIP_CH; i += 256) {
            const int sd = i / (16 * STRIP_CH), j = i % (16 * STRIP_CH);
            const int cc = anchor + 16 * cb_lo + j;
            const int y = fd.cinv[sd][dp + min(cc, nf - 1)];
            cmap[sd][j] = (chd[sd] >= 0 && cc >= col_lo && cc < col_hi) ? y : -1;
        }
    }
    fetchA(cb_lo, pa[0]);
    stashA(0, pa[0]);
    __syncthreads();
    // entries of C this lane's epilogue needs for block cb: the children's (GATHER) or the front's own -- requested a stage ahead;
    // `okm` keeps which of them exist (the select waits for the data, so it is left to the epilogue)
    constexpr int NG = GATHER ? 8 : 4;
    auto request = [&](int cb, double (&g)[NG], unsigned& okm) {
        okm = 0;
        const int j0 = 16 * (cb - cb_lo);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int cc = anchor + 16 * cb + l4 + 4 * reg;
            const bool tri = rok && r >= cc && cc >= col_lo && cc < col_hi;
            if (GATHER) {
#pragma unroll
                for (int sd = 0; sd < 2; ++sd) {
                    const int x = xrow[sd], y = cmap[sd][j0 + l4 + 4 * reg];
                    const bool ok = tri && x >= 0 && y >= 0;
                    const int lo = min(x, y) - npc[sd], hi = max(x, y) - npc[sd];
                    g[2 * reg + sd] = Sc[sd][ok ? hi + (size_t)nbc[sd] * lo : 0];
                    okm |= (ok ? 1u : 0u) << (2 * reg + sd);
                }
            } else {
                const double* src = tri ? fv.col(cc) + r : fv.P;      // (one load from a selected address, not a load in a branch)
                g[reg] = *src;
                okm |= (tri ? 1u : 0u) << reg;
            }
        }
    };
    const int* gd = fd.dofs + dp;
    // one stage: block cb with the entries `g` requested DEPTH stages earlier; the next block's panel rows and the entries of block
    // cb + DEPTH go on their way first
    auto stage = [&](int cb, const double (&g)[NG], unsigned okm, double (&gnext)[NG], unsigned& oknext, d2 (&anew)[NQ], const d2 (&anext)[NQ]) {
        const int buf = (cb - cb_lo) & 1;
        if (cb + DEPTH + 1 < cb_hi) fetchA(cb + DEPTH + 1, anew);
        if (cb + DEPTH < cb_hi) request(cb + DEPTH, gnext, oknext);
        const int cj = anchor + 16 * cb;
        // the wave has entries in this block if its last row reaches the block's first column (lower triangle)
        if (rw < nf && rw + 15 >= cj) {
            mfma_d4 acc = (mfma_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < KQ; ++kk)
                if (4 * kk < kw) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sA[buf][4 * kk + l4][l15], b[kk], acc, 0, 0, 0);   // D[i = column][j = row]
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int cc = cj + l4 + 4 * reg;
                const bool tri = rok && r >= cc && cc >= col_lo && cc < col_hi;
                double cv;
                if (GATHER) {
                    cv = (((okm >> (2 * reg)) & 1u) ? g[2 * reg] : 0.0) + (((okm >> (2 * reg + 1)) & 1u) ? g[2 * reg + 1] : 0.0);
                    if (mask && tri && r == cc && cc < np && mask[gd[r]]) cv = 1.0;      // strong-BC pivots: unit diagonal, as in the extend-add
                } else {
                    cv = g[reg];
                }
                if (tri) fv.col(cc)[r] = cv - acc[reg];
            }
        }
        if (cb + 1 < cb_hi) stashA(buf ^ 1, anext);
        __syncthreads();
    };
    // DEPTH + 1 register sets each for the requested entries and for the panel rows, rotated by unrolling (a copy would wait for the
    // loads just issued): at stage n the panel rows of block n + DEPTH + 1 are requested, then the entries of block n + DEPTH.
    // A stage lasts K / 4 MFMAs (0.35-0.9 us), a load of the children's entries from HBM 2-3 us under load.
    double g[NSET][NG];
    unsigned okm[NSET];
#pragma unroll
    for (int d = 1; d <= DEPTH; ++d)
        if (cb_lo + d < cb_hi) fetchA(cb_lo + d, pa[d % NSET]);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        okm[d] = 0;
        if (cb_lo + d < cb_hi) request(cb_lo + d, g[d], okm[d]);
    }
    for (int cb = cb_lo; cb < cb_hi; cb += NSET) {
#pragma unroll
        for (int u = 0; u < NSET; ++u)
            if (cb + u < cb_hi) stage(cb + u, g[u], okm[u], g[(u + DEPTH) % NSET], okm[(u + DEPTH) % NSET], pa[u], pa[(u + 1) % NSET]);
    }
}

// ---- the same rank-k update on 128 x 128 tiles (schur 1, 2, 5: the updates of whole Schur complements / trailing matrices).
// The per-level counter table (profiles/r3_pmc_trailing_levels.md) shows where the 64 x 64 kernel loses: on the MFMA-bound
// levels (6 and up) it moves 2.2-3 x its compulsory bytes at ~4 TB/s -- every tile re-reads 2 x 64 rows of the K panel, and
// the tiles of a front are spread over eight L2s -- and it issues one LDS read per MFMA.  A workgroup of the same four waves on a
// 128 x 128 tile (a wave: 64 x 64 = 4 x 4 MFMA blocks, 128 accumulator registers) reads half the panel bytes per updated entry and
// one LDS operand per TWO MFMAs.  Used where the fronts are large enough to fill the chip with such tiles (host: option "big_nb").
constexpr int TSB = 128;
constexpr int LSTRB = TSB + 16;

template <bool GATHER>
__global__ void __launch_bounds__(256, 2)
k_trailing_big(FrontDev fd, const int* __restrict__ level_nodes, int first, int C0, int schur, int K0, int KW,
               const unsigned char* __restrict__ mask) {
    const int t = level_nodes[first + blockIdx.z];
    const int np = fd.npiv[t];
    const int nf = fd.nf[t];
    const TrailRange tr = trail_range(schur, C0, K0, KW, np, nf);
    const int kc0 = tr.kc0, kw = tr.kw, col_lo = tr.col_lo, col_hi = tr.col_hi;
    if (kw <= 0) return;
    const int lin = blockIdx.x;
    int ti = (int)((sqrt(8.0 * lin + 1.0) - 1.0) * 0.5);
    while ((ti + 1) * (ti + 2) / 2 <= lin) ++ti;
    while (ti * (ti + 1) / 2 > lin) --ti;
    const int by = lin - ti * (ti + 1) / 2, bx = ti - by;          // column tile by, row tile by + bx
    const int cj = (col_lo & ~1) + by * TSB;
    if (cj >= col_hi) return;
    const int ri = cj + bx * TSB;
    if (ri >= nf) return;
    const FrontView fv = front_view(fd, t);
    constexpr int KC = 16;
    extern __shared__ __attribute__((aligned(16))) double lds_big[];
    typedef double stage_t[KC][LSTRB];
    stage_t* si = reinterpret_cast<stage_t*>(lds_big);             // si[buf][k][r] = L[ri + r][kc0 + k0 + k]
    stage_t* sj = si + 2;                                          // sj[buf][k][c] = L[cj + c][kc0 + k0 + k]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = (wv & 1) * 64, wc = (wv >> 1) * 64;
    const int l15 = lane & 15, l4 = lane >> 4;
    // staging: thread (row pair rp of 64, group cg of 4) loads factor columns cg, cg + 4, cg + 8, cg + 12 of both operand blocks
    const int rp = tid & 63, cg = tid >> 6;
    const int ldp = ldp_of(nf);
    const int rowi = min(ri + 2 * rp, ldp - 2), rowj = min(cj + 2 * rp, ldp - 2);
    const char* base = reinterpret_cast<const char*>(fv.P + (size_t)ldp * kc0);
    const size_t stage_bytes = (size_t)ldp * KC * 8;
    unsigned obi[4], obj[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        obi[q] = 8u * (unsigned)(rowi + ldp * (cg + 4 * q));
        obj[q] = 8u * (unsigned)(rowj + ldp * (cg + 4 * q));
    }
    d2 pi[4], pj[4];
    auto fetch = [&](const char* b) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            pi[q] = *reinterpret_cast<const d2*>(b + obi[q]);
            pj[q] = *reinterpret_cast<const d2*>(b + obj[q]);
        }
    };
    auto fetch_tail = [&](const char* b, int left) {
        const d2 z = {0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned a = 8u * (unsigned)(ldp * min(cg + 4 * q, left - 1));
            const d2 vi = *reinterpret_cast<const d2*>(b + 8u * (unsigned)rowi + a), vj = *reinterpret_cast<const d2*>(b + 8u * (unsigned)rowj + a);
            pi[q] = cg + 4 * q < left ? vi : z;
            pj[q] = cg + 4 * q < left ? vj : z;
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            *reinterpret_cast<d2*>(&si[buf][cg + 4 * q][2 * rp]) = pi[q];
            *reinterpret_cast<d2*>(&sj[buf][cg + 4 * q][2 * rp]) = pj[q];
        }
    };
    if (kw >= KC) fetch(base); else fetch_tail(base, kw);
    base += stage_bytes;
    mfma_d4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (mfma_d4){0.0, 0.0, 0.0, 0.0};
    stash(0);
    __syncthreads();
    int cur = 0;
    for (int k0 = 0; k0 < kw; k0 += KC) {
        const int left = kw - k0 - KC;
        if (left > 0) {
            if (left >= KC) fetch(base); else fetch_tail(base, left);
            base += stage_bytes;
        }
#pragma unroll
        for (int kk = 0; kk < KC; kk += 4) {
            double av[4], bv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) av[a] = sj[cur][kk + l4][wc + 16 * a + l15];
#pragma unroll
            for (int b = 0; b < 4; ++b) bv[b] = si[cur][kk + l4][wr + 16 * b + l15];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
        if (left > 0) stash(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    // epilogue, one 16-column block of the wave's quarter at a time (a): C -= D, or C = (children's entries) - D
    const long long dp = fd.doff[t];
    int rr[2][4];
    const double* Sc[2] = {nullptr, nullptr};
    int npc[2] = {0, 0}, nbc[2] = {0, 0}, chv[2] = {-1, -1};
    if (GATHER) {
#pragma unroll
        for (int sd = 0; sd < 2; ++sd) {
            const int ch = fd.child[sd][t];
            const int chs = ch >= 0 ? ch : t;
            chv[sd] = ch;
            npc[sd] = fd.npiv[chs]; nbc[sd] = fd.nf[chs] - npc[sd];
            Sc[sd] = fd.S + fd.soff[chs];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int r = ri + wr + 16 * b + l15;
                rr[sd][b] = fd.cinv[sd][dp + min(r, nf - 1)];
                if (ch < 0 || r >= nf) rr[sd][b] = -1;
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        bool cok[4];
        double* cp[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int cc = cj + wc + 16 * a + l4 + 4 * reg;
            cok[reg] = cc >= col_lo && cc < col_hi;
            cp[reg] = cok[reg] ? fv.col(cc) : fv.P - (nf - 1);
        }
        double cv[4][4];
        if (!GATHER) {
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int cc = cj + wc + 16 * a + l4 + 4 * reg;
                    const int r = ri + wr + 16 * b + l15;
                    const bool ok = cok[reg] && r < nf && r >= cc;
                    cv[b][reg] = cp[reg][ok ? r : nf - 1];
                }
        } else {
            int cr[2][4];
#pragma unroll
            for (int sd = 0; sd < 2; ++sd)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int cc = cj + wc + 16 * a + l4 + 4 * reg;
                    cr[sd][reg] = fd.cinv[sd][dp + min(cc, nf - 1)];
                    if (chv[sd] < 0 || !cok[reg]) cr[sd][reg] = -1;
                }
            double g0[4][4], g1[4][4];
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int cc = cj + wc + 16 * a + l4 + 4 * reg;
                    const int r = ri + wr + 16 * b + l15;
                    const bool tri = r >= cc;
                    {
                        const int x = rr[0][b], y = cr[0][reg];
                        const bool ok = tri && x >= 0 && y >= 0;
                        const int lo = min(x, y) - npc[0], hi = max(x, y) - npc[0];
                        g0[b][reg] = Sc[0][ok ? hi + (size_t)nbc[0] * lo : 0];
                        if (!ok) g0[b][reg] = 0.0;
                    }
                    {
                        const int x = rr[1][b], y = cr[1][reg];
                        const bool ok = tri && x >= 0 && y >= 0;
                        const int lo = min(x, y) - npc[1], hi = max(x, y) - npc[1];
                        g1[b][reg] = Sc[1][ok ? hi + (size_t)nbc[1] * lo : 0];
                        if (!ok) g1[b][reg] = 0.0;
                    }
                }
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) cv[b][reg] = g0[b][reg] + g1[b][reg];
            if (mask && bx == 0) {
                const int* gd = fd.dofs + dp;
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int cc = cj + wc + 16 * a + l4 + 4 * reg;
                        const int r = ri + wr + 16 * b + l15;
                        if (r == cc && cok[reg] && cc < np && mask[gd[r]]) cv[b][reg] = 1.0;
                    }
            }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int cc = cj + wc + 16 * a + l4 + 4 * reg;
                const int r = ri + wr + 16 * b + l15;
                if (cok[reg] && r < nf && r >= cc) cp[reg][r] = cv[b][reg] - acc[a][b][reg];
            }
    }
}

// ------------------------------------------------------------------------------------------ solves
// M^-1 = L^-T L^-1 applied level by level.  Each sweep is split per level into a sequential part on
// the triangular pivot block L11 (one workgroup per front, panels chained through the stored diagonal
// block inverses) and a wide part on the rectangular block L21 (plain / transposed GEMV spread over
// many workgroups), which holds most of the factor's bytes.

// ---- small fronts (tree levels whose largest pivot block is <= WIDE_NP): one workgroup per front does the
// whole sweep of that front; inner loops are unrolled so that many loads are in flight per thread.

// Both kernels are latency-bound per workgroup (a front is 100-300 KB, a level has one to a few rounds of workgroups):
// every phase spreads its loads over all 256 threads -- lanes along the rows of the column-major factor, thread groups
// along the columns, partial sums joined through LDS -- so that a phase is one or two batches of loads in flight, not a
// loop of dependent batches on 32 or 100 active threads (measured at 1M DOF: levels 1-3 70-89 -> see DESIGN.md).
constexpr int SMALL_PART = 512;      // doubles of LDS for the partial sums of a phase

// forward: y_p = L11^-1 v_p -> yv ; v_B -= L21 y_p
__global__ void __launch_bounds__(256, 4)
k_front_fwd_small(FrontDev fd, const int* __restrict__ level_nodes, double* __restrict__ v, double* __restrict__ yv) {
    const int t = level_nodes[blockIdx.x];
    const int np = fd.npiv[t], nf = fd.nf[t];
    if (np == 0) return;
    const double* F = fd.P + fd.poff[t];                       // the factor columns [L11; L21]
    const int ldp = ldp_of(nf);
    const int* gd = fd.dofs + fd.doff[t];
    extern __shared__ double sh[];
    double* y = sh;              // np
    double* part = sh + np;      // SMALL_PART
    const int tid = threadIdx.x;
    for (int p = tid; p < np; p += 256) y[p] = v[gd[p]];
    __syncthreads();
    const int npan = (np + NB - 1) / NB;
    for (int k = 0; k < npan; ++k) {
        const int c0 = k * NB, wb = min(NB, np - c0);
        const double* Li = fd.Linv + fd.linvoff[t] + (size_t)k * NB * NB;
        {   // y_k = Linv_k y[c0 .. c0 + wb): thread (row r, group g) takes the columns g, g + 8, g + 16, g + 24 that are <= r
            const int r = tid & 31, g = tid >> 5;
            double a[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int mm = g + 8 * q; a[q] = (mm <= r && r < wb) ? Li[r + NB * mm] : 0.0; }
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int mm = g + 8 * q; s += a[q] * y[c0 + min(mm, wb - 1)]; }
            part[32 * g + r] = s;
        }
        __syncthreads();
        if (tid < 32) {
            double s = 0.0;
#pragma unroll
            for (int g = 0; g < 8; ++g) s += part[32 * g + tid];
            part[256 + tid] = s;                               // y_k, read by the update below
            if (tid < wb) y[c0 + tid] = s;
        }
        __syncthreads();
        // the pivot rows below the panel: y[r] -= L11[r][c0 .. c0 + wb) y_k; thread (row slot tid & 127, half tid >> 7)
        const int r1 = c0 + wb;
        if (r1 < np) {
            const int h = tid >> 7;
            for (int rb = r1; rb < np; rb += 128) {
                const int r = rb + (tid & 127);
                double a[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) { const int mm = 16 * h + q; a[q] = (r < np && mm < wb) ? F[r + (size_t)ldp * (c0 + mm)] : 0.0; }
                double s = 0.0;
#pragma unroll
                for (int q = 0; q < 16; ++q) s += a[q] * part[256 + 16 * h + q];
                if (h) part[tid & 127] = s;
                __syncthreads();
                if (!h && r < np) y[r] -= s + part[tid & 127];
                __syncthreads();
            }
        }
    }
    for (int p = tid; p < np; p += 256) yv[gd[p]] = y[p];
    // v_B -= L21 y: row slots of 64, 128 or 256 and 4, 2 or 1 column groups, 16 loads in flight per thread
    const int nb = nf - np;
    if (nb > 0) {
        const int slots = nb <= 64 ? 64 : nb <= 128 ? 128 : 256, G = 256 / slots;
        const int rs = tid % slots, g = tid / slots;
        const int cper = (np + G - 1) / G, cbeg = g * cper, cend = min(np, cbeg + cper);
        for (int rb = 0; rb < nb; rb += slots) {
            const int r = rb + rs;
            const double* row = F + np + min(r, nb - 1);
            double s = 0.0;
            // 32 loads in flight per thread: on the levels that fit the chip in one round of workgroups the sweep lasts as long as one
            // workgroup's chain of dependent load rounds does
            for (int cb = cbeg; cb < cend; cb += 32) {
                double a[32];
#pragma unroll
                for (int q = 0; q < 32; ++q) a[q] = cb + q < cend ? row[(size_t)ldp * (cb + q)] : 0.0;
#pragma unroll
                for (int q = 0; q < 32; ++q) s += a[q] * y[min(cb + q, np - 1)];
            }
            if (G > 1) {
                if (g) part[slots * (g - 1) + rs] = s;
                __syncthreads();
                if (!g)
                    for (int gg = 1; gg < G; ++gg) s += part[slots * (gg - 1) + rs];
            }
            if (!g && r < nb) atomicAdd(&v[gd[np + r]], -s);
            if (G > 1) __syncthreads();
        }
    }
}

// backward: x_p = L11^-T (y_p - L21^T x_B)
template <bool BFLY>
__global__ void __launch_bounds__(256, 4)
k_front_bwd_small(FrontDev fd, const int* __restrict__ level_nodes, const double* __restrict__ sv, double* __restrict__ xv) {
    const int t = level_nodes[blockIdx.x];
    const int np = fd.npiv[t], nf = fd.nf[t];
    if (np == 0) return;
    const double* F = fd.P + fd.poff[t];                       // the factor columns [L11; L21]
    const int ldp = ldp_of(nf);
    const int* gd = fd.dofs + fd.doff[t];
    extern __shared__ double sh[];
    double* x = sh;              // nf: s_p (then x_p) in [0, np), x_B behind
    double* part = sh + nf;      // SMALL_PART
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int p = tid; p < nf; p += 256) x[p] = p < np ? sv[gd[p]] : xv[gd[p]];
    __syncthreads();
    // s_p -= L21^T x_B: a wave takes eight columns at a time, lanes along the rows, up to four row chunks: 32 loads in
    // flight per lane, then eight wave reductions
    const int nb = nf - np;
    if (nb > 0) {
        for (int cb = 8 * wid; cb < np; cb += 32) {
            double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            for (int rb = 0; rb < nb; rb += 256) {
                double a[8][4];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const double* col = F + np + (size_t)ldp * min(cb + q, np - 1);
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const int r = rb + lane + 64 * u; a[q][u] = r < nb ? col[r] : 0.0; }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = rb + lane + 64 * u;
                    const double xr = x[np + min(r, nb - 1)];
#pragma unroll
                    for (int q = 0; q < 8; ++q) s[q] += a[q][u] * xr;
                }
            }
            if (BFLY) {
                int q;
                const double tot = wave_sum_cols<8>(s, lane, q);
                if (!(lane & 7) && cb + q < np) x[cb + q] -= tot;            // a column belongs to one wave
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const double tot = wave_sum(s[q]);
                    if (lane == 0 && cb + q < np) x[cb + q] -= tot;
                }
            }
        }
        __syncthreads();
    }
    const int npan = (np + NB - 1) / NB;
    for (int k = npan - 1; k >= 0; --k) {
        const int c0 = k * NB, wb = min(NB, np - c0);
        const double* Li = fd.Linv + fd.linvoff[t] + (size_t)k * NB * NB;
        // s_k -= L11[r1 .., panel]^T x[r1 ..): the pivot rows below the panel (solved already); wave = eight columns
        const int r1 = c0 + wb, nr = np - r1;
        if (nr > 0) {
            const int cb = 8 * wid;
            double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            for (int rb = 0; rb < nr; rb += 256) {
                double a[8][4];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const double* col = F + r1 + (size_t)ldp * (c0 + min(cb + q, wb - 1));
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const int r = rb + lane + 64 * u; a[q][u] = r < nr ? col[r] : 0.0; }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = rb + lane + 64 * u;
                    const double xr = x[r1 + min(r, nr - 1)];
#pragma unroll
                    for (int q = 0; q < 8; ++q) s[q] += a[q][u] * xr;
                }
            }
            if (BFLY) {
                int q;
                const double tot = wave_sum_cols<8>(s, lane, q);
                if (!(lane & 7) && cb + q < wb) x[c0 + cb + q] -= tot;
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const double tot = wave_sum(s[q]);
                    if (lane == 0 && cb + q < wb) x[c0 + cb + q] -= tot;
                }
            }
            __syncthreads();
        }
        {   // x_k = Linv_k^T s_k: thread (column c, group g) takes the rows c + g, c + g + 8, ... < wb
            const int c = tid & 31, g = tid >> 5;
            double a[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int r = c + g + 8 * q; a[q] = (r < wb && c < wb) ? Li[r + NB * c] : 0.0; }
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int r = c + g + 8 * q; s += a[q] * x[c0 + min(r, wb - 1)]; }
            part[32 * g + c] = s;
        }
        __syncthreads();
        if (tid < wb) {
            double s = 0.0;
#pragma unroll
            for (int g = 0; g < 8; ++g) s += part[32 * g + tid];
            x[c0 + tid] = s;
        }
        __syncthreads();
    }
    for (int p = tid; p < np; p += 256) xv[gd[p]] = x[p];
}

// ---- wide levels (few fronts, or large pivot blocks): the sweeps are four plain matrix-vector products per level
//
//   forward    y_p = X b_p                 (k_sweep_gemv_n<true>)     v_B -= L21 y_p          (k_sweep_gemv_n<false>)
//   backward   s_p = y_p - L21^T x_B       (k_sweep_gemv_t<false>)    x_p  = X^T s_p          (k_sweep_gemv_t<true>)
//
// with X = L11^-1 formed explicitly after the factorisation (k_xinv below).  Substitution through L11 is a chain of one
// dependent step per 128 pivot columns -- at the top of the tree (pivot blocks of 1000-1500 columns, a handful of
// fronts) that chain was 10-12 launches of ~9 us per sweep and level with the chip idle; as products with X and L21
// every sweep is two launches per level whose 128 x 128 tiles spread over the whole chip, and each factor byte is
// still read exactly once.  Partial sums of a tile are added atomically (the output entries are zeroed by one memset
// per sweep).

// 128 x 128 tile, 256 threads: thread (row lr, column half ch) keeps 32 loads in flight at a time
template <bool TRI>
__device__ __forceinline__ void gemv_n_tile(const FrontDev& fd, int t, int bx, const double* __restrict__ in, double* __restrict__ out) {
    const int np = fd.npiv[t], nf = fd.nf[t];
    if (np == 0) return;
    const int nct = (np + 127) / 128;
    int ti, tj;                                              // row tile, column tile
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
    const double* M = TRI ? fd.X + fd.xoff[t] : fd.P + fd.poff[t] + np;       // L21 starts at row np of the pivot columns
    const int nrows = TRI ? np : nf - np;
    // L21: the pivot columns are dealt evenly over the front's column tiles (np = 210: 106 + 104 columns, not 128 + 82 -- a
    // workgroup lasts as long as its longest thread); X keeps square tiles, its triangle is enumerated that way
    const int cw = TRI ? 128 : ((np + nct - 1) / nct + 1) & ~1, hw = cw / 2;
    const int r0 = 128 * ti, c0 = cw * tj;
    __shared__ double xs[128];
    __shared__ double part[128];
    const int tid = threadIdx.x, lr = tid & 127, ch = tid >> 7;
    if (tid < 128) xs[tid] = (tid < cw && c0 + tid < np) ? in[gd[c0 + tid]] : 0.0;
    const int r = r0 + lr;
    const double* row = M + r + (size_t)ld * (c0 + hw * ch);
    // triangular tiles on the diagonal: only columns <= row
    const int cmax = min(np - c0 - hw * ch, hw);             // columns of this half inside the pivot block
    const int clim = (TRI && ti == tj) ? min(cmax, lr - hw * ch + 1) : cmax;
    double a[32];
    double s = 0.0;
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int k = 0; k < 32; ++k) a[k] = (r < nrows && 32 * h + k < clim) ? row[(size_t)ld * (32 * h + k)] : 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) s += a[k] * xs[hw * ch + 32 * h + k];              // hw <= 64: inside xs
    }
    if (ch) part[lr] = s;
    __syncthreads();
    if (!ch && r < nrows) {
        s += part[lr];
        if (TRI) atomicAdd(&out[gd[r]], s);
        else atomicAdd(&out[gd[np + r]], -s);
    }
}
template <bool TRI>
__global__ void __launch_bounds__(256)
k_sweep_gemv_n(FrontDev fd, const int* __restrict__ level_nodes, int first, const double* __restrict__ in, double* __restrict__ out) {
    gemv_n_tile<TRI>(fd, level_nodes[first + blockIdx.y], (int)blockIdx.x, in, out);
}

// transposed products: tile of 128 rows x 128 columns, wave w owns columns 32 w .. 32 w + 31 with its lanes along the
// (contiguous) rows, 64 loads in flight per lane; per column one wave reduction
template <bool TRI>
__device__ __forceinline__ void gemv_t_tile(const FrontDev& fd, int t, int bx, const double* __restrict__ in, double* __restrict__ out) {
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
    const int rbase = TRI ? 0 : np;                          // front row of M's row 0
    const int r0 = 128 * ti, c0 = 128 * tj;
    __shared__ double xs[128];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < 128) xs[tid] = (r0 + tid < nrows) ? in[gd[rbase + r0 + tid]] : 0.0;
    const int ra = r0 + lane, rb = r0 + lane + 64;
    double a0[32], a1[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const int c = c0 + 32 * wv + k;
        const double* col = M + (size_t)ld * c;
        // triangular: rows >= column
        const bool ca = c < np && ra < nrows && (!TRI || ra >= c);
        const bool cb = c < np && rb < nrows && (!TRI || rb >= c);
        a0[k] = ca ? col[ra] : 0.0;
        a1[k] = cb ? col[rb] : 0.0;
    }
    __syncthreads();
    const double x0 = xs[lane], x1 = xs[lane + 64];
    // the wave's 32 column sums in one butterfly (wave_sum_cols), out in ONE atomic instruction of 32 lanes
    double p[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) p[k] = a0[k] * x0 + a1[k] * x1;
    int col;
    const double sum = wave_sum_cols<32>(p, lane, col);
    const int c = c0 + 32 * wv + col;
    if (!(lane & 1) && c < np) atomicAdd(&out[gd[c]], TRI ? sum : -sum);
}
template <bool TRI>
__global__ void __launch_bounds__(256)
k_sweep_gemv_t(FrontDev fd, const int* __restrict__ level_nodes, int first, const double* __restrict__ in, double* __restrict__ out) {
    gemv_t_tile<TRI>(fd, level_nodes[first + blockIdx.y], (int)blockIdx.x, in, out);
}

// s_p = y_p - L21^T x_B with one workgroup per 16 pivot columns and ALL boundary rows of the front (no atomics, fixed
// summation order): the better shape for the many medium fronts of the middle levels, where a front's L21 is a few
// hundred rows; the tiled kernel above takes over where one workgroup per 32 columns could not pull the block out of HBM.
constexpr int BB_COLS = 16;
// ATOMIC: the sums are subtracted from sv with atomic adds (the W form of the sweeps, where other workgroups of the same launch add
// to the same entries)
template <bool BFLY, bool ATOMIC>
__device__ __forceinline__ void bnd_cols_block(const FrontDev& fd, int t, int bx, double* __restrict__ sv, const double* __restrict__ xv, double* xs) {
    const int np = fd.npiv[t], nf = fd.nf[t];
    const int nb = nf - np;
    const int c0 = bx * BB_COLS;
    if (c0 >= np || nb == 0) return;
    const int* gd = fd.dofs + fd.doff[t];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int r = tid; r < nb; r += 256) xs[r] = xv[gd[np + r]];
    __syncthreads();
    // a wave owns four columns; 4 columns x 8 row chunks = 32 loads in flight per lane (the upper levels have only one
    // or two workgroups per CU: what is in flight per wave is what pulls the block out of HBM)
    const int ldp = ldp_of(nf);
    const double* L21 = fd.P + fd.poff[t] + np;          // rows np.., column c at + ldp * c
    const int cb = c0 + 4 * wv;
    const double* col[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) col[k] = L21 + (size_t)ldp * min(cb + k, np - 1);
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int rb = 0; rb < nb; rb += 512) {
        double a[4][8];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int r = rb + lane + 64 * u; a[k][u] = r < nb ? col[k][r] : 0.0; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double xr = xs[min(rb + lane + 64 * u, nb - 1)];
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] += a[k][u] * xr;
        }
    }
    if (BFLY) {
        int k;
        const double tot = wave_sum_cols<4>(s, lane, k);
        if (!(lane & 15) && cb + k < np) { if (ATOMIC) atomicAdd(&sv[gd[cb + k]], -tot); else sv[gd[cb + k]] -= tot; }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double tot = wave_sum(s[k]);
            if (lane == 0 && cb + k < np) { if (ATOMIC) atomicAdd(&sv[gd[cb + k]], -tot); else sv[gd[cb + k]] -= tot; }
        }
    }
}
template <bool BFLY>
__global__ void __launch_bounds__(256)
k_sweep_bnd_cols(FrontDev fd, const int* __restrict__ level_nodes, int first, double* __restrict__ sv, const double* __restrict__ xv) {
    extern __shared__ double xs_bnd[];                   // nb
    bnd_cols_block<BFLY, false>(fd, level_nodes[first + blockIdx.y], (int)blockIdx.x, sv, xv, xs_bnd);
}

// ---- the W form of the wide levels (option "sweep_w"): with W = L21 X stored in the place of L21 (k_w_inplace, beside the
// factorisation like X itself) both products of a level read the SAME input,
//   forward    y_p = X b_p,  v_B -= W b_p            backward   x_p = X^T y_p - W^T x_B,
// so a level is ONE launch per direction instead of two dependent ones: the product with X -- ~10 us whatever it moves, the chip
// mostly idle -- runs beside the tiles of W instead of in front of them.  The bytes are the same (W has the shape of L21).
__global__ void __launch_bounds__(256)
k_sweep_fwd_w(FrontDev fd, const int* __restrict__ level_nodes, int first, int nx, double* __restrict__ v, double* __restrict__ y) {
    const int t = level_nodes[first + blockIdx.y];
    if ((int)blockIdx.x < nx) gemv_n_tile<true>(fd, t, (int)blockIdx.x, v, y);
    else gemv_n_tile<false>(fd, t, (int)blockIdx.x - nx, v, v);          // reads the pivot entries of v, adds to the boundary entries
}
template <bool BFLY>
__global__ void __launch_bounds__(256)
k_sweep_bwd_w(FrontDev fd, const int* __restrict__ level_nodes, int first, int nx, const double* __restrict__ y, double* __restrict__ x) {
    extern __shared__ double xs_bnd[];                   // nb
    const int t = level_nodes[first + blockIdx.y];
    if ((int)blockIdx.x < nx) gemv_t_tile<true>(fd, t, (int)blockIdx.x, y, x);
    else bnd_cols_block<BFLY, true>(fd, t, (int)blockIdx.x - nx, x, x, xs_bnd);     // reads the boundary entries of x, adds to the pivot entries
}

// ---- the wide levels as ONE launch per sweep direction: tiles as tasks with per-front dependencies.
// Level by level, a sweep pays two dependent launches per wide level, and the first of the two (the product with X) lasts ~10 us
// whatever it moves: at 1 M DOF twenty launches per sweep, 0.2 ms of an application of 1.5 ms, with the chip mostly idle.  Here every
// 128 x 128 tile of X and of L21 (every block of 16 columns in the backward sweep) of ALL consecutive wide levels is one workgroup of
// one grid, listed in the order the level-wise schedule would run them, and what a launch boundary used to enforce is a counter:
//   forward    X tiles of front t wait until both children are complete; the L21 tiles of t wait until all X tiles of t have added
//              their part of y_p -- with their own first 32 loads per thread already in flight;
//   backward   the column blocks of L21^T x_B of front t wait until the parent is complete, the X^T tiles of t until all column
//              blocks of t have subtracted their part.
// Workgroups are dispatched in grid order and every dependency points to an EARLIER workgroup, so a waiting workgroup never
// waits for one that cannot start.  Memory model (MI355X_MICROARCH.md, inter-workgroup visibility: the per-XCD L2s are not
// coherent, a CU's L1 is never refreshed): every value that crosses workgroups inside the launch is WRITTEN by an agent-scope
// atomic add and READ by a returning agent-scope atomic (add of zero) -- both execute at the memory side, never in a cache;
// a workgroup signals (one lane, agent-scope atomic add on the front's counter) only after every one of its waves has waited
// for its own atomics (s_waitcnt vmcnt(0)) and a workgroup barrier; the consumer polls the counter with agent-scope loads.
struct SweepTask { int slot, kt; };                       // position of the front in level_nodes; kind << 24 | tile
__device__ __host__ inline int sweep_nct(int np) { return (np + 127) / 128; }
__device__ __host__ inline int sweep_xtiles(int np) { const int n = sweep_nct(np); return n * (n + 1) / 2; }
__device__ __host__ inline int sweep_ltiles(int np, int nb) { return ((nb + 127) / 128) * sweep_nct(np); }
constexpr int BB_COLS_F = 16;                             // columns per workgroup of the fused L21^T x_B (== BB_COLS)
__device__ __host__ inline int sweep_bblocks(int np, int nb) { return nb > 0 ? (np + BB_COLS_F - 1) / BB_COLS_F : 0; }

// Every wait has an exit: after ~4 million polls (seconds; a dependency is microseconds away) the workgroup gives up, poisons what it
// was about to compute with a NaN -- the Krylov loop then stops with "NaN residual" -- and signals as if it had finished, so that the
// grid drains whatever went wrong (a workgroup order this code did not expect) instead of hanging the device.
__device__ __forceinline__ bool sweep_wait(const int* cnt, int want) {
    for (int it = 0; __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want; ++it) {
        if (it > (1 << 22)) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    return true;
}
// RM 0: a returning atomic (add of zero): the value as the memory side holds it; RM 1: an agent-scope load (global_load sc1: past
// this CU's L1); RM 2: a plain load (an experiment: NOT safe across workgroups)
template <int RM>
__device__ __forceinline__ double sweep_read(double* p) {
    if (RM == 0) return __hip_atomic_fetch_add(p, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (RM == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
__device__ __forceinline__ void sweep_add(double* p, double v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every wave's atomics have been acknowledged, then one lane counts the workgroup in
__device__ __forceinline__ void sweep_signal(int* c0, int* c1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (c0) __hip_atomic_fetch_add(c0, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (c1) __hip_atomic_fetch_add(c1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// forward: y_p = X b_p (kind 0), v_B -= L21 y_p (kind 1).  cnt[2 slot]: X tiles of the front done; cnt[2 slot + 1]: all tiles done.
template <int RM>
__global__ void __launch_bounds__(256)
k_sweep_wide_fwd(FrontDev fd, const int* __restrict__ level_nodes, const SweepTask* __restrict__ tasks, int slot_lo,
                 const int* __restrict__ slot_of, int* __restrict__ cnt, double* __restrict__ v, double* __restrict__ y) {
    const SweepTask tk = tasks[blockIdx.x];
    const int slot = tk.slot, kind = tk.kt >> 24, tile = tk.kt & 0xffffff;
    const int t = level_nodes[slot];
    const int np = fd.npiv[t], nf = fd.nf[t];
    const int nct = sweep_nct(np);
    const int* gd = fd.dofs + fd.doff[t];
    int* xdone = cnt + 2 * slot;
    int* fin = xdone + 1;
    __shared__ double xs[128];
    __shared__ double part[128];
    const int tid = threadIdx.x, lr = tid & 127, ch = tid >> 7;
    if (kind == 0) {
        int ti = (int)((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
        while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
        while (ti * (ti + 1) / 2 > tile) --ti;
        const int tj = tile - ti * (ti + 1) / 2;
        const int ld = ldx_of(np);
        const double* M = fd.X + fd.xoff[t];
        const int r0 = 128 * ti, c0 = 128 * tj, r = r0 + lr;
        const double* row = M + r;                                 // (r < ld: inside X whatever np is)
        const int cb0 = c0 + 64 * ch;                              // first column of this thread's half
        const int cmax = min(np - cb0, 64);
        const int clim = ti == tj ? min(cmax, lr - 64 * ch + 1) : cmax;
        // the tile's own entries first (they depend on nothing), then the wait for the children; addresses are clamped into the
        // front's columns and the values masked where they are used (no load sits in a branch)
        double a[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) a[k] = row[(size_t)ld * min(cb0 + k, np - 1)];
        if (tid == 0) {
            bool ok = true;
#pragma unroll
            for (int sd = 0; sd < 2; ++sd) {
                const int c = fd.child[sd][t];
                if (c < 0) continue;
                const int cs = slot_of[c];
                if (cs < slot_lo) continue;                    // finished by an earlier launch
                const int cnp = fd.npiv[c];
                ok = sweep_wait(cnt + 2 * cs + 1, sweep_xtiles(cnp) + sweep_ltiles(cnp, fd.nf[c] - cnp)) && ok;
            }
            part[0] = ok ? 0.0 : __builtin_nan("");
        }
        __syncthreads();
        const double poison = part[0];                          // 0, or NaN after a wait that gave up
        __syncthreads();
        if (tid < 128) xs[tid] = (c0 + tid < np ? sweep_read<RM>(&v[gd[c0 + tid]]) : 0.0) + poison;
        __syncthreads();
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) s += (r < np && k < clim ? a[k] : 0.0) * xs[64 * ch + k];
#pragma unroll
        for (int k = 0; k < 32; ++k) a[k] = row[(size_t)ld * min(cb0 + 32 + k, np - 1)];
#pragma unroll
        for (int k = 0; k < 32; ++k) s += (r < np && 32 + k < clim ? a[k] : 0.0) * xs[64 * ch + 32 + k];
        if (ch) part[lr] = s;
        __syncthreads();
        if (!ch && r < np) sweep_add(&y[gd[r]], s + part[lr]);
        sweep_signal(xdone, fin);
    } else {
        const int nb = nf - np;
        const int ti = tile / nct, tj = tile % nct;
        const int ld = ldp_of(nf);
        const double* M = fd.P + fd.poff[t] + np;                  // L21 starts at row np of the pivot columns
        const int cw = ((np + nct - 1) / nct + 1) & ~1, hw = cw / 2;
        const int r0 = 128 * ti, c0 = cw * tj, r = r0 + lr;
        const double* row = M + min(r, nb - 1);
        const int cb0 = c0 + hw * ch;
        const int clim = min(np - cb0, hw);
        double a[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) a[k] = row[(size_t)ld * min(cb0 + k, np - 1)];
        if (tid == 0) part[0] = sweep_wait(xdone, sweep_xtiles(np)) ? 0.0 : __builtin_nan("");
        __syncthreads();
        const double poison = part[0];
        __syncthreads();
        if (tid < 128) xs[tid] = ((tid < cw && c0 + tid < np) ? sweep_read<RM>(&y[gd[c0 + tid]]) : 0.0) + poison;
        __syncthreads();
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) s += (k < clim ? a[k] : 0.0) * xs[hw * ch + k];
#pragma unroll
        for (int k = 0; k < 32; ++k) a[k] = row[(size_t)ld * min(cb0 + 32 + k, np - 1)];
#pragma unroll
        for (int k = 0; k < 32; ++k) s += (32 + k < clim ? a[k] : 0.0) * xs[hw * ch + 32 + k];
        if (ch) part[lr] = s;
        __syncthreads();
        if (!ch && r < nb) sweep_add(&v[gd[np + r]], -(s + part[lr]));
        sweep_signal(nullptr, fin);
    }
}

// backward: s_p = y_p - L21^T x_B (kind 0, one workgroup per 16 pivot columns), x_p = X^T s_p (kind 1).
// cnt[2 slot]: column blocks of the front done; cnt[2 slot + 1]: X^T tiles done (the front is complete when all are).
template <int RM>
__global__ void __launch_bounds__(256)
k_sweep_wide_bwd(FrontDev fd, const int* __restrict__ level_nodes, const SweepTask* __restrict__ tasks, int slot_hi,
                 const int* __restrict__ slot_of, int* __restrict__ cnt, double* __restrict__ sv, double* __restrict__ xv) {
    const SweepTask tk = tasks[blockIdx.x];
    const int slot = tk.slot, kind = tk.kt >> 24, tile = tk.kt & 0xffffff;
    const int t = level_nodes[slot];
    const int np = fd.npiv[t], nf = fd.nf[t];
    const int nb = nf - np;
    const int* gd = fd.dofs + fd.doff[t];
    int* bdone = cnt + 2 * slot;
    int* xfin = bdone + 1;
    extern __shared__ double xsd[];                       // kind 0: nb doubles (x_B); kind 1: 128
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (kind == 0) {
        const int c0 = tile * BB_COLS_F;
        const int ldp = ldp_of(nf);
        const double* L21 = fd.P + fd.poff[t] + np;
        const int cb = c0 + 4 * wv;
        const double* col[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) col[k] = L21 + (size_t)ldp * min(cb + k, np - 1);
        // the first 512 rows of the wave's four columns are requested before the wait
        double a[4][8];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int u = 0; u < 8; ++u) a[k][u] = col[k][min(lane + 64 * u, nb - 1)];
        __shared__ double poison_s;
        if (tid == 0) {
            const int p = fd.parent[t];
            bool ok = true;
            if (p >= 0 && slot_of[p] < slot_hi) ok = sweep_wait(cnt + 2 * slot_of[p] + 1, sweep_xtiles(fd.npiv[p]));
            poison_s = ok ? 0.0 : __builtin_nan("");
        }
        __syncthreads();
        for (int r = tid; r < nb; r += 256) xsd[r] = sweep_read<RM>(&xv[gd[np + r]]) + poison_s;
        __syncthreads();
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        for (int rb = 0; rb < nb; rb += 512) {
            if (rb > 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int u = 0; u < 8; ++u) a[k][u] = col[k][min(rb + lane + 64 * u, nb - 1)];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = rb + lane + 64 * u;
                const double xr = r < nb ? xsd[r] : 0.0;
#pragma unroll
                for (int k = 0; k < 4; ++k) s[k] += a[k][u] * xr;
            }
        }
        int k;
        const double tot = wave_sum_cols<4>(s, lane, k);
        if (!(lane & 15) && cb + k < np) sweep_add(&sv[gd[cb + k]], -tot);
        sweep_signal(bdone, nullptr);
    } else {
        int ti = (int)((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
        while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
        while (ti * (ti + 1) / 2 > tile) --ti;
        const int tj = tile - ti * (ti + 1) / 2;
        const int ld = ldx_of(np);
        const double* M = fd.X + fd.xoff[t];
        const int r0 = 128 * ti, c0 = 128 * tj;
        const int ra = r0 + lane, rb = r0 + lane + 64;
        double a0[32], a1[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const int c = c0 + 32 * wv + k;
            const double* colp = M + (size_t)ld * min(c, np - 1);
            a0[k] = colp[min(ra, np - 1)];
            a1[k] = colp[min(rb, np - 1)];
        }
        __shared__ double poison_x;
        if (tid == 0) poison_x = sweep_wait(bdone, sweep_bblocks(np, nb)) ? 0.0 : __builtin_nan("");
        __syncthreads();
        if (tid < 128) xsd[tid] = (r0 + tid < np ? sweep_read<RM>(&sv[gd[r0 + tid]]) : 0.0) + poison_x;
        __syncthreads();
        const double x0 = xsd[lane], x1 = xsd[lane + 64];
        double p[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const int c = c0 + 32 * wv + k;
            const bool ca = c < np && ra < np && ra >= c, cbb = c < np && rb < np && rb >= c;     // triangular: rows >= column
            p[k] = (ca ? a0[k] : 0.0) * x0 + (cbb ? a1[k] : 0.0) * x1;
        }
        int colw;
        const double sum = wave_sum_cols<32>(p, lane, colw);
        const int c = c0 + 32 * wv + colw;
        if (!(lane & 1) && c < np) sweep_add(&xv[gd[c]], sum);
        sweep_signal(nullptr, xfin);
    }
}

// X = L11^-1 beyond its 128 x 128 diagonal blocks (which k_diag_block leaves in place), by recursive doubling: at block
// size bs = 128, 256, 512, ... the inverse of every aligned 2 bs block [[A, 0], [C, B]] of L11 is completed from the
// inverses XA, XB of its halves,  X_BA = -XB (C XA):
//   PHASE 0:  T = C XA  (into Xtmp, at the place of X_BA);   PHASE 1:  X_BA = -XB T.
// One workgroup per 128 x 64 tile of the result; fp64 MFMA with the product formed transposed (stores along rows).
// Entries above the diagonal of X are never read as data: triangular operands are masked while they are staged.
// Tile TM x TN per workgroup (four waves, 2 x 2): 128 x 64 where a level has many fronts, 64 x 32 at the top of the tree
// (a handful of fronts: more, shorter workgroups -- the K loop of one tile is the critical path there).
template <int PHASE, int TM, int TN>
__global__ void __launch_bounds__(256)
k_xinv(FrontDev fd, const int* __restrict__ level_nodes, int first, int bs) {
    const int t = level_nodes[first + blockIdx.y];
    const int np = fd.npiv[t];
    const int rts = bs / TM, cts = bs / TN;
    int lin = blockIdx.x;
    const int ct = lin % cts; lin /= cts;
    const int rt = lin % rts;
    const int pair = lin / rts;
    const int a0 = 2 * pair * bs, b0 = a0 + bs;
    if (b0 >= np) return;
    const int mB = min(bs, np - b0);
    const int r0 = rt * TM;
    if (r0 >= mB) return;
    const int c0 = ct * TN;
    const int nf = fd.nf[t], ldx = ldx_of(np);
    const double* F = fd.P + fd.poff[t];
    double* X = fd.X + fd.xoff[t];
    double* T = fd.Xtmp + fd.xoff[t];
    // XA is lower triangular: its rows k < c0 vanish in the columns >= c0;  XB likewise: columns k > row vanish
    const int k_lo = PHASE == 0 ? (c0 & ~15) : 0;
    const int k_hi = PHASE == 0 ? bs : min(mB, r0 + TM);
    constexpr int KC = 16, SA = TM + 16, SB = TN + 16;
    constexpr int GA = 256 / TM, QA = KC / GA;            // staging of A: row lr, k = kq + GA q
    constexpr int QB = TN / 16;                           // staging of B: k = kb, column cb + 16 q
    constexpr int MR = TM / 32, NC = TN / 32;             // 16 x 16 MFMA blocks per wave: rows, columns
    __shared__ double sA[2][KC][SA];                      // sA[.][k][r]: the operand whose rows are the result's rows
    __shared__ double sB[2][KC][SB];                      // sB[.][k][c]: the operand whose columns are the result's columns
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int lr = tid % TM, kq = tid / TM;
    const int kb = tid & 15, cb = tid >> 4;
    const bool rok = r0 + lr < mB;
    double pa[QA], pb[QB];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < QA; ++q) {
            const int k = k0 + kq + GA * q;
            if (PHASE == 0) pa[q] = (rok && k < k_hi) ? F[(b0 + r0 + lr) + (size_t)ldp_of(nf) * (a0 + k)] : 0.0;
            else pa[q] = (rok && k < k_hi && k <= r0 + lr) ? X[(b0 + r0 + lr) + (size_t)ldx * (b0 + k)] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            const int k = k0 + kb, c = c0 + cb + 16 * q;
            if (PHASE == 0) pb[q] = (k < k_hi && k >= c) ? X[(a0 + k) + (size_t)ldx * (a0 + c)] : 0.0;
            else pb[q] = (k < k_hi) ? T[(b0 + k) + (size_t)ldx * (a0 + c)] : 0.0;
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int q = 0; q < QA; ++q) sA[buf][kq + GA * q][lr] = pa[q];
#pragma unroll
        for (int q = 0; q < QB; ++q) sB[buf][kb][cb + 16 * q] = pb[q];
    };
    mfma_d4 acc[NC][MR];
#pragma unroll
    for (int jc = 0; jc < NC; ++jc)
#pragma unroll
        for (int ir = 0; ir < MR; ++ir) acc[jc][ir] = (mfma_d4){0.0, 0.0, 0.0, 0.0};
    fetch(k_lo);
    stash(0);
    __syncthreads();
    int cur = 0;
    const int wr = (wv & 1) * (TM / 2), wc = (wv >> 1) * (TN / 2);
    for (int k0 = k_lo; k0 < k_hi; k0 += KC) {
        const bool more = k0 + KC < k_hi;
        if (more) fetch(k0 + KC);
#pragma unroll
        for (int kk = 0; kk < KC; kk += 4) {
            double ac[NC], br[MR];
#pragma unroll
            for (int jc = 0; jc < NC; ++jc) ac[jc] = sB[cur][kk + l4][wc + 16 * jc + l15];
#pragma unroll
            for (int ir = 0; ir < MR; ++ir) br[ir] = sA[cur][kk + l4][wr + 16 * ir + l15];
#pragma unroll
            for (int jc = 0; jc < NC; ++jc)
#pragma unroll
                for (int ir = 0; ir < MR; ++ir) acc[jc][ir] = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[jc], br[ir], acc[jc][ir], 0, 0, 0);
        }
        if (more) stash(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    double* dst = PHASE == 0 ? T : X;
#pragma unroll
    for (int jc = 0; jc < NC; ++jc)
#pragma unroll
        for (int ir = 0; ir < MR; ++ir)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int c = c0 + wc + 16 * jc + l4 + 4 * reg;     // D[i = l4 + 4 reg -> column][j = l15 -> row]
                const int r = r0 + wr + 16 * ir + l15;
                if (r < mB) dst[(b0 + r) + (size_t)ldx * (a0 + c)] = PHASE == 0 ? acc[jc][ir][reg] : -acc[jc][ir][reg];
            }
}

// W = L21 X, written over L21 (option "sweep_w": the one-launch form of the wide levels' sweeps, k_sweep_fwd_w / k_sweep_bwd_w).  Row r
// of W depends on row r of L21 only, so a workgroup owns 64 boundary rows and walks the column blocks from the left:
//   W[r][c0 .. c0 + 127] = sum_{k >= c0} L21[r][k] X[k][c0 .. c0 + 127]      (X lower triangular),
// block c0 reads the entries of its rows in the columns >= c0 and then overwrites the columns c0 .. c0 + 127 -- which no later block
// reads.  fp64 MFMA, both operands staged through LDS 16 columns of k at a time (the scheme of k_xinv); runs on stream3 behind the
// level's k_xinv launches, beside the factorisation of the levels above.
constexpr int WT_M = 64, WT_N = 128;
__global__ void __launch_bounds__(256, 2)
k_w_inplace(FrontDev fd, const int* __restrict__ level_nodes, int first) {
    const int t = level_nodes[first + blockIdx.y];
    const int np = fd.npiv[t], nf = fd.nf[t], nb = nf - np;
    const int r0 = blockIdx.x * WT_M;
    if (np == 0 || r0 >= nb) return;
    const int ldp = ldp_of(nf), ldx = ldx_of(np);
    double* L = fd.P + fd.poff[t] + np;                   // L21: rows np.. of the pivot columns
    const double* X = fd.X + fd.xoff[t];
    constexpr int KC = 16, SA = WT_M + 16, SB = WT_N + 16;
    constexpr int QA = KC / 4, QB = WT_N / 16;
    __shared__ double sA[2][KC][SA];                      // sA[.][k][r] = L21[r0 + r][k]
    __shared__ double sB[2][KC][SB];                      // sB[.][k][c] = X[k][c0 + c]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int lr = tid & 63, kq = tid >> 6;
    const int kb = tid & 15, cb = tid >> 4;
    const bool rok = r0 + lr < nb;
    const double* arow = L + min(r0 + lr, nb - 1);
    const int wr = (wv & 1) * 32, wc = (wv >> 1) * 64;
    for (int c0 = 0; c0 < np; c0 += WT_N) {
        double pa[QA], pb[QB];
        auto fetch = [&](int k0) {
#pragma unroll
            for (int q = 0; q < QA; ++q) {
                const int k = k0 + kq + 4 * q;
                const double v = arow[(size_t)ldp * min(k, np - 1)];
                pa[q] = (rok && k < np) ? v : 0.0;
            }
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                const int k = k0 + kb, c = c0 + cb + 16 * q;
                pb[q] = (k < np && c < np && k >= c) ? X[k + (size_t)ldx * c] : 0.0;
            }
        };
        auto stash = [&](int buf) {
#pragma unroll
            for (int q = 0; q < QA; ++q) sA[buf][kq + 4 * q][lr] = pa[q];
#pragma unroll
            for (int q = 0; q < QB; ++q) sB[buf][kb][cb + 16 * q] = pb[q];
        };
        mfma_d4 acc[4][2];
#pragma unroll
        for (int jc = 0; jc < 4; ++jc)
#pragma unroll
            for (int ir = 0; ir < 2; ++ir) acc[jc][ir] = (mfma_d4){0.0, 0.0, 0.0, 0.0};
        fetch(c0);
        stash(0);
        __syncthreads();
        int cur = 0;
        for (int k0 = c0; k0 < np; k0 += KC) {
            const bool more = k0 + KC < np;
            if (more) fetch(k0 + KC);
#pragma unroll
            for (int kk = 0; kk < KC; kk += 4) {
                double ac[4], br[2];
#pragma unroll
                for (int jc = 0; jc < 4; ++jc) ac[jc] = sB[cur][kk + l4][wc + 16 * jc + l15];
#pragma unroll
                for (int ir = 0; ir < 2; ++ir) br[ir] = sA[cur][kk + l4][wr + 16 * ir + l15];
#pragma unroll
                for (int jc = 0; jc < 4; ++jc)
#pragma unroll
                    for (int ir = 0; ir < 2; ++ir) acc[jc][ir] = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[jc], br[ir], acc[jc][ir], 0, 0, 0);
            }
            if (more) stash(cur ^ 1);
            __syncthreads();                              // after the last pass: every wave has read the block's own columns
            cur ^= 1;
        }
#pragma unroll
        for (int jc = 0; jc < 4; ++jc)
#pragma unroll
            for (int ir = 0; ir < 2; ++ir)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int c = c0 + wc + 16 * jc + l4 + 4 * reg;     // D[i = l4 + 4 reg -> column][j = l15 -> row]
                    const int r = r0 + wr + 16 * ir + l15;
                    if (r < nb && c < np) L[r + (size_t)ldp * c] = acc[jc][ir][reg];
                }
    }
}

}  // namespace femo
