// libfemo_hip.so: C ABI (include/femo_hip.h) over the gfx950 shell kernels.
// One context = one mesh on one GPU with every buffer resident in HBM.
#include "../../include/femo_hip.h"
#include "shell_device.h"
#include "frontal.h"
#include "sweeps_multi.h"
#include "shape_sens.h"
#include "stress.h"
#include "csr_map.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

using namespace femo;

// tree levels whose largest pivot block exceeds this run the triangular solves with the wide (multi-workgroup)
// kernels and the precomputed 128 x 128 diagonal-block inverses; smaller fronts use one workgroup per front
constexpr int WIDE_NP_DEFAULT = 512;    // option "wide_np" overrides it for the plan being uploaded (tests force the wide path on small meshes)
constexpr int WIDE_CNT_DEFAULT = 512;   // option "wide_cnt": levels with at most this many fronts also take the wide (many workgroups per
                                        // front) solve kernels -- one workgroup per front cannot pull a level's factor out of HBM

#define FEMO_VERSION 100

static std::string g_create_error;

struct femo_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int krylov = 0;                          // 0: conjugate gradients, 1: BiCGStab (femo_set_krylov)
    double* bi[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // BiCGStab work vectors (allocated on first use)
    int tab_nq = 0;                          // quadrature points of the operator's tables (c->tab)
    double hK_mean = 0;                      // mean cell diameter (the yardstick of a change of uhat, option "stale_factor")
    // multi-right-hand-side solves (femo_solve_linear_multi): two interleaved buffers of 4 ndof doubles for the sweeps, 5 work vectors
    // per right-hand side (x, r, z, p, Ap), 8 device scalars per right-hand side; allocated on first use
    double *mr_v = nullptr, *mr_y = nullptr, *mr_work = nullptr, *mr_scal = nullptr, *mr_scal_host = nullptr;
    double* mr_io = nullptr;                 // right-hand sides, solutions and gradients of the multi entry points (grown on demand, kept)
    size_t mr_io_cap = 0;
    long long opt_version = 0;               // bumped by every femo_set_option
    hipStream_t stream2 = nullptr;           // look-ahead: the bulk of a trailing update runs beside the next panel
    hipEvent_t ev_la[2] = {nullptr, nullptr};
    hipEvent_t ev_sp[2] = {nullptr, nullptr};
    hipStream_t stream_g = nullptr;          // second stream of the levels whose fronts are dealt to two streams (option "split_cnt")
    hipEvent_t ev_g[2] = {nullptr, nullptr};
    hipStream_t stream_m = nullptr;          // diagonal look-ahead: a stream that may not use the CUs reserved for the diagonal blocks
    hipEvent_t ev_da[2] = {nullptr, nullptr};
    hipStream_t stream3 = nullptr;           // L11^-1 of a finished level is formed beside the factorisation of the next ones
    hipStream_t stream_a = nullptr;          // option "sweep_ahead": the forward sweep of the lower levels beside the factorisation of the top
    hipEvent_t ev_a[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_x[2] = {nullptr, nullptr};
    int nn = 0, nel = 0, nvc = 0, npc = 0, nP2 = 0, ndof_u = 0, ndof = 0, ld = 0;   // ndof = vector length = mesh DOFs + nghost
    int nghost = 0;
    bool quad = true, ewm = false, ewp = false, has_uhat = false;
    bool cr = false;                         // element CG2CR1 (linear_shell_model.py:68-73, triangles): rotation on the edge midpoints (Crouzeix-Raviart)
    int nrot = 0;                            // rotation nodes: nn, or the number of edges for CG2CR1
    int* frnode = nullptr; double* fMR = nullptr;   // CG2CR1: rotation nodes and 3 x 3 rotation blocks of the penalty facets
    bool cg1 = false;                        // element CG1CG1 (linear_shell_model.py:74-79): displacement on the vertices too (nP2 == nn)
    int64_t nT = 0, nF = 0;
    // mesh
    double* xyz = nullptr;
    int* cells = nullptr;
    int* cellp2 = nullptr;
    int* eorder = nullptr;      // elements along a Morton curve of their centroids (locality of gathers)
    int *n2e_off = nullptr, *n2e_ent = nullptr;   // inverted connectivity: P2 node -> (slot in Morton order) * npc + local node
    double* ybuf = nullptr;     // element results of the operator, YSTRIDE doubles per slot
    double* hK = nullptr;
    Tables* tab = nullptr;
    Tables* tab_s = nullptr;     // degree-4 rule of the p-norm stress measure (3x3 Gauss on quads)
    Tables* tab_pre = nullptr;   // rule of the front assembly when the operator's has more than 4 x 4 points (option "precond_nquad"); null: c->tab
    int tab_pre_nq = 0;
    double stress_m = 1e-6, stress_rho = 100.0, stress_alpha = -1.0, stress_reg = 0.0;
    int* ctag = nullptr;                     // sub-domain index per cell
    int csel = -1, ntags = 0;
    std::vector<double> alpha_tag;           // reference area of every sub-domain (frozen at first use, like stress_alpha)
    double* gradbuf = nullptr;
    double* eq = nullptr;                    // option "equilibrate": the factorisation is that of D K D, D = diag(eq) (allocated on first use)
    double* fp[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // force -> pressure solve: x, r, z, p, Ap, diag (3 nn each, on first use)
    // element-partitioned driver (femo_dist_*): replicated separator entries, dot weights, gradient scatter map
    struct Dist {
        bool ready = false;
        int ntop = 0, nranks = 1, nl = 0, nsel = 0;
        int *top_idx = nullptr, *sel = nullptr;
        double *wdot = nullptr, *topbuf = nullptr, *topsave = nullptr, *gloc = nullptr;
    } di;
    // transient march (femo_newmark_*): history, force history and work vectors resident in HBM
    struct Newmark {
        bool ready = false;
        int levels = 0, flevels = 0;
        double dt = 0, a = 0, b = 0;
        double *W = nullptr, *Fh = nullptr, *wdot = nullptr, *Fsw = nullptr, *mu0 = nullptr, *mu1 = nullptr, *Lam = nullptr, *Gh = nullptr;
        bool has_sw = false;
    } nm;
    // CSR assembly
    long long csr_ncontrib = 0; int csr_nnz = 0;
    int *csr_perm = nullptr, *csr_dest = nullptr, *csr_rowptr = nullptr, *csr_colidx = nullptr;
    double *csr_vals = nullptr, *csr_ke = nullptr;
    double op_aK = 1.0, op_aM = 0.0;      // the operator every solve / factorisation uses: aK * K + aM * M
    int nquad = 4, nred = 0;
    // fields
    double *h = nullptr, *E = nullptr, *nu = nullptr, *rho = nullptr, *f = nullptr, *uhat = nullptr;
    // dirichlet
    int nf = 0;
    int *fcell = nullptr, *fledge = nullptr, *funode = nullptr, *fvnode = nullptr;
    double *fM2 = nullptr, *fM1 = nullptr;
    double beta = 1e15;
    bool penalty_dirty = true;
    double* gdir = nullptr;       // prescribed values g of the penalty term beta/h_E |..| (w - g).v (linear_shell_model.py:323-333); null = zero
    bool has_g = false;
    unsigned char* mask = nullptr;
    bool has_mask = false;
    // vectors
    double *w = nullptr, *lam = nullptr, *r = nullptr, *z = nullptr, *p = nullptr, *Ap = nullptr, *dinv = nullptr,
           *b = nullptr, *tmp = nullptr;
    double* scal = nullptr;        // device, 8 slots
    double* scal_host = nullptr;   // pinned, 8 slots
    bool jacobi_dirty = true;      // the Jacobi diagonal is stale (fields / Dirichlet data / operator changed)
    // schedule switches and failure policy (femo_set_option); never read from the environment
    struct Options {
        int trailing = 0;             // rank-k update schedule: 0 auto, 1 left-looking, 2 right-looking
        int left_min = 64, left_max = 2048;   // auto: levels with this many fronts are left-looking
        int lookahead = 1, lookahead_cnt = 16;
        int super_panel = 512, super_panel_cnt = 64, super_panel_ahead = 0;
        int rows_preload_wg = 0;              // k_panel_rows launches of at most this many workgroups preload S into LDS
        // k_panel_rows launches of at most this many (64-row) workgroups run k_panel_rows_fine (16 rows per workgroup).  Measured at 1 M DOF
        // (scripts/r4_ab.py, one process, interleaved): rows per panel of the two top levels 16 -> 10 us; factorisation 12.73 ms without the
        // fine kernels, 12.61 with this one at 96 or 256, 12.72 at 768 (levels of 8-32 fronts lose), 12.52 with both at (96, 128)
        int rows_fine_wg = 96;
        int narrow_split = 1, narrow_split_wg = 1024;
        int narrow_fine_wg = 128;             // narrow updates of at most this many 64 x 64 tiles run k_trailing_fine (32 x 32 tiles, a 16 x 16 block per wave): the
                                              // root's updates 24 -> 18 us each; levels of four fronts and more lose (512: +5 % on level 11)
        // fronts of non-wide levels with at least fuse_rows_cnt fronts whose widest front has at most fuse_rows_np pivots: the rows under a
        // diagonal block are formed inside k_diag_block.  Up to round 4 only single-panel levels (np <= 128) of >= 4096 fronts took this
        // path, and a handful of leaves with 129-150 pivots sent the whole leaf level through separate row launches (config 2 and 5
        // always, config 3 with the measured bisection: +0.18 ms).  In-process A/B (scripts/r4_ab.py): (128, 4096) -> (256, 2048)
        // factorisation 11.93 -> 11.77 ms at 1 M DOF, 3.235 -> 3.205 on the 255 k plate, 41.6 -> 41.0 at 4 M DOF
        int fuse_rows = 1, fuse_rows_cnt = 2048;
        int fuse_rows_np = 2 * NBO;
        int sweep_graph = 0;                        // the preconditioner application of the PCG loop replayed as a HIP graph
        int diag_v1_cnt = 512;                      // levels of at least this many fronts: k_diag_block (80 KB of LDS, two workgroups per CU)
        int split_cnt = 0, split_groups = 2;        // levels of 2..split_cnt fronts: dealt to two streams in split_groups groups (off: no gain measured)
        int super_tiles = 0, super_tiles_min = 8;   // rank-k updates of few large fronts: 4 x 4 super-tiles per XCD from this many 64-row tiles
        int diag_ahead = 0;                   // super-panel levels: the next diagonal block runs beside the rest of this panel's rows and updates
        int fused_schur = 1;          // left-looking levels: the Schur update gathers its block from the children
        // Experiment: factorise D K D with D = diag(K)^-1/2 (1) or its nearest powers of two (2) and apply
        // D (L L^T)^-1 D.  Measured: no change in what one application of the factor leaves -- every operation of
        // a Cholesky factorisation and of the sweeps commutes with a power-of-two scaling, so mode 2 reproduces mode 0 bit for bit
        int equilibrate = 0;
        // Gauss points per direction of the FRONT ASSEMBLY on quadrilaterals (0: the operator's rule, the default).  The factor is only a
        // preconditioner, so a lighter rule is legitimate -- but it does not pay on thin shells: on the 1 M DOF wing skin (operator
        // 5 x 5) fronts assembled with 4 x 4 points take 4 + 4 PCG iterations instead of 2 + 2 (3 x 3: 6 + 6), i.e. 1.93 -> 1.51 ms of
        // assembly bought with +2.7 ms of forward solve and +3.1 ms of adjoint (profiles/r4_precond_nquad.txt).  Results are the same
        // to 1e-11 either way (PCG iterates on the operator's residual); kept as an option for meshes where assembly dominates.
        int precond_nquad = 0;
        int grid_chunk = 32768;       // fronts per launch along grid y / z (extent limit 65535)
        int wide_np = WIDE_NP_DEFAULT, wide_cnt = WIDE_CNT_DEFAULT;   // read when the plan is uploaded
        int strict = 1;               // a Krylov solve that stops at maxit without reaching rtol is an error
        int allow_pivot_repair = 0;   // non-positive pivots: 0 = the factorisation fails, 1 = replace and count
        int profile_verbose = 0;
        // 128 x 128 tiles for the triangular-grid rank-k updates when a launch has at least big_min_wg of them.  Built, validated (schedule
        // fuzz) and measured SLOWER than the 64 x 64 kernel on every level at 1M DOF (rank-k updates 11.6 against 9.2 ms; levels 6-11:
        // 26-38 against 33-43 TFLOP/s): two waves per SIMD instead of four hide the stage barriers worse, and diagonal / edge tiles waste
        // twice as much.  Off by default.
        int big_tiles = 0, big_min_wg = 512;
        // rank-k updates with K <= strip_kmax on levels of at least strip_cnt fronts: k_schur_strip (a workgroup per 64-row strip of a front,
        // panel rows, maps and metadata once per strip, the next block's operands and child entries in flight behind the current one)
        // Measured (profiles/r5_strip_ab.txt): parity on the leaves, 1.4-1.7 x SLOWER than the tile kernel on the gathering levels 1-5, and more
        // look-ahead makes it worse -- off by default (strip_cnt 0), kept as a validated alternative schedule
        int strip_cnt = 0, strip_kmax = STRIP_KMAX, strip_depth = 1;     // strip_depth: blocks the children's entries are requested ahead (1..2)
        int diag_v1 = 0;              // diagonal-block kernel: 0 auto (see the launch), 1 round-2 kernel (sequential phases), 2 overlapped kernel
        int swork_slots = 8192;       // cap of the diagonal-block scratch (1 GB); larger levels are factorised in chunks (read at plan upload)
        int xinv_small_cnt = 32;      // inversion of L11: levels with at most this many fronts use 64 x 32 tiles
        // backward sweep, L21^T x: levels whose largest boundary has at least this many rows take the tiled (atomic) kernel.
        // Measured at 1M DOF: one workgroup per 32 columns wins on every level (43-57 us against 49-115), so the default is never
        int bnd_tiled_nb = 1 << 30;
        int sweep_butterfly = 3;              // backward sweep column sums in one butterfly: bit 0 k_front_bwd_small, bit 1 k_sweep_bnd_cols
        // wide levels: every run of consecutive wide levels is ONE launch per sweep direction, tiles ordered by per-front counters
        // instead of launch boundaries (k_sweep_wide_fwd / k_sweep_wide_bwd).  Measured at 1 M DOF (profiles/r5_sweep_fuse_ab.txt): the ten
        // wide levels take 517-546 us forward and 437-492 us backward in one launch each against 397 / 380 us as twenty launches -- what a
        // launch boundary costs (~10 us per level and phase) is less than what replaces it: every tile waits for the acknowledgement of its
        // own atomics before it may signal, and a tile in flight holds one of 512 slots for ~17 us instead of ~9.  Off; kept as a validated
        // alternative schedule (schedule fuzz)
        int sweep_fuse = 0;
        // wide levels: W = L21 X takes the place of L21 in the factor store (k_w_inplace, on stream3 behind the inversion of L11), and a
        // wide level is ONE launch per sweep direction: both of its products read the same input (k_sweep_fwd_w / k_sweep_bwd_w).
        // Measured at 1 M DOF (profiles/r5_sweep_w_ab.txt): an application of the preconditioner 1.459 ms against 1.520 -- inside the
        // replayed graph the launch that goes away costs ~3 us, not the ~11 us the event-marked level tables show -- while the 28 GFLOP
        // of W add 1.5 ms to the factorisation and 0.9 ms of waiting to the first sweep: forward 19.10 ms against 16.84.  Off; kept as a
        // validated alternative schedule (schedule fuzz)
        int sweep_w = 0;
        // front assembly with one workgroup per leaf front (k_front_assemble_fc): the front is zeroed, filled and written once, no float
        // atomics (which execute at the memory side on this chip and bound k_front_assemble).  0: one wave per element + k_zero_fronts
        int assemble_fc = 1;
        // Optimisation loops: when only FIELDS changed since the last factorisation (a new thickness), keep that factor as the PCG
        // preconditioner and re-factorise only if the solve has not converged after this many iterations (0: always re-factorise, the
        // default and what the bench measures).  PCG iterates on the CURRENT matrix-free operator, so the answer is the same either way; the
        // reference never refreshes its derivative matrices at all (quirk Q2, csdl_alpha_opt/state_operation.py:130-131)
        // The first preconditioner application of a cold solve starts from z = b, which is known before the factorisation: its forward
        // sweep through the levels [0, nlevels - sweep_ahead) runs on a stream of its own as soon as those levels are factorised,
        // beside the chain of the top levels (31 fronts, 4.4 of 11.5 ms at 1 M DOF, most of the chip idle); the solve joins it before
        // the sweep of the top levels.  0 = off.
        int sweep_ahead = 2;
        int apply_lanes = 0;      // lanes per element of the matrix-free operator (k_apply4): 0 / 4 = a DPP quad; 5 = twelve elements on 60 lanes (25 points: 5 each instead of 7 + 6 + 6 + 6): measured SLOWER, 118.6 against 103.5 us (profiles/r6_apply_lanes.txt)
        int diag_t = 0;           // 1 / 2: classes of fewer than four sub-blocks take k_diag_block_t (LDL / Cholesky elimination): measured slower / equal (profiles/r6_diag_ab.txt)
        int multi_rhs = 1;        // femo_solve_linear_multi / femo_total_gradients: 1 = right-hand sides share the sweeps in groups of up to 4; 0 = one at a time
        int stale_factor = 0;
        // ... and only while no field has moved further than this from the factor's design (relative L2 norm).  Measured at 1 M DOF
        // (profiles/r5_stale_factor.txt): forward + adjoint with the kept factor 12.8 / 15.7 / 18.7 / 24.7 ms at a relative nodal change of
        // 1e-4 / 1e-3 / 3e-3 / 1e-2 (L2: 0.58 of that) against 20.0 ms re-factorised -- the break-even sits at ~2.5e-3
        double stale_rel = 2e-3;
        int sweep_read_mode = 0;              // how a fused sweep reads what other workgroups of the launch wrote: 0 returning atomic, 1 agent-scope load, 2 plain (experiment)
    } opt;
    // solver
    int precond = 0;
    double rtol = 1e-10;
    int maxit = 200000, check_every = 50;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    double timing[5] = {0, 0, 0, 0, 0};
    std::string err;
    // multifrontal preconditioner (precond == 2)
    struct Frontal {
        bool ready = false, factored = false;
        bool have_factor = false;             // a complete factorisation of SOME earlier operator sits in the panel store (option "stale_factor")
        double* snap[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // thickness, E, nu, density, uhat as they were when that factor was made
        bool snap_valid = false;
        bool w_mode = false;                  // the wide levels of the stored factor hold W = L21 X where L21 was (option "sweep_w" at the time of the factorisation)
        double* ahead_vec = nullptr;          // option "sweep_ahead": the vector whose forward sweep the running factorisation starts (null: none)
        int ahead_levels = 0;                 // ... through the levels [0, ahead_levels); set by the factorisation when it did
        bool x_inflight = false;              // k_xinv launches on stream3 that the main stream has not waited for yet (event ev_x[1])
        int ntree = 0, nlevels = 0;
        std::vector<int> h_nf, h_npiv, h_level_off, h_level_nodes, h_level_maxnp, h_level_maxnb;
        std::vector<char> h_level_wide;              // level takes the wide solve kernels (and keeps S in Sinv)
        int *nf = nullptr, *npiv = nullptr, *dofs = nullptr, *upmap = nullptr, *parent = nullptr, *left = nullptr,
            *right = nullptr, *level_nodes = nullptr, *elem_front = nullptr, *elem_map = nullptr, *info = nullptr,
            *cinv0 = nullptr, *cinv1 = nullptr;
        long long *poff = nullptr, *soff = nullptr, *doff = nullptr, *linvoff = nullptr, *xoff = nullptr;
        double *P = nullptr, *S = nullptr, *Linv = nullptr, *X = nullptr, *Xtmp = nullptr, *Swork = nullptr;
        // fused sweeps of the wide levels (option "sweep_fuse"): tiles as tasks of one launch per run of consecutive wide levels
        int* slot_of = nullptr;               // position of every front in level_nodes
        int *fel_off = nullptr, *fel = nullptr;      // elements of every level-0 front, by position in the level (front-centric assembly)
        bool fc_ok = false;                   // every element belongs to a level-0 front
        int* sweep_cnt = nullptr;             // two counters per position (k_sweep_wide_fwd / _bwd)
        SweepTask *ftasks = nullptr, *btasks = nullptr;
        std::vector<long long> h_ft_off;      // forward table (levels ascending): tasks of level L at [h_ft_off[L], h_ft_off[L + 1])
        std::vector<long long> h_bt_begin, h_bt_end;   // backward table (levels descending): tasks of level L at [begin, end)
        hipGraphExec_t sweep_graph = nullptr; // one preconditioner application on c->z, captured (option "sweep_graph")
        long long sweep_graph_key = -1;       // options version the graph was captured under
        std::vector<long long> h_soff;        // host copy of the Schur offsets (femo_front_schur_get / block_set)
        long long p_doubles = 0, s_doubles = 0, linv_doubles = 0, x_doubles = 0;
        int swork_slots = 1;                  // 128 x 128 scratch blocks for the diagonal-block inverses of the non-wide levels
        int max_nf = 0;
        double t_factor_ms = 0, t_assemble_ms = 0;
        int pivots_fixed = 0;
        bool profile = false;                 // time every kernel class with HIP events (slower)
        // kernel classes: 0 rows below the diagonal blocks, 1 diagonal blocks, 2 trailing updates, 3 extend-add, 4 front assembly,
        // 5 zero fill, 6 inversion of L11 (k_xinv)
        double prof_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        long long prof_calls[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        double prof_flops[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // algorithmic flops of what the launches of a class execute (lower triangles only)
        double prof_bytes[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // compulsory HBM bytes of those launches (every operand entry once, results read + written)
        std::vector<hipEvent_t> pev;
        std::vector<double> pmeta;            // per profiled launch group: algorithmic flops, compulsory bytes (rank-k updates only)
        double last_fl = 0, last_by = 0;
        int cur_level = 0;
    } fr;
};

#define HIPCHK(ctx, call)                                                                        \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            char buf_[512];                                                                      \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            (ctx)->err = buf_;                                                                   \
            return 1;                                                                            \
        }                                                                                        \
    } while (0)

static int fail(femo_ctx* c, const std::string& msg) {
    c->err = msg;
    return 2;
}

// whatever the operator A = aK K + aM M (+ Dirichlet treatment) depends on has changed: both preconditioners are stale
static void operator_changed(femo_ctx* c, bool fields_only = false);
static int snapshot_fields(femo_ctx* c);
static int frontal_fwd(femo_ctx* c, double* v, int l0, int l1, std::vector<hipEvent_t>* marks = nullptr);

// ------------------------------------------------------------------------------------------ tables
static void gauss_legendre(int n, double* x, double* w) {
    // The rules the element kernels use (n <= 6) as decimal literals, i.e. the CORRECTLY ROUNDED doubles of the exact nodes and
    // weights.  Not pedantry: at BASELINE config 3 (1 M DOF, 1.27 mm skin) replacing the 5-point weights by values that differ in
    // the last place (<= 4e-16: this routine's own Newton iteration below against numpy's leggauss) moves the displacement, the
    // compliance and d compliance / d thickness by 3.5e-7, 2.8e-7 and 3.5e-7 (tests/golden/make_config3_golden.py with either
    // table; a weight error is the same relative stiffness error in every cell, and the thin skin amplifies it).  The oracle keeps
    // its own copy of the same literals, so the two sides integrate with bit-identical tables.
    static const double GX[7][6] = {{0}, {0.0},
        {-0.5773502691896257645091488, 0.5773502691896257645091488},
        {-0.7745966692414833770358531, 0.0, 0.7745966692414833770358531},
        {-0.8611363115940525752239465, -0.3399810435848562648026658, 0.3399810435848562648026658, 0.8611363115940525752239465},
        {-0.9061798459386639927976269, -0.5384693101056830910363144, 0.0, 0.5384693101056830910363144, 0.9061798459386639927976269},
        {-0.9324695142031520278123015545, -0.661209386466264513661399595, -0.2386191860831969086305017217, 0.2386191860831969086305017217,
         0.661209386466264513661399595, 0.9324695142031520278123015545}};
    static const double GW[7][6] = {{0}, {2.0}, {1.0, 1.0},
        {0.5555555555555555555555556, 0.8888888888888888888888889, 0.5555555555555555555555556},
        {0.3478548451374538573730639, 0.6521451548625461426269361, 0.6521451548625461426269361, 0.3478548451374538573730639},
        {0.236926885056189087514264, 0.4786286704993664680412915, 0.5688888888888888888888889, 0.4786286704993664680412915,
         0.236926885056189087514264},
        {0.1713244923791703450402961422, 0.3607615730481386075698335138, 0.467913934572691047389870344, 0.467913934572691047389870344,
         0.3607615730481386075698335138, 0.1713244923791703450402961422}};
    if (n >= 1 && n <= 6) {
        for (int i = 0; i < n; ++i) { x[i] = GX[n][i]; w[i] = GW[n][i]; }
        return;
    }
    // Newton on Legendre polynomials
    for (int i = 0; i < n; ++i) {
        double t = cos(M_PI * (i + 0.75) / (n + 0.5));
        for (int it = 0; it < 100; ++it) {
            double p0 = 1.0, p1 = t;
            for (int k = 2; k <= n; ++k) {
                const double p2 = ((2 * k - 1) * t * p1 - (k - 1) * p0) / k;
                p0 = p1;
                p1 = p2;
            }
            const double dp = n * (t * p1 - p0) / (t * t - 1.0);
            const double dt = p1 / dp;
            t -= dt;
            if (fabs(dt) < 1e-16) break;
        }
        double p0 = 1.0, p1 = t;
        for (int k = 2; k <= n; ++k) {
            const double p2 = ((2 * k - 1) * t * p1 - (k - 1) * p0) / k;
            p0 = p1;
            p1 = p2;
        }
        const double dp = n * (t * p1 - p0) / (t * t - 1.0);
        x[n - 1 - i] = t;
        w[n - 1 - i] = 2.0 / ((1.0 - t * t) * dp * dp);
    }
}

static void lag2(double t, double* v, double* d) {
    v[0] = 0.5 * t * (t - 1.0); v[1] = 1.0 - t * t; v[2] = 0.5 * t * (t + 1.0);
    d[0] = t - 0.5; d[1] = -2.0 * t; d[2] = t + 0.5;
}
static void lag1(double t, double* v, double* d) {
    v[0] = 0.5 * (1.0 - t); v[1] = 0.5 * (1.0 + t);
    d[0] = -0.5; d[1] = 0.5;
}

// Fully symmetric rules on the unit triangle, orbit by orbit (scripts/derive_triangle_rules.py: Newton's iteration on the moment
// equations in 60-digit arithmetic; every coordinate is its own correctly rounded literal, and oracle/rm_shell_oracle.py::_TRI holds
// the same numbers -- the table is part of the discrete problem).  S21 rows {a, b = 1 - 2a, w} give the points
// (a,a) (b,a) (a,b); S111 rows {a, b, c = 1 - a - b, w} give (a,b) (b,a) (a,c) (c,a) (b,c) (c,b); the weights sum to one.
//   degree  4,  6 points: the reference's p-norm stress measure (quadrature_degree 4, rm_shell_model.py:200-205; basix 0.5.0 takes the
//                         Xiao-Gimbutas table of that degree on a simplex, the same six points)
//   degree  6, 12 points: exact for the static forms on affine cells with uniform E, nu and uhat = 0 (integrand of degree <= 5)
//   degree  9, 19 points: what UFL estimates for the static forms on triangles (plain dx, linear_shell_model.py:88-103;
//                         scripts/ufl_degree_estimate.py) -- the rule when E / nu vary over a cell or the mesh moves (uhat != 0)
//   degree 12, 33 points: the convergence check beyond it
// Returns the number of points (0: no such rule); P[q] = {x, y, weight}.
static int triangle_rule(int degree, double (*P)[3]) {
    static const double S21_4[2][3] = {{0.4459484909159648863183293, 0.1081030181680702273633415, 0.2233815896780114656950070},
                                       {0.09157621350977074345957146, 0.8168475729804585130808571, 0.1099517436553218676383263}};
    static const double S21_6[2][3] = {{0.06308901449150222834033160, 0.8738219710169955433193368, 0.05084490637020681692093681},
                                       {0.2492867451709104212916386, 0.5014265096581791574167229, 0.1167862757263793660252896}};
    static const double S111_6[1][4] = {{0.05314504984481694735324967, 0.3103524510337844054166077, 0.6365024991213986472301426,
                                         0.08285107561837357519355346}};
    static const double S3_9 = 0.09713579628279883381924198;
    static const double S21_9[4][3] = {{0.4896825191987376277837069, 0.02063496160252474443258615, 0.03133470022713907053685483},
                                       {0.4370895914929366372699304, 0.1258208170141267254601393, 0.07782754100477427931673936},
                                       {0.1882035356190327302409613, 0.6235929287619345395180774, 0.07964773892721025303289177},
                                       {0.04472951339445270986510659, 0.9105409732110945802697868, 0.02557767565869803126167880}};
    static const double S111_9[1][4] = {{0.03683841205473628363481760, 0.2219629891607656956751025, 0.7411985987844980206900799,
                                         0.04328353937728937728937729}};
    static const double S21_12[5][3] = {{0.4882173897738048825646621, 0.02356522045239023487067587, 0.02573106644045533541779092},
                                        {0.4397243922944602729797366, 0.1205512154110794540405268, 0.04369254453803840213545726},
                                        {0.2712103850121159223459513, 0.4575792299757681553080973, 0.06285822421788510035427051},
                                        {0.1275761455415859246738963, 0.7448477089168281506522073, 0.03479611293070894298932840},
                                        {0.02131735045321037024685698, 0.9573652990935792595062860, 0.006166261051559017233866484}};
    static const double S111_12[3][4] = {
        {0.1153434945346979991690112, 0.2757132696855141939747963, 0.6089432357797878068561924, 0.04037155776638092951782870},
        {0.02283833222225702961023378, 0.2813255809899395482481307, 0.6958360867878034221416355, 0.02235677320230344571183908},
        {0.02573405054833022816810924, 0.1162519159075971412413541, 0.8580140335440726305905366, 0.01731623110865889237164210}};
    const double(*s21)[3] = nullptr;
    const double(*s111)[4] = nullptr;
    int n21 = 0, n111 = 0, n = 0;
    switch (degree) {
    case 4: s21 = S21_4; n21 = 2; break;
    case 6: s21 = S21_6; n21 = 2; s111 = S111_6; n111 = 1; break;
    case 9: s21 = S21_9; n21 = 4; s111 = S111_9; n111 = 1; break;
    case 12: s21 = S21_12; n21 = 5; s111 = S111_12; n111 = 3; break;
    default: return 0;
    }
    auto put = [&](double x, double y, double w) { P[n][0] = x; P[n][1] = y; P[n][2] = w; ++n; };
    if (degree == 9) put(1.0 / 3.0, 1.0 / 3.0, S3_9);
    for (int i = 0; i < n21; ++i) {
        const double a = s21[i][0], b = s21[i][1], w = s21[i][2];
        put(a, a, w); put(b, a, w); put(a, b, w);
    }
    for (int i = 0; i < n111; ++i) {
        const double a = s111[i][0], b = s111[i][1], cc = s111[i][2], w = s111[i][3];
        put(a, b, w); put(b, a, w); put(a, cc, w); put(cc, a, w); put(b, cc, w); put(cc, b, w);
    }
    return n;
}

// nred > 0 (quads): the membrane / bending / shear energies are integrated with nred x nred Gauss points, everything
// else with nquad x nquad: the table then lists both point sets, each with a zero weight for the terms of the other
static void build_tables(bool quad, int nquad, Tables& T, int nred = 0, bool cg1 = false, bool cr = false) {
    memset(&T, 0, sizeof T);
    if (quad) {
        static const int Q2I[9][2] = {{0, 0}, {2, 0}, {2, 2}, {0, 2}, {1, 0}, {2, 1}, {1, 2}, {0, 1}, {1, 1}};
        static const int Q1I[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
        double gx[8], gw[8];
        int q0 = 0;
        for (int pass = (nred > 0 ? 0 : 1); pass < 2; ++pass) {
        const int n1 = pass == 0 ? nred : nquad;
        gauss_legendre(n1, gx, gw);
        for (int i = 0; i < n1; ++i)
            for (int j = 0; j < n1; ++j) {
                const int q = q0 + i * n1 + j;
                const double xi = gx[i], eta = gx[j];
                T.w[q] = pass == 1 ? gw[i] * gw[j] : 0.0;
                T.wS[q] = (pass == 0 || nred <= 0) ? gw[i] * gw[j] : 0.0;
                double a[3], da[3], b[3], db[3], c[2], dc[2], d[2], dd[2];
                lag2(xi, a, da); lag2(eta, b, db); lag1(xi, c, dc); lag1(eta, d, dd);
                for (int n = 0; n < 9; ++n) {
                    const int ii = Q2I[n][0], jj = Q2I[n][1];
                    T.N2[q][n] = a[ii] * b[jj];
                    T.dN2[q][n][0] = da[ii] * b[jj];
                    T.dN2[q][n][1] = a[ii] * db[jj];
                }
                for (int n = 0; n < 4; ++n) {
                    const int ii = Q1I[n][0], jj = Q1I[n][1];
                    T.N1[q][n] = c[ii] * d[jj];
                    T.dN1[q][n][0] = dc[ii] * d[jj];
                    T.dN1[q][n][1] = c[ii] * dd[jj];
                }
            }
        q0 += n1 * n1;
        }
        T.nq = q0;
    } else {
        // nquad is the DEGREE of the symmetric rule on triangles: 4 / 6 / 9 / 12 (triangle_rule)
        double P[MAXQ][3];
        const int npts = triangle_rule(nquad, P);
        const double dL[3][2] = {{-1, -1}, {1, 0}, {0, 1}};
        static const int ED[3][2] = {{0, 1}, {1, 2}, {2, 0}};
        T.nq = npts;
        for (int q = 0; q < npts; ++q) {
            const double x = P[q][0], y = P[q][1];
            T.w[q] = 0.5 * P[q][2];
            T.wS[q] = T.w[q];
            const double L[3] = {1 - x - y, x, y};
            for (int i = 0; i < 3; ++i) {
                T.N1[q][i] = L[i];
                T.dN1[q][i][0] = dL[i][0];
                T.dN1[q][i][1] = dL[i][1];
                T.N2[q][i] = L[i] * (2 * L[i] - 1);
                T.dN2[q][i][0] = (4 * L[i] - 1) * dL[i][0];
                T.dN2[q][i][1] = (4 * L[i] - 1) * dL[i][1];
            }
            for (int k = 0; k < 3; ++k) {
                const int i = ED[k][0], j = ED[k][1];
                T.N2[q][3 + k] = 4 * L[i] * L[j];
                T.dN2[q][3 + k][0] = 4 * (L[i] * dL[j][0] + L[j] * dL[i][0]);
                T.dN2[q][3 + k][1] = 4 * (L[i] * dL[j][1] + L[j] * dL[i][1]);
            }
        }
    }
    if (cg1) {
        // CG1CG1: the displacement is interpolated with the vertex functions -- its tables are those of the rotation
        const int nv = quad ? 4 : 3;
        for (int q = 0; q < T.nq; ++q)
            for (int n = 0; n < 9; ++n) {
                T.N2[q][n] = n < nv ? T.N1[q][n] : 0.0;
                T.dN2[q][n][0] = n < nv ? T.dN1[q][n][0] : 0.0;
                T.dN2[q][n][1] = n < nv ? T.dN1[q][n][1] : 0.0;
            }
    }
    // the rotation's tables: the vertex functions, or (CG2CR1, triangles) the Crouzeix-Raviart functions NR_k = 1 - 2 lambda_(k+2) of the
    // midpoint of edge k (vertex k -> k + 1): one on its own edge's midpoint, zero on the other two
    for (int q = 0; q < T.nq; ++q)
        for (int n = 0; n < 4; ++n) {
            if (cr && !quad && n < 3) {
                const int o = (n + 2) % 3;
                T.NR[q][n] = 1.0 - 2.0 * T.N1[q][o];
                T.dNR[q][n][0] = -2.0 * T.dN1[q][o][0];
                T.dNR[q][n][1] = -2.0 * T.dN1[q][o][1];
            } else {
                T.NR[q][n] = T.N1[q][n];
                T.dNR[q][n][0] = T.dN1[q][n][0];
                T.dNR[q][n][1] = T.dN1[q][n][1];
            }
        }
}

// ------------------------------------------------------------------------------------------ launch helpers
static inline int nblk(int64_t n, int bs) { return (int)((n + bs - 1) / bs); }
static inline int vec_grid(int64_t n) { return (int)std::min<int64_t>((n + 255) / 256, 2048); }
// reductions end in one atomic per workgroup on a single address (~13 ns each, serialised): keep the grid at one
// workgroup per CU
static inline int red_grid(int64_t n) { return (int)std::min<int64_t>((n + 255) / 256, 256); }

static MeshDev mesh_dev(const femo_ctx* c) {
    MeshDev m;
    m.nn = c->nn; m.nel = c->nel; m.nP2 = c->nP2; m.ndof_u = c->ndof_u; m.ndof = c->ndof;
    m.xyz = c->xyz; m.cells = c->cells; m.cellp2 = c->cellp2; m.hK = c->hK;
    m.ctag = c->ctag; m.csel = c->csel; m.cr = c->cr ? 1 : 0;
    return m;
}
static FieldsDev fields_dev(const femo_ctx* c) {
    FieldsDev f;
    f.h = c->h; f.E = c->E; f.nu = c->nu; f.rho = c->rho; f.f = c->f; f.uhat = c->uhat;
    f.ewm = c->ewm; f.ewp = c->ewp;
    return f;
}
static FacetDev facet_dev(const femo_ctx* c) {
    FacetDev fd;
    fd.nf = c->nf; fd.cell = c->fcell; fd.ledge = c->fledge; fd.unode = c->funode; fd.vnode = c->fvnode;
    fd.M2 = c->fM2; fd.M1 = c->fM1; fd.rnode = c->frnode; fd.MR = c->fMR;
    return fd;
}

// KERNEL is a template <NPC,NVC,QUAD,UHAT>; EXTRA may carry more template args (leading comma).  Every element kernel is instantiated
// for the reference's own element choice CG2CG1 (rm_shell_pde.py:27; NPC = 9 / 6) and for CG1CG1 (NPC == NVC: the displacement lives
// on the vertices, linear_shell_model.py:74-79).
#define ELEM_LAUNCH(c, KERNEL, EXTRA, grid, block, ...)                                                  \
    do {                                                                                                      \
        if ((c)->cg1) {                                                                                       \
        if ((c)->quad) {                                                                                      \
            if ((c)->has_uhat) hipLaunchKernelGGL((KERNEL<4, 4, true, true EXTRA>), dim3(grid), dim3(block), 0, (c)->stream, __VA_ARGS__); \
            else hipLaunchKernelGGL((KERNEL<4, 4, true, false EXTRA>), dim3(grid), dim3(block), 0, (c)->stream, __VA_ARGS__);              \
        } else {                                                                                              \
            if ((c)->has_uhat) hipLaunchKernelGGL((KERNEL<3, 3, false, true EXTRA>), dim3(grid), dim3(block), 0, (c)->stream, __VA_ARGS__); \
            else hipLaunchKernelGGL((KERNEL<3, 3, false, false EXTRA>), dim3(grid), dim3(block), 0, (c)->stream, __VA_ARGS__);              \
        }                                                                                                     \
        } else {                                                                                              \
        if ((c)->quad) {                                                                                      \
            if ((c)->has_uhat) hipLaunchKernelGGL((KERNEL<9, 4, true, true EXTRA>), dim3(grid), dim3(block), 0, (c)->stream, __VA_ARGS__); \
            else hipLaunchKernelGGL((KERNEL<9, 4, true, false EXTRA>), dim3(grid), dim3(block), 0, (c)->stream, __VA_ARGS__);              \
        } else {                                                                                              \
            if ((c)->has_uhat) hipLaunchKernelGGL((KERNEL<6, 3, false, true EXTRA>), dim3(grid), dim3(block), 0, (c)->stream, __VA_ARGS__); \
            else hipLaunchKernelGGL((KERNEL<6, 3, false, false EXTRA>), dim3(grid), dim3(block), 0, (c)->stream, __VA_ARGS__);              \
        }                                                                                                     \
        }                                                                                                     \
    } while (0)
#define ELEM_LAUNCH_S(c, KERNEL, EXTRA, grid, block, shm, ...)                                           \
    do {                                                                                                      \
        if ((c)->cg1) {                                                                                       \
        if ((c)->quad) {                                                                                      \
            if ((c)->has_uhat) hipLaunchKernelGGL((KERNEL<4, 4, true, true EXTRA>), dim3(grid), dim3(block), shm, (c)->stream, __VA_ARGS__); \
            else hipLaunchKernelGGL((KERNEL<4, 4, true, false EXTRA>), dim3(grid), dim3(block), shm, (c)->stream, __VA_ARGS__);              \
        } else {                                                                                              \
            if ((c)->has_uhat) hipLaunchKernelGGL((KERNEL<3, 3, false, true EXTRA>), dim3(grid), dim3(block), shm, (c)->stream, __VA_ARGS__); \
            else hipLaunchKernelGGL((KERNEL<3, 3, false, false EXTRA>), dim3(grid), dim3(block), shm, (c)->stream, __VA_ARGS__);              \
        }                                                                                                     \
        } else {                                                                                              \
        if ((c)->quad) {                                                                                      \
            if ((c)->has_uhat) hipLaunchKernelGGL((KERNEL<9, 4, true, true EXTRA>), dim3(grid), dim3(block), shm, (c)->stream, __VA_ARGS__); \
            else hipLaunchKernelGGL((KERNEL<9, 4, true, false EXTRA>), dim3(grid), dim3(block), shm, (c)->stream, __VA_ARGS__);              \
        } else {                                                                                              \
            if ((c)->has_uhat) hipLaunchKernelGGL((KERNEL<6, 3, false, true EXTRA>), dim3(grid), dim3(block), shm, (c)->stream, __VA_ARGS__); \
            else hipLaunchKernelGGL((KERNEL<6, 3, false, false EXTRA>), dim3(grid), dim3(block), shm, (c)->stream, __VA_ARGS__);              \
        }                                                                                                     \
        }                                                                                                     \
    } while (0)
#define QPOINT_LDS(c) ((size_t)(c)->tab_nq * ((c)->cg1 ? ((c)->quad ? sizeof(QPoint<4, 4>) : sizeof(QPoint<3, 3>)) \
                                                        : ((c)->quad ? sizeof(QPoint<9, 4>) : sizeof(QPoint<6, 3>))))
#define NOEXTRA
#define COMMA_H , DERIV_H
#define COMMA_E , DERIV_E
#define COMMA_NU , DERIV_NU

static const int EB = 128;   // element kernels: threads per block (one element per thread)

// fields_only: a material / thickness / geometry field was re-uploaded -- the old factor no longer belongs to the operator, but it is still the
// factor of a NEARBY symmetric positive definite operator on the same pattern and may serve as the PCG preconditioner (option "stale_factor");
// any other change (Dirichlet data, operator coefficients, quadrature, scaling) discards it
static void operator_changed(femo_ctx* c, bool fields_only) {
    c->jacobi_dirty = true; c->fr.factored = false;
    if (!fields_only) c->fr.have_factor = false;
}

static int refresh_penalty(femo_ctx* c) {
    if (c->nf == 0 || !c->penalty_dirty) return 0;
    ELEM_LAUNCH(c, k_penalty_setup, NOEXTRA, nblk(c->nf, 64), 64, mesh_dev(c), fields_dev(c), facet_dev(c), c->beta);
    HIPCHK(c, hipGetLastError());
    c->penalty_dirty = false;
    return 0;
}

// y = K_elastic x (+ penalty): element pass into ybuf, then one gather-sum per node (no atomics, fixed order)
static int op_apply(femo_ctx* c, const double* x, double* y, double* dotslot, double* za, double* zb, bool with_penalty,
                    double aK = 1.0, double aM = 0.0) {
    {
        const int lanes = c->opt.apply_lanes == 5 ? 5 : 4;
        const int nb = ((nblk(c->nel, apply_epb(lanes)) + 7) / 8) * 8;      // multiple of 8 for the XCD-aware block order
#define COMMA_TRUE , true
#define COMMA_FALSE , false
#define COMMA_TRUE_4 , true, 4
#define COMMA_FALSE_4 , false, 4
#define COMMA_TRUE_5 , true, 5
#define COMMA_FALSE_5 , false, 5
        if (lanes == 5) {
            if (aM != 0.0) ELEM_LAUNCH(c, k_apply4, COMMA_TRUE_5, nb, 256, mesh_dev(c), fields_dev(c), c->tab, c->eorder, aK, aM, x, c->ybuf, dotslot, za, zb);
            else ELEM_LAUNCH(c, k_apply4, COMMA_FALSE_5, nb, 256, mesh_dev(c), fields_dev(c), c->tab, c->eorder, aK, aM, x, c->ybuf, dotslot, za, zb);
        } else {
            if (aM != 0.0) ELEM_LAUNCH(c, k_apply4, COMMA_TRUE_4, nb, 256, mesh_dev(c), fields_dev(c), c->tab, c->eorder, aK, aM, x, c->ybuf, dotslot, za, zb);
            else ELEM_LAUNCH(c, k_apply4, COMMA_FALSE_4, nb, 256, mesh_dev(c), fields_dev(c), c->tab, c->eorder, aK, aM, x, c->ybuf, dotslot, za, zb);
        }
        const int nthreads = c->nP2 + c->nghost;
#define GATHER_SUM(NPC_, NVC_) hipLaunchKernelGGL((k_gather_sum<NPC_, NVC_>), dim3(nblk(nthreads, 256)), dim3(256), 0, c->stream, c->nP2, c->nn, \
                                                  c->ndof_u, c->ndof, c->n2e_off, c->n2e_ent, c->ybuf, y, c->cr ? 1 : 0, c->nrot)
        if (c->cg1) { if (c->quad) GATHER_SUM(4, 4); else GATHER_SUM(3, 3); }
        else if (c->quad) GATHER_SUM(9, 4);
        else GATHER_SUM(6, 3);
#undef GATHER_SUM
    }
    if (with_penalty && c->nf > 0) {
        if (refresh_penalty(c)) return 1;
        hipLaunchKernelGGL(k_penalty_apply, dim3(nblk(c->nf, 64)), dim3(64), 0, c->stream, facet_dev(c), c->ndof_u, 0, x, y,
                           dotslot);
    }
    HIPCHK(c, hipGetLastError());
    return 0;
}

static int refresh_diag(femo_ctx* c) {
    if (!c->jacobi_dirty) return 0;
    const int64_t n = c->ndof;
    hipLaunchKernelGGL(k_fill, dim3(vec_grid(n)), dim3(256), 0, c->stream, c->dinv, 0.0, n);
    ELEM_LAUNCH(c, k_diag, NOEXTRA, nblk(c->nel, EB), EB, mesh_dev(c), fields_dev(c), c->tab, c->dinv);
    if (c->nf > 0) {
        if (refresh_penalty(c)) return 1;
        hipLaunchKernelGGL(k_penalty_apply, dim3(nblk(c->nf, 64)), dim3(64), 0, c->stream, facet_dev(c), c->ndof_u, 1,
                           (const double*)nullptr, c->dinv, (double*)nullptr);
    }
    hipLaunchKernelGGL(k_invert_diag, dim3(vec_grid(n)), dim3(256), 0, c->stream, c->dinv,
                       c->has_mask ? c->mask : (const unsigned char*)nullptr, n);
    HIPCHK(c, hipGetLastError());
    c->jacobi_dirty = false;
    return 0;
}

static int load_vector_dev(femo_ctx* c, double* F, const double* f_override = nullptr) {
    const int64_t n = c->ndof;
    hipLaunchKernelGGL(k_fill, dim3(vec_grid(n)), dim3(256), 0, c->stream, F, 0.0, n);
    FieldsDev fdv = fields_dev(c);
    if (f_override) fdv.f = const_cast<double*>(f_override);        // a level of the resident force history (femo_newmark_*)
    ELEM_LAUNCH(c, k_load, NOEXTRA, nblk(c->nel, EB), EB, mesh_dev(c), fdv, c->tab, F, 1.0);
    if (c->has_g && c->nf > 0) {
        // penalty with prescribed values: R = ... + P (w - g)  ->  the right-hand side gains P g
        if (refresh_penalty(c)) return 1;
        hipLaunchKernelGGL(k_penalty_apply, dim3(nblk(c->nf, 64)), dim3(64), 0, c->stream, facet_dev(c), c->ndof_u, 0, (const double*)c->gdir, F,
                           (double*)nullptr);
    }
    if (c->has_mask) hipLaunchKernelGGL(k_mask_zero, dim3(vec_grid(n)), dim3(256), 0, c->stream, F, c->mask, n);
    HIPCHK(c, hipGetLastError());
    return 0;
}

// common exit of the Krylov solvers: report iterations / residual; stopping at maxit short of rtol is an error unless the
// caller opted out (option "strict" = 0).  The reference solves with a direct LU (fea/utils_dolfinx.py:466,514-531), so an
// unconverged state or adjoint has no counterpart there and must not flow silently into gradients.
static int finish_solve(femo_ctx* c, const char* who, int k, double rr, double bb, double target, int32_t* iters, double* relres) {
    const double rel = bb > 0 ? sqrt(rr / bb) : 0.0;
    if (iters) *iters = k;
    if (relres) *relres = rel;
    if (bb > 0 && !(rr <= target) && c->opt.strict) {
        char buf[256];
        snprintf(buf, sizeof buf, "%s did not converge: relative residual %.3e after %d iterations (rtol %.1e, maxit %d)", who, rel, k,
                 c->rtol, c->maxit);
        c->err = buf;
        return 4;
    }
    return 0;
}

// Jacobi-preconditioned CG on the device; b is overwritten only on masked rows (set to zero)
static int pcg(femo_ctx* c, double* b, double* x, bool zero_guess, int32_t* iters, double* relres) {
    if (c->op_aM != 0.0 || c->op_aK != 1.0) return fail(c, "the Jacobi solver only handles the static operator; use preconditioner 2");
    const int64_t n = c->ndof;
    const int vg = vec_grid(n);
    const unsigned char* mask = c->has_mask ? c->mask : nullptr;
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    if (refresh_diag(c)) return 1;
    HIPCHK(c, hipMemsetAsync(c->scal, 0, 8 * sizeof(double), c->stream));
    if (mask) hipLaunchKernelGGL(k_mask_zero, dim3(vg), dim3(256), 0, c->stream, b, mask, n);
    hipLaunchKernelGGL(k_dot, dim3(red_grid(n)), dim3(256), 0, c->stream, b, b, n, c->scal + 6);
    hipLaunchKernelGGL(k_fill, dim3(vg), dim3(256), 0, c->stream, c->Ap, 0.0, n);
    if (zero_guess) {
        hipLaunchKernelGGL(k_fill, dim3(vg), dim3(256), 0, c->stream, x, 0.0, n);
    } else {
        if (mask) hipLaunchKernelGGL(k_mask_zero, dim3(vg), dim3(256), 0, c->stream, x, mask, n);
        if (op_apply(c, x, c->Ap, nullptr, nullptr, nullptr, true, c->op_aK, c->op_aM)) return 1;
    }
    hipLaunchKernelGGL(k_pcg_init, dim3(vg), dim3(256), 0, c->stream, b, c->Ap, c->dinv, mask, c->r, c->z, c->p, n, c->scal,
                       zero_guess ? 0 : 1);
    HIPCHK(c, hipMemcpyAsync(c->scal_host, c->scal, 8 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const double bb = c->scal_host[6];
    double rr = c->scal_host[4];
    int k = 0;
    int napply = 0;
    if (bb == 0.0) {
        hipLaunchKernelGGL(k_fill, dim3(vg), dim3(256), 0, c->stream, x, 0.0, n);
        rr = 0.0;
    } else {
        const double target = c->rtol * c->rtol * bb;
        while (rr > target && k < c->maxit) {
            const int chunk = std::min(c->check_every, c->maxit - k);
            for (int j = 0; j < chunk; ++j, ++k) {
                const int s = k & 1;
                // K1 also clears the rz/rr slots the update kernel of this iteration accumulates into
                if (op_apply(c, c->p, c->Ap, c->scal + s, c->scal + 2 + (1 - s), c->scal + 4 + (1 - s), true, c->op_aK, c->op_aM)) return 1;
                hipLaunchKernelGGL(k_pcg_update, dim3(vg), dim3(256), 0, c->stream, x, c->r, c->z, c->p, c->Ap, c->dinv, mask, n,
                                   c->scal, s);
                hipLaunchKernelGGL(k_pcg_direction, dim3(vg), dim3(256), 0, c->stream, c->p, c->z, c->Ap, n, c->scal, s);
                ++napply;
            }
            HIPCHK(c, hipMemcpyAsync(c->scal_host, c->scal, 8 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            rr = c->scal_host[4 + (k & 1)];   // after iteration k-1 (s = (k-1)&1) the fresh value sits in slot 1-s = k&1
            if (!(rr == rr)) return fail(c, "PCG broke down (NaN residual)");
        }
    }
    HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float t_setup = 0, t_loop = 0;
    hipEventElapsedTime(&t_setup, c->ev[0], c->ev[1]);
    hipEventElapsedTime(&t_loop, c->ev[1], c->ev[2]);
    c->timing[0] = t_setup; c->timing[1] = t_loop; c->timing[2] = t_setup + t_loop; c->timing[3] = 0; c->timing[4] = napply;
    return finish_solve(c, "Jacobi-PCG", k, rr, bb, c->rtol * c->rtol * bb, iters, relres);
}


// ------------------------------------------------------------------------------------------ multifrontal driver
static FrontDev front_dev(const femo_ctx* c) {
    FrontDev fd;
    fd.ntree = c->fr.ntree; fd.nf = c->fr.nf; fd.npiv = c->fr.npiv; fd.poff = c->fr.poff; fd.soff = c->fr.soff; fd.doff = c->fr.doff;
    fd.dofs = c->fr.dofs; fd.upmap = c->fr.upmap; fd.parent = c->fr.parent; fd.child[0] = c->fr.left; fd.child[1] = c->fr.right;
    fd.linvoff = c->fr.linvoff; fd.xoff = c->fr.xoff; fd.P = c->fr.P; fd.S = c->fr.S; fd.Linv = c->fr.Linv; fd.X = c->fr.X; fd.Xtmp = c->fr.Xtmp;
    fd.cinv[0] = c->fr.cinv0; fd.cinv[1] = c->fr.cinv1;
    return fd;
}

// event pair around one launch group when profiling
struct ProfScope {
    femo_ctx* c; int cls; hipStream_t s; hipEvent_t a = nullptr, b = nullptr;
    ProfScope(femo_ctx* c_, int cls_, hipStream_t s_ = nullptr) : c(c_), cls(cls_), s(s_ ? s_ : c_->stream) {
        if (!c->fr.profile) return;
        hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, s);
    }
    ~ProfScope() {
        if (!c->fr.profile) return;
        hipEventRecord(b, s);
        c->fr.pev.push_back(a); c->fr.pev.push_back(b); c->fr.pev.push_back((hipEvent_t)(intptr_t)(cls + 16 * c->fr.cur_level));
        c->fr.pmeta.push_back(cls == 2 ? c->fr.last_fl : 0.0); c->fr.pmeta.push_back(cls == 2 ? c->fr.last_by : 0.0);
        c->fr.last_fl = c->fr.last_by = 0;
    }
};

// grid y / z extents are limited to 65535: levels with more fronts are launched in chunks
// (option "grid_chunk" lowers the chunk so that the tests reach this path on small meshes)
#define FOR_FRONT_CHUNKS(cnt_, off_, n_) for (int off_ = 0, gc_ = std::max(1, c->opt.grid_chunk), n_ = std::min((cnt_), gc_); off_ < (cnt_); off_ += gc_, n_ = std::min((cnt_) - off_, gc_))

// L11^-1 is formed on stream3 beside the factorisation of the levels above and is first needed by a triangular sweep: the
// main stream joins stream3 lazily (before the first wide level of the next sweep, or before the fronts are rewritten)
static int join_xinv(femo_ctx* c) {
    if (c->fr.x_inflight) {
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_x[1], 0));
        c->fr.x_inflight = false;
    }
    return 0;
}

// levels [l0, l1) of the elimination tree; assemble != 0 first zeroes the fronts and sums the element matrices in
static int frontal_factorize_range(femo_ctx* c, int l0, int l1, bool assemble) {
    auto& fr = c->fr;
    if (!fr.ready) return fail(c, "no frontal plan: call femo_set_frontal_plan first");
    if (l0 < 0 || l1 > fr.nlevels || l0 > l1) return fail(c, "bad level range");
    if (assemble) for (int i = 0; i < 8; ++i) { fr.prof_ms[i] = 0; fr.prof_calls[i] = 0; fr.prof_flops[i] = 0; fr.prof_bytes[i] = 0; }
    bool x_pending = false;
    if (join_xinv(c)) return 1;             // a previous factorisation's inversion must be done before X is written again
    if (l0 == 0) fr.w_mode = c->opt.sweep_w != 0;
    const FrontDev fd = front_dev(c);
    const unsigned char* mask = c->has_mask ? c->mask : nullptr;
    HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
    if (assemble) {
        // option "assemble_fc": 1 = where it was measured to pay (same box, profiles/r5_assemble_fc_ab.txt): elements whose columns and
        // quadrature parts fit ONE wave (triangles, CG1CG1: no barrier between waves), or at least 20 quadrature points per cell (5 x 5 and
        // 6 x 6 rules: +1.8 % at 1 M DOF; with 16 or 9 points the one-wave-per-element kernel is 1 % ahead); 2 = always; 0 = never
        // (the points of the rule the ASSEMBLY runs with: option "precond_nquad" may name a lighter one than the operator's, below)
        const bool lighter = c->quad && c->nred == 0 && c->opt.precond_nquad > 0 && c->opt.precond_nquad < c->nquad;
        const bool fc_pays = assemble_block(c->ld) == 64 || (lighter ? c->opt.precond_nquad * c->opt.precond_nquad : c->tab_nq) >= 20;
        const bool fc = (c->opt.assemble_fc == 2 || (c->opt.assemble_fc == 1 && fc_pays)) && fr.fc_ok && fr.fel_off;
        if (!fc)
        { ProfScope ps(c, 5);
          // only the leaf fronts start from zero (element matrices are added into them); every other front is written
          // entry by entry by the extend-add gather of its level
          const int cnt0 = fr.h_level_off[1] - fr.h_level_off[0];
          int mx = 0;
          for (int i = fr.h_level_off[0]; i < fr.h_level_off[1]; ++i) mx = std::max(mx, fr.h_nf[fr.h_level_nodes[i]]);
          const int nt0 = (mx + TS - 1) / TS;
          FOR_FRONT_CHUNKS(cnt0, off, n)
              hipLaunchKernelGGL(k_zero_fronts, dim3(nt0 * (nt0 + 1) / 2, n), dim3(256), 0, c->stream, fd, fr.level_nodes, off); }
        HIPCHK(c, hipMemsetAsync(fr.info, 0, sizeof(int), c->stream));
        if (refresh_penalty(c)) return 1;
        const double* eq = nullptr;
        if (c->opt.equilibrate) {
            if (c->op_aM != 0.0 || c->op_aK != 1.0 || l1 != fr.nlevels || l0 != 0)
                return fail(c, "option equilibrate: static operator and whole factorisations only (an experiment)");
            if (!c->eq) HIPCHK(c, hipMalloc((void**)&c->eq, (size_t)c->ndof * sizeof(double)));
            c->jacobi_dirty = true;
            if (refresh_diag(c)) return 1;
            hipLaunchKernelGGL(k_eq_scale, dim3(vec_grid(c->ndof)), dim3(256), 0, c->stream, c->eq, (const double*)c->dinv, mask, c->opt.equilibrate, (int64_t)c->ndof);
            eq = c->eq;
        }
        // the rule of the front assembly (option "precond_nquad"): a lighter one than the operator's on quadrilaterals with the plain
        // (un-split) quadrature -- the transient path's reduced strain rule keeps the operator's tables
        const Tables* atab = c->tab;
        int anq = c->tab_nq;
        if (lighter) {
            if (!c->tab_pre || c->tab_pre_nq != c->opt.precond_nquad * c->opt.precond_nquad) {
                Tables TP;                                           // 8 KB on this thread's stack (contexts of several threads factorise at once)
                build_tables(true, c->opt.precond_nquad, TP, 0, c->cg1, c->cr);
                if (!c->tab_pre) HIPCHK(c, hipMalloc((void**)&c->tab_pre, sizeof(Tables)));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                HIPCHK(c, hipMemcpy(c->tab_pre, &TP, sizeof(Tables), hipMemcpyHostToDevice));
                c->tab_pre_nq = TP.nq;
            }
            atab = c->tab_pre; anq = c->tab_pre_nq;
        }
        const size_t qlds = (size_t)anq * (c->cg1 ? (c->quad ? sizeof(QPoint<4, 4>) : sizeof(QPoint<3, 3>)) : (c->quad ? sizeof(QPoint<9, 4>) : sizeof(QPoint<6, 3>)));
        { ProfScope ps(c, 4);
        if (fc) {
            // one workgroup per leaf front: zero fill, element columns and their sums without atomics (k_front_assemble_fc)
            const int cnt0 = fr.h_level_off[1] - fr.h_level_off[0];
            const int ablk = assemble_block(c->ld);
            const size_t alds = assemble_lds(c->ld, qlds);
            if (c->op_aM != 0.0)
                ELEM_LAUNCH_S(c, k_front_assemble_fc, COMMA_TRUE, cnt0, ablk, alds, mesh_dev(c), fields_dev(c), atab, c->op_aK, c->op_aM, fd,
                              (const int*)fr.level_nodes, (const int*)fr.fel_off, (const int*)fr.fel, fr.elem_map, mask, eq);
            else
                ELEM_LAUNCH_S(c, k_front_assemble_fc, COMMA_FALSE, cnt0, ablk, alds, mesh_dev(c), fields_dev(c), atab, c->op_aK, c->op_aM, fd,
                              (const int*)fr.level_nodes, (const int*)fr.fel_off, (const int*)fr.fel, fr.elem_map, mask, eq);
        } else if (c->op_aM != 0.0)
            ELEM_LAUNCH_S(c, k_front_assemble, COMMA_TRUE, c->nel, 64, qlds, mesh_dev(c), fields_dev(c), atab, c->op_aK, c->op_aM, fd, fr.elem_front,
                          fr.elem_map, mask, eq);
        else
            ELEM_LAUNCH_S(c, k_front_assemble, COMMA_FALSE, c->nel, 64, qlds, mesh_dev(c), fields_dev(c), atab, c->op_aK, c->op_aM, fd, fr.elem_front,
                          fr.elem_map, mask, eq); }
        if (c->nf > 0)
            hipLaunchKernelGGL(k_front_penalty, dim3(nblk(c->nf, 64)), dim3(64), 0, c->stream, facet_dev(c), fd, fr.elem_front,
                               fr.elem_map, c->ld, c->npc, c->nvc, mask, eq);
        if (mask) hipLaunchKernelGGL(k_front_mask_diag, dim3(fr.ntree), dim3(64), 0, c->stream, fd, mask);
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
    for (int L = l0; L < l1; ++L) {
        fr.cur_level = L;
        const int b = fr.h_level_off[L], e = fr.h_level_off[L + 1];
        const int cnt = e - b;
        const int* lev = fr.level_nodes + b;
        int max_np = 0;
        for (int i = b; i < e; ++i) max_np = std::max(max_np, fr.h_npiv[fr.h_level_nodes[i]]);
        // Levels above the leaves whose Schur complements receive exactly ONE rank-k update (left-looking levels; right-looking
        // levels of single-panel fronts without look-ahead): that update gathers the block from the children itself, and
        // the extend-add fills the pivot columns only (option "fused_schur").  Measured at 1M DOF: extend-add 3.3 -> 2.3 ms,
        // rank-k updates 8.4 -> 8.7 ms.
        const bool left_level = !(c->opt.trailing == 2 ? true : c->opt.trailing == 1 ? false : (cnt < c->opt.left_min || cnt > c->opt.left_max));
        // the super-panel width the schedule really uses (whole outer panels): every decision below tests THIS value, so an
        // option between two multiples of 128 cannot switch the gathering update on while the super-panel branch stays off
        const int SP_opt = c->opt.super_panel / NBO * NBO;
        const bool sp_level = !left_level && SP_opt > NBO && cnt <= c->opt.super_panel_cnt && max_np > NBO && !c->opt.super_panel_ahead;
        const bool single_update = left_level || (max_np <= NBO && !(cnt < c->opt.lookahead_cnt && c->opt.lookahead != 0));
        // super-panel levels: the FIRST update behind a super-panel reaches every Schur complement of the level: it gathers,
        // the later ones read and write what it stored
        const bool fused_schur = L > 0 && (single_update || sp_level) && c->opt.fused_schur != 0;
        if (L > 0) {
            // children Schur complements into the parents of this level
            int max_nb = 0;
            for (int i = b; i < e; ++i) {
                const int t = fr.h_level_nodes[i];
                max_nb = std::max(max_nb, fr.h_nf[t]);     // child boundary <= parent front
            }
            const int nt = (max_nb + TS - 1) / TS;
            const dim3 grid(nt * (nt + 1) / 2, cnt);
            { ProfScope ps(c, 3);
              FOR_FRONT_CHUNKS(cnt, off, n)
                  hipLaunchKernelGGL(k_extend_gather, dim3(grid.x, n), dim3(256), 0, c->stream, fd, lev, off, mask, fused_schur ? 1 : 0); }
        }
        const bool wide = fr.h_level_wide[L];                      // these levels keep S (inside X) for the triangular solves
        const int cnt_level = cnt, max_np_level = max_np;
        const int* lev_level = lev;
        // the other levels park the inverse of a diagonal block in Swork between k_diag_block and k_panel_rows, one 128 x 128
        // slot per front: levels with more fronts than Swork has slots are factorised in chunks of that many fronts
        // Levels of a handful of large fronts (option "split_cnt": at most that many, at least two): the fronts are dealt to TWO streams in
        // "split_groups" contiguous groups (fronts are sorted by size: groups alternate between the streams).  Each stream runs the whole
        // chain of its groups -- while one is inside a diagonal block (one workgroup per front: the chip is empty), the other's rows
        // and rank-k updates fill it.  Only wide levels (their diagonal-block inverses go to X, not to the shared Swork slots), and
        // not with the schedules that use the second stream themselves.
        const bool level_right = c->opt.trailing == 2 ? true : c->opt.trailing == 1 ? false : (cnt_level < c->opt.left_min || cnt_level > c->opt.left_max);
        const bool level_sp = level_right && SP_opt > NBO && cnt_level <= c->opt.super_panel_cnt && max_np_level > NBO;
        const bool level_la = level_right && !level_sp && cnt_level < c->opt.lookahead_cnt && c->opt.lookahead != 0;
        const bool split = wide && c->stream_g && cnt_level >= 2 && cnt_level <= c->opt.split_cnt && !level_la &&
                           !(level_sp && (c->opt.super_panel_ahead || c->opt.diag_ahead));
        const int ngroups = split ? std::max(2, std::min(cnt_level, c->opt.split_groups)) : 1;
        const int chunk_cap = split ? (cnt_level + ngroups - 1) / ngroups : wide ? cnt_level : std::min(cnt_level, fr.swork_slots);
        if (split) {
            HIPCHK(c, hipEventRecord(c->ev_g[0], c->stream));                  // the extend-add of the level
            HIPCHK(c, hipStreamWaitEvent(c->stream_g, c->ev_g[0], 0));
        }
        for (int chunk_b = b; chunk_b < fr.h_level_off[L + 1]; chunk_b += chunk_cap) {
        hipStream_t st = (split && ((chunk_b - fr.h_level_off[L]) / chunk_cap) % 2 == 1) ? c->stream_g : c->stream;
        const int b = chunk_b, e = std::min(chunk_b + chunk_cap, fr.h_level_off[L + 1]);
        const int cnt = e - b;
        const int* lev = fr.level_nodes + b;
        int max_np = 0;
        for (int i = b; i < e; ++i) max_np = std::max(max_np, fr.h_npiv[fr.h_level_nodes[i]]);
        int max_nf = 0;
        for (int i = b; i < e; ++i) max_nf = std::max(max_nf, fr.h_nf[fr.h_level_nodes[i]]);
        // outer panels of NBO columns, three launches each: the diagonal block (factor + inverse, one workgroup per
        // front), the rows below it (one GEMM against the inverse), the trailing update
        int max_nb = 0;
        for (int i = b; i < e; ++i) max_nb = std::max(max_nb, fr.h_nf[fr.h_level_nodes[i]] - fr.h_npiv[fr.h_level_nodes[i]]);
        // Rank-k updates, two schedules (option "trailing": 0 auto, 1 left, 2 right):
        //   right-looking -- after every outer panel, everything behind it is updated with that panel's 128 columns;
        //   left-looking  -- a panel's columns receive all earlier columns' updates just before they are factorised and
        //                    the Schur complement is updated once with K = npiv (each entry read and written once).
        // Left-looking wins where a level has enough fronts to fill the chip with the narrow panel updates (measured at
        // 1M DOF: levels of 16..2048 fronts), right-looking at the top of the tree and on the single-panel leaves.
        const bool right_looking = c->opt.trailing == 2 ? true : c->opt.trailing == 1 ? false : (cnt_level < c->opt.left_min || cnt_level > c->opt.left_max);
        const bool use_sp = right_looking && SP_opt > NBO && cnt_level <= c->opt.super_panel_cnt && max_np_level > NBO;
        const bool lookahead = right_looking && !use_sp && cnt_level < c->opt.lookahead_cnt && c->opt.lookahead != 0;
        bool bulk_pending = false;
        // flops of one k_trailing_mfma launch over this level, with the kernel's own column / K ranges (profiling only)
        auto count_trailing = [&](int C0, int schur, int K0 = 0, int KW = NBO) {
            if (!fr.profile) return;
            double fl = 0, by = 0;
            for (int i = b; i < e; ++i) {
                const int t = fr.h_level_nodes[i], np = fr.h_npiv[t], nf = fr.h_nf[t];
                const TrailRange tr = trail_range(schur, C0, K0, KW, np, nf);
                const int kw = tr.kw, col_lo = tr.col_lo, col_hi = tr.col_hi;
                if (kw <= 0 || col_lo >= col_hi) continue;
                const double ncol = col_hi - col_lo;
                const double entries = ncol * nf - 0.5 * ncol * (col_lo + col_hi - 1.0);
                fl += 2.0 * kw * entries;
                by += 16.0 * entries + 8.0 * kw * (nf - col_lo);     // C read + written once, the factor rows [col_lo, nf) x kw read once
            }
            fr.prof_flops[2] += fl; fr.prof_bytes[2] += by;
            fr.last_fl += fl; fr.last_by += by;
        };
        // 64-row tiles a k_trailing_mfma launch needs from its even column anchor down: the largest extent over the chunk's
        // fronts (a grid sized by the largest front alone launches mostly idle workgroups on levels of small fronts)
        auto trail_tiles = [&](int C0, int schur, int K0 = 0, int KW = NBO) {
            int need = 0;
            for (int i = b; i < e; ++i) {
                const int t = fr.h_level_nodes[i], nf = fr.h_nf[t];
                const TrailRange tr = trail_range(schur, C0, K0, KW, fr.h_npiv[t], nf);
                if (tr.kw > 0 && tr.col_lo < tr.col_hi) need = std::max(need, nf - (tr.col_lo & ~1));
            }
            return (need + TS - 1) / TS;
        };
        // Triangular-grid updates (schur 1, 2, 5): 128 x 128 tiles where that still fills the chip (at least "big_min_wg" workgroups
        // over the launch), 64 x 64 tiles otherwise.  `ntr`: 64-row tiles the launch needs (trail_tiles).
        auto launch_tri = [&](bool gather, int ntr, int off, int n, int C0_, int mode, int K0_, int KW_, hipStream_t st) {
            // levels of many small fronts with a short K range (HBM-bound updates): one workgroup per 64-row strip of a front
            // (k_schur_strip; option "strip_cnt": levels of at least that many fronts, 0 = never)
            if (c->opt.strip_cnt > 0 && cnt_level >= c->opt.strip_cnt) {
                int kmax = 0, gx = 0;
                for (int i = b; i < e; ++i) {
                    const int t = fr.h_level_nodes[i], nf = fr.h_nf[t];
                    const TrailRange tr = trail_range(mode, C0_, K0_, KW_, fr.h_npiv[t], nf);
                    if (tr.kw <= 0 || tr.col_lo >= tr.col_hi) continue;
                    kmax = std::max(kmax, tr.kw);
                    const int anchor = tr.col_lo & ~1;
                    gx = std::max(gx, strip_wgs(nf - anchor, tr.col_hi - anchor));
                }
                if (kmax > 0 && kmax <= std::min(c->opt.strip_kmax, STRIP_KMAX)) {
                    const dim3 grid(gx, 1, n);
#define STRIP_LAUNCH(G_, KQ_, D_) hipLaunchKernelGGL((k_schur_strip<G_, KQ_, D_>), grid, dim3(256), 0, st, fd, lev, off, C0_, mode, K0_, KW_, mask)
#define STRIP_DEPTHS(G_, KQ_) do { if (c->opt.strip_depth <= 1) STRIP_LAUNCH(G_, KQ_, 1); else STRIP_LAUNCH(G_, KQ_, 2); } while (0)
                    if (gather) { if (kmax <= 64) STRIP_DEPTHS(true, 16); else if (kmax <= 112) STRIP_DEPTHS(true, 28); else if (kmax <= 128) STRIP_DEPTHS(true, 32); else STRIP_DEPTHS(true, 40); }
                    else        { if (kmax <= 64) STRIP_DEPTHS(false, 16); else if (kmax <= 128) STRIP_DEPTHS(false, 32); else STRIP_DEPTHS(false, 40); }
#undef STRIP_DEPTHS
#undef STRIP_LAUNCH
                    return;
                }
            }
            const int ntb = (ntr + 1) / 2;
            const long long big_wgs = (long long)ntb * (ntb + 1) / 2 * n;
            if (c->opt.big_tiles && big_wgs >= c->opt.big_min_wg) {
                const size_t shm = 4 * sizeof(double) * 16 * LSTRB;
                if (gather) hipLaunchKernelGGL(k_trailing_big<true>, dim3(ntb * (ntb + 1) / 2, 1, n), dim3(256), shm, st, fd, lev, off, C0_, mode, K0_, KW_, mask);
                else hipLaunchKernelGGL(k_trailing_big<false>, dim3(ntb * (ntb + 1) / 2, 1, n), dim3(256), shm, st, fd, lev, off, C0_, mode, K0_, KW_, mask);
            } else {
                // levels of few large fronts: 4 x 4 super-tiles per XCD (k_trailing_mfma, tile_map 1); the grid covers whole rounds of
                // eight super-tiles, the tiles beyond the front leave at once
                const bool sup = c->opt.super_tiles && n < 256 && ntr >= c->opt.super_tiles_min;
                const int nst = (ntr + 3) / 4, nsup = nst * (nst + 1) / 2;
                const int gx = sup ? (nsup + 7) / 8 * 128 : ntr * (ntr + 1) / 2;
                if (gather) hipLaunchKernelGGL(k_trailing_mfma<true>, dim3(gx, 1, n), dim3(256), 0, st, fd, lev, off, C0_, mode, K0_, KW_, mask, 0, sup ? 1 : 0);
                else hipLaunchKernelGGL(k_trailing_mfma<false>, dim3(gx, 1, n), dim3(256), 0, st, fd, lev, off, C0_, mode, K0_, KW_, mask, 0, sup ? 1 : 0);
            }
        };
        // narrow (schur 0) updates of few workgroups: their K range is cut into slices that run side by side and add their
        // products with atomics (options "narrow_split": most slices, "narrow_split_wg": most tiles of a launch that is cut)
        auto narrow_slices = [&](int gx, int n, int K) {
            if (c->opt.narrow_split <= 1 || (long long)gx * n > c->opt.narrow_split_wg) return 1;
            return std::max(1, std::min(c->opt.narrow_split, K / 32));
        };
        // rows below a diagonal block: launches of few workgroups (what one workgroup takes is what the launch takes) use the
        // kernel that brings all of S into LDS at once (option "rows_preload_wg": up to that many workgroups per launch)
        auto launch_rows = [&](int ntiles, int off, int n, int C0_, const double* sw_, int tile_first, hipStream_t st) {
            if ((long long)ntiles * n <= c->opt.rows_fine_wg)
                hipLaunchKernelGGL(k_panel_rows_fine, dim3(4 * ntiles, n), dim3(256), 0, st, fd, lev, off, C0_, sw_, tile_first);
            else if ((long long)ntiles * n <= c->opt.rows_preload_wg)
                hipLaunchKernelGGL(k_panel_rows_preload, dim3(ntiles, n), dim3(256), PANEL_ROWS_PRELOAD_LDS, st, fd, lev, off, C0_, sw_, tile_first);
            else
                hipLaunchKernelGGL(k_panel_rows, dim3(ntiles, n), dim3(256), 0, st, fd, lev, off, C0_, sw_, tile_first);
        };
        // k_diag_block: factor + inverse of the kw x kw block ~ 2/3 kw^3 + 1/3 kw^3; k_panel_rows: rows x triangular S
        auto count_panel = [&](int C0) {
            if (!fr.profile) return;
            for (int i = b; i < e; ++i) {
                const int t = fr.h_level_nodes[i], np = fr.h_npiv[t], nf = fr.h_nf[t];
                if (C0 >= np) continue;
                const double kw = std::min(NBO, np - C0), rows = nf - C0 - kw;
                fr.prof_flops[1] += kw * kw * kw;
                fr.prof_flops[0] += rows * kw * (kw + 1.0);
                fr.prof_bytes[1] += 8.0 * (kw * (kw + 1.0) / 2 + kw * kw);           // block read, factor sub-blocks + inverse written
                fr.prof_bytes[0] += 16.0 * rows * kw + 4.0 * kw * kw;
            }
        };
        // right-looking levels with few, large fronts work in super-panels of SP columns: inside one the panels are
        // updated left-looking, and everything behind it is updated once, with K = SP -- the K = 128 update of the whole
        // trailing matrix is HBM-bound (16 flop per byte moved).  With "super_panel_ahead" that bulk update runs on the second
        // stream beside the NEXT super-panel's chain (diagonal block -> rows -> narrow update, latency-bound at the top
        // of the tree): it leaves the next super-panel's pivot columns out (schur 5), and those columns take the factor
        // columns of both super-panels in their own narrow updates (K0 = start of the previous super-panel).  Measured at
        // 1M DOF: slower than the plain super-panel schedule (19.0 against 18.4 ms) -- the one-workgroup-per-front
        // diagonal-block kernels lose more to sharing their CUs with the bulk update than the overlap gains; off by default.
        const int SP = use_sp ? SP_opt : NBO;
        const bool sp_ahead = use_sp && c->opt.super_panel_ahead != 0;
        int sp_bulks = 0;                                // bulk updates of this chunk issued on stream2 so far
        // Diagonal look-ahead inside a super-panel (option "diag_ahead"): the chain diagonal block -> rows -> narrow update of
        // the next panel -> next diagonal block is what the top of the tree waits for, but the next diagonal block needs only
        // the rows of the NEXT panel (two 64-row tiles) and the three tiles of the narrow update that cover it.  Those stay on
        // the main stream; the other row tiles and the other tiles of the narrow update go to stream_m and run beside the
        // next diagonal block, whose rows wait for them (ev_da[1]).
        // (wide levels only: elsewhere the inverse of a diagonal block sits in the front's ONE Swork slot, which the next diagonal block
        // would overwrite while the rest of this panel's rows still read it)
        const bool diag_ahead = use_sp && !sp_ahead && c->opt.diag_ahead && c->stream_m && wide;
        constexpr int DIAG_TILES = 3, NEXT_ROW_TILES = NBO / TS;
        bool rest_pending = false;
        // single-panel fronts on a level that keeps no S: rows inside the diagonal-block kernel (option "fuse_rows")
        const bool fuse_rows = !wide && max_np_level <= std::max(NBO, c->opt.fuse_rows_np) && c->opt.fuse_rows != 0 && cnt_level >= c->opt.fuse_rows_cnt;
        for (int C0 = 0; C0 < max_np; C0 += NBO) {
            double* sw = wide ? nullptr : fr.Swork;
            const int S0 = C0 / SP * SP;                 // start of this panel's super-panel (== C0 when SP == NBO)
            const int K0 = !right_looking ? 0 : sp_ahead ? std::max(0, S0 - SP) : S0;
            if (sp_ahead && C0 == S0 && sp_bulks >= 2)
                // the bulk update two super-panels back wrote this super-panel's columns
                HIPCHK(c, hipStreamWaitEvent(st, c->ev_sp[sp_bulks & 1], 0));
            if (C0 > K0) {
                // left-looking update of this panel's columns with the factor columns [K0, C0) to their left
                ProfScope ps(c, 2, st);
                count_trailing(C0, 0, K0);
                const int ntr = trail_tiles(C0, 0, K0);
                // fused levels: these pivot columns are touched here for the first time (left-looking: every panel behind
                // the first; super-panel schedule: the panels of the first super-panel) -- gathered from the children
                const bool gather_panel = fused_schur && K0 == 0;
                // diagonal look-ahead: only the tiles of the diagonal block here, the others were issued on stream_m behind the previous rows
                const int gx = diag_ahead ? std::min(DIAG_TILES, ntr * (NBO / TS)) : ntr * (NBO / TS);
                // few tiles (the top of the tree): the fine kernel -- a quarter of the MFMA chain per wave, four times the workgroups
                const bool fine = !diag_ahead && (C0 - K0) % 64 == 0 && (long long)gx * cnt <= c->opt.narrow_fine_wg;
                if (ntr > 0)
                FOR_FRONT_CHUNKS(cnt, off, n) {
                    if (fine) {
                        const int gx32 = 2 * ntr * (NBO / 32);           // 32-row tiles from the even column anchor down, four column tiles
                        if (gather_panel) hipLaunchKernelGGL(k_trailing_fine<true>, dim3(gx32, 1, n), dim3(256), 0, st, fd, lev, off, C0, K0, mask, 0);
                        else hipLaunchKernelGGL(k_trailing_fine<false>, dim3(gx32, 1, n), dim3(256), 0, st, fd, lev, off, C0, K0, mask, 0);
                    } else if (gather_panel) hipLaunchKernelGGL(k_trailing_mfma<true>, dim3(gx, 1, n), dim3(256), 0, st, fd, lev, off, C0, 0, K0, NBO, mask, 0, 0);
                    else hipLaunchKernelGGL(k_trailing_mfma<false>, dim3(gx, narrow_slices(gx, n, C0 - K0), n), dim3(256), 0, st, fd, lev, off, C0, 0, K0, NBO, mask, 0, 0);
                }
            }
            count_panel(C0);
            { ProfScope ps(c, 1, st);
              // classes of equal sub-block count (fronts are sorted by pivot count); small levels go in one launch
              const int* hn = fr.h_level_nodes.data() + b;
              auto first_above = [&](int thr) {          // first position whose front has more than thr pivots
                  int lo = 0, hi = cnt;
                  while (lo < hi) { const int mid = (lo + hi) / 2; if (fr.h_npiv[hn[mid]] > thr) hi = mid; else lo = mid + 1; }
                  return lo;
              };
              int start = first_above(C0);
              for (int nb4 = 1; nb4 <= NBO / NB && start < cnt; ++nb4) {
                  int end = nb4 == NBO / NB ? cnt : first_above(C0 + NB * nb4);
                  int nblk = nb4;
                  if (cnt < 512) { end = cnt; nblk = std::min(NBO / NB, (fr.h_npiv[hn[cnt - 1]] - C0 + NB - 1) / NB); }
                  if (end > start) {
                      // The overlapped kernel (k_diag_block2) needs more LDS per workgroup (the blocks of the inverse and a scratch
                      // block per wave).  On the levels of many small fronts what counts is workgroups per CU, not the latency of one:
                      // classes of fewer than four sub-blocks keep the round-2 kernel there (measured at 1M DOF: levels 0-3 766 /
                      // 261 / 242 / 178 us with the new kernel throughout against 517 / 164 / 192 / 156).  Option "diag_v1": 1 forces
                      // the old kernel everywhere, 2 the new one.
                      const bool v1 = fuse_rows || c->opt.diag_v1 == 1 || (c->opt.diag_v1 == 0 && cnt >= c->opt.diag_v1_cnt) || (c->opt.diag_v1 == 3 && cnt >= 512 && nblk < NBO / NB);
                      if (v1 && c->opt.diag_t && nblk < NBO / NB) {
                          // classes of at most 1 / 2 / 3 sub-blocks: the compile-time-bounded kernel (3-4 workgroups per CU, LDL elimination)
#define DIAG_T(NBLK_, REP_) do { if (c->opt.diag_t == 2) hipLaunchKernelGGL((k_diag_block_t<NBLK_, REP_, false>), dim3(end - start), dim3(256), diag_block_lds_blocks(NBLK_) * sizeof(blk32), st, \
                                               fd, lev, start, C0, sw, fr.info, fuse_rows ? 1 : 0); \
                            else hipLaunchKernelGGL((k_diag_block_t<NBLK_, REP_, true>), dim3(end - start), dim3(256), diag_block_lds_blocks(NBLK_) * sizeof(blk32), st, \
                                               fd, lev, start, C0, sw, fr.info, fuse_rows ? 1 : 0); } while (0)
                          const bool rep = c->opt.allow_pivot_repair != 0;
                          if (nblk == 1) { if (rep) DIAG_T(1, true); else DIAG_T(1, false); }
                          else if (nblk == 2) { if (rep) DIAG_T(2, true); else DIAG_T(2, false); }
                          else { if (rep) DIAG_T(3, true); else DIAG_T(3, false); }
#undef DIAG_T
                      } else if (v1)
                          hipLaunchKernelGGL(k_diag_block, dim3(end - start), dim3(256), diag_block_lds_blocks(nblk) * sizeof(blk32), st,
                                             fd, lev, start, nblk, C0, sw, fr.info, fuse_rows ? 1 : 0);
                      else if (c->opt.allow_pivot_repair)
                          hipLaunchKernelGGL(k_diag_block2<true>, dim3(end - start), dim3(256), diag_block2_lds_blocks(nblk) * sizeof(blk32), st,
                                             fd, lev, start, nblk, C0, sw, fr.info);
                      else
                          hipLaunchKernelGGL(k_diag_block2<false>, dim3(end - start), dim3(256), diag_block2_lds_blocks(nblk) * sizeof(blk32), st,
                                             fd, lev, start, nblk, C0, sw, fr.info);
                  }
                  start = end;
              } }
            int rows_below = 0;                          // rows under the diagonal block, the most over the chunk's fronts
            for (int i = b; i < e; ++i) {
                const int t = fr.h_level_nodes[i], np = fr.h_npiv[t];
                if (np > C0) rows_below = std::max(rows_below, fr.h_nf[t] - C0 - std::min(NBO, np - C0));
            }
            const int tiles = fuse_rows ? 0 : (rows_below + TS - 1) / TS;        // fused: the diagonal-block kernel did the rows
            // the rest of this panel's narrow update ran on stream_m: these rows read what it wrote
            if (rest_pending) { HIPCHK(c, hipStreamWaitEvent(st, c->ev_da[1], 0)); rest_pending = false; }
            // the next panel of the same super-panel takes a narrow update from this one
            const bool ahead = diag_ahead && C0 + NBO < S0 + SP && C0 + NBO < max_np;
            const int tiles_main = ahead ? std::min(tiles, NEXT_ROW_TILES) : tiles;
            if (tiles_main > 0) {
                ProfScope ps(c, 0, st);
                FOR_FRONT_CHUNKS(cnt, off, n)
                    launch_rows(tiles_main, off, n, C0, sw, 0, st);
            }
            if (ahead) {
                HIPCHK(c, hipEventRecord(c->ev_da[0], st));
                HIPCHK(c, hipStreamWaitEvent(c->stream_m, c->ev_da[0], 0));
                if (tiles > tiles_main) {
                    ProfScope ps(c, 0, c->stream_m);
                    FOR_FRONT_CHUNKS(cnt, off, n)
                        launch_rows(tiles - tiles_main, off, n, C0, sw, tiles_main, c->stream_m);
                }
                const int C1 = C0 + NBO;
                const int gx = trail_tiles(C1, 0, K0) * (NBO / TS) - DIAG_TILES;
                if (gx > 0) {
                    ProfScope ps(c, 2, c->stream_m);
                    const bool gather_panel = fused_schur && K0 == 0;
                    FOR_FRONT_CHUNKS(cnt, off, n) {
                        if (gather_panel) hipLaunchKernelGGL(k_trailing_mfma<true>, dim3(gx, 1, n), dim3(256), 0, c->stream_m, fd, lev, off, C1, 0, K0, NBO, mask, DIAG_TILES, 0);
                        else hipLaunchKernelGGL(k_trailing_mfma<false>, dim3(gx, narrow_slices(gx, n, C1 - K0), n), dim3(256), 0, c->stream_m, fd, lev, off, C1, 0, K0, NBO, mask, DIAG_TILES, 0);
                    }
                }
                HIPCHK(c, hipEventRecord(c->ev_da[1], c->stream_m));
                rest_pending = true;
            }
            if (right_looking && SP > NBO) {
                if (C0 + NBO >= S0 + SP || C0 + NBO >= max_np) {
                    // the super-panel is complete: one update of everything behind it
                    const int mode = sp_ahead ? 5 : 2;
                    const int ntr = trail_tiles(S0, mode, 0, SP);
                    if (ntr > 0) {
                        hipStream_t bs = sp_ahead ? c->stream2 : st;
                        if (sp_ahead) {
                            HIPCHK(c, hipEventRecord(c->ev_la[0], st));
                            HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_la[0], 0));
                        }
                        { ProfScope ps(c, 2, bs);
                          count_trailing(S0, mode, 0, SP);
                          FOR_FRONT_CHUNKS(cnt, off, n) launch_tri(fused_schur && S0 == 0, ntr, off, n, S0, mode, 0, SP, bs); }
                        if (sp_ahead) {
                            HIPCHK(c, hipEventRecord(c->ev_sp[sp_bulks & 1], c->stream2));
                            ++sp_bulks;
                        }
                    }
                }
            } else if (right_looking && trail_tiles(C0, 2) > 0) {
                const int ntr = trail_tiles(C0, 2);              // tiles are anchored at an even column, at most one before the first updated one
                if (!lookahead) {
                    ProfScope ps(c, 2, st);
                    count_trailing(C0, 2);
                    // only the FIRST update behind the pivot columns may gather: a later one would overwrite the earlier panels' updates
                    FOR_FRONT_CHUNKS(cnt, off, n) launch_tri(fused_schur && !left_level && C0 == 0, ntr, off, n, C0, 2, 0, NBO, st);
                } else {
                    // look-ahead: the next panel's 128 columns are updated first, on the main stream; everything behind
                    // them goes to the second stream and runs beside the next diagonal block and its rows, which are a
                    // chain of latency-bound launches at the top of the tree.  The next narrow update touches columns
                    // the bulk update also writes, so it waits for it (ev_la[1]).
                    if (bulk_pending) { HIPCHK(c, hipStreamWaitEvent(st, c->ev_la[1], 0)); bulk_pending = false; }
                    { ProfScope ps(c, 2, st);
                      count_trailing(C0, 3);
                      hipLaunchKernelGGL(k_trailing_mfma<false>, dim3(ntr * (NBO / TS), 1, cnt), dim3(256), 0, st, fd, lev, 0, C0, 3, 0, NBO, mask, 0, 0); }
                    const int ntb = trail_tiles(C0, 4);
                    if (ntb > 0) {
                        HIPCHK(c, hipEventRecord(c->ev_la[0], st));
                        HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_la[0], 0));
                        { ProfScope ps(c, 2, c->stream2);
                          count_trailing(C0, 4);
                          hipLaunchKernelGGL(k_trailing_mfma<false>, dim3(ntb * (ntb + 1) / 2, 1, cnt), dim3(256), 0, c->stream2, fd, lev, 0, C0, 4, 0, NBO, mask, 0, 0); }
                        HIPCHK(c, hipEventRecord(c->ev_la[1], c->stream2));
                        bulk_pending = true;
                    }
                }
            }
        }
        if (bulk_pending) { HIPCHK(c, hipStreamWaitEvent(st, c->ev_la[1], 0)); bulk_pending = false; }
        if (rest_pending) { HIPCHK(c, hipStreamWaitEvent(st, c->ev_da[1], 0)); rest_pending = false; }
        if (sp_bulks > 0) HIPCHK(c, hipStreamWaitEvent(st, c->ev_sp[(sp_bulks - 1) & 1], 0));     // stream2 runs its launches in order
        if (max_nb > 0 && !right_looking) {
            // Schur complement: one update with all npiv factor columns
            ProfScope ps(c, 2, st);
            count_trailing(0, 1);
            const int ntr = trail_tiles(0, 1);
            if (ntr > 0)
            FOR_FRONT_CHUNKS(cnt, off, n) launch_tri(fused_schur, ntr, off, n, 0, 1, 0, NBO, st);
        }
        }   // chunks of the level
        if (split) {
            HIPCHK(c, hipEventRecord(c->ev_g[1], c->stream_g));
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_g[1], 0));
        }
        const bool w_level = wide && fr.w_mode && fr.h_level_maxnb[L] > 0;
        if (wide && (max_np_level > NBO || w_level)) {
            const int max_np = max_np_level, cnt = cnt_level;
            const int* lev = lev_level;
            // L11^-1 of this level's fronts beyond the diagonal blocks, on its own stream: nothing in the factorisation of the
            // levels above reads what it reads (the factor columns) or writes (X); only the triangular sweeps need it
            HIPCHK(c, hipEventRecord(c->ev_x[0], c->stream));
            HIPCHK(c, hipStreamWaitEvent(c->stream3, c->ev_x[0], 0));
            ProfScope ps(c, 6, c->stream3);
            const bool small_tiles = cnt <= c->opt.xinv_small_cnt;      // few fronts: 64 x 32 tiles, four times the workgroups
            for (int bs = NBO; bs < max_np; bs *= 2) {
                const int npairs = (max_np + 2 * bs - 1) / (2 * bs);
                FOR_FRONT_CHUNKS(cnt, off, n) {
                    if (small_tiles) {
                        const dim3 grid(npairs * (bs / 64) * (bs / 32), n);
                        hipLaunchKernelGGL((k_xinv<0, 64, 32>), grid, dim3(256), 0, c->stream3, fd, lev, off, bs);
                        hipLaunchKernelGGL((k_xinv<1, 64, 32>), grid, dim3(256), 0, c->stream3, fd, lev, off, bs);
                    } else {
                        const dim3 grid(npairs * (bs / 128) * (bs / 64), n);
                        hipLaunchKernelGGL((k_xinv<0, 128, 64>), grid, dim3(256), 0, c->stream3, fd, lev, off, bs);
                        hipLaunchKernelGGL((k_xinv<1, 128, 64>), grid, dim3(256), 0, c->stream3, fd, lev, off, bs);
                    }
                }
            }
            if (w_level) {
                // W = L21 X over L21: nothing in the factorisation reads L21 of this level again (its Schur complements are complete)
                const int nrt = (fr.h_level_maxnb[L] + WT_M - 1) / WT_M;
                FOR_FRONT_CHUNKS(cnt, off, n) hipLaunchKernelGGL(k_w_inplace, dim3(nrt, n), dim3(256), 0, c->stream3, fd, lev, off);
            }
            x_pending = true;
            if (fr.profile) {                       // profiling: no overlap, so that every class is timed on an otherwise idle chip
                HIPCHK(c, hipEventRecord(c->ev_x[1], c->stream3));
                HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_x[1], 0));
            }
        }
        HIPCHK(c, hipGetLastError());
        if (fr.ahead_vec && L + 1 == fr.nlevels - c->opt.sweep_ahead && l0 == 0 && l1 == fr.nlevels && !fr.profile) {
            // option "sweep_ahead": levels [0, L] are factorised (main stream) and their X is on its way (stream3): the forward sweep of
            // the waiting vector through them starts now, on its own stream, beside the chain of the levels above
            HIPCHK(c, hipEventRecord(c->ev_a[0], c->stream));
            HIPCHK(c, hipStreamWaitEvent(c->stream_a, c->ev_a[0], 0));
            HIPCHK(c, hipEventRecord(c->ev_a[1], c->stream3));
            HIPCHK(c, hipStreamWaitEvent(c->stream_a, c->ev_a[1], 0));
            hipStream_t main_stream = c->stream;
            const bool keep_inflight = fr.x_inflight;
            c->stream = c->stream_a; fr.x_inflight = false;            // (the sweep's own join has nothing to wait for: done above)
            const int rc_a = frontal_fwd(c, fr.ahead_vec, 0, L + 1, nullptr);
            c->stream = main_stream; fr.x_inflight = keep_inflight;
            if (rc_a) return rc_a;
            HIPCHK(c, hipEventRecord(c->ev_a[2], c->stream_a));
            fr.ahead_levels = L + 1;
        }
    }
    if (x_pending) {
        HIPCHK(c, hipEventRecord(c->ev_x[1], c->stream3));
        fr.x_inflight = true;
        if (fr.profile && join_xinv(c)) return 1;
    }
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    int info = 0;
    HIPCHK(c, hipMemcpyAsync(&info, fr.info, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ta = 0, tf = 0;
    hipEventElapsedTime(&ta, c->ev[2], c->ev[3]);
    hipEventElapsedTime(&tf, c->ev[3], c->ev[0]);
    if (assemble) { fr.t_assemble_ms = ta; fr.t_factor_ms = 0; }
    fr.t_factor_ms += tf; fr.pivots_fixed = info;
    fr.factored = (l1 == fr.nlevels);
    fr.have_factor = fr.factored;
    // option "stale_factor": the design this factor belongs to is remembered HERE, on every path that completes a factorisation
    // (pcg_frontal, femo_factorize, femo_factorize_profile, a femo_factorize_range that reaches the root) -- a partial range leaves
    // the panel store a mixture and the snapshot invalid
    if (fr.factored) { if (snapshot_fields(c)) return 1; }
    else fr.snap_valid = false;
    int pivot_rc = 0;
    if (info > 0 && !c->opt.allow_pivot_repair) {
        // the reference's LU would factorise an indefinite matrix; a Cholesky factor of one does not exist, and a silently
        // repaired factor is a preconditioner of something else: say so (option "allow_pivot_repair" restores the old behaviour)
        char buf[200];
        snprintf(buf, sizeof buf, "multifrontal Cholesky: %d non-positive pivot(s) -- the operator is not positive definite "
                 "(negative thickness / modulus, or no Dirichlet data?)", info);
        c->err = buf;
        fr.factored = false; fr.have_factor = false; fr.snap_valid = false;
        pivot_rc = 5;
    }
    for (size_t i = 0; i + 2 < fr.pev.size() + 0 && fr.profile; i += 3) {
        float ms = 0;
        hipEventElapsedTime(&ms, fr.pev[i], fr.pev[i + 1]);
        const int tag = (int)(intptr_t)fr.pev[i + 2];
        const int cls = tag % 16;
        fr.prof_ms[cls] += ms; fr.prof_calls[cls] += 1;
        if (cls == 2) {
            // class 7: the rank-k launches whose arithmetic intensity (algorithmic flops / compulsory bytes) lies above the ridge of
            // the chip, 78.6 TFLOP/s / 8 TB/s = 9.8 flop per byte -- the ones the matrix cores can bound; the rest of class 2 is
            // bounded by HBM whatever the kernel does
            const double fl = fr.pmeta[i / 3 * 2], by = fr.pmeta[i / 3 * 2 + 1];
            if (by > 0 && fl / by >= 78.6e12 / 8.0e12) { fr.prof_ms[7] += ms; fr.prof_calls[7] += 1; fr.prof_flops[7] += fl; fr.prof_bytes[7] += by; }
        }
        if (c->opt.profile_verbose) {
            if (cls == 2) fprintf(stderr, "prof level %d class %d %.1f us flops %.6e bytes %.6e\n", tag / 16, cls, ms * 1e3, fr.pmeta[i / 3 * 2], fr.pmeta[i / 3 * 2 + 1]);
            else fprintf(stderr, "prof level %d class %d %.1f us\n", tag / 16, cls, ms * 1e3);
        }
        hipEventDestroy(fr.pev[i]); hipEventDestroy(fr.pev[i + 1]);
    }
    fr.pev.clear();
    fr.pmeta.clear();
    return pivot_rc;
}

static int frontal_factorize(femo_ctx* c) { return frontal_factorize_range(c, 0, c->fr.nlevels, true); }

// v <- (L L^T)^-1 v   (c->tmp is the scratch vector: forward v -> tmp, backward tmp -> v)
// Wide levels: two matrix-vector products per sweep (X = L11^-1 and L21), accumulated with atomics into entries that the
// memset at the start of the sweep zeroed; the other levels: one workgroup per front.
static int frontal_fwd(femo_ctx* c, double* v, int l0, int l1, std::vector<hipEvent_t>* marks) {
    auto& fr = c->fr;
    const FrontDev fd = front_dev(c);
    double* y = c->tmp;
    auto mark = [&]() { if (marks) { hipEvent_t e; hipEventCreate(&e); hipEventRecord(e, c->stream); marks->push_back(e); } };
    if (l0 == 0) HIPCHK(c, hipMemsetAsync(y, 0, (size_t)c->ndof * sizeof(double), c->stream));
    mark();
    for (int L = l0; L < l1; ++L) {
        const int b = fr.h_level_off[L], cnt = fr.h_level_off[L + 1] - b;
        const int* lev = fr.level_nodes + b;
        const int maxnp = fr.h_level_maxnp[L], maxnb = fr.h_level_maxnb[L];
        if (maxnp == 0) continue;
        if (fr.h_level_wide[L] && c->opt.sweep_fuse && !fr.w_mode) {
            // the run of consecutive wide levels from here: one launch, tiles ordered by per-front counters
            if (join_xinv(c)) return 1;
            int Le = L;
            while (Le < l1 && fr.h_level_wide[Le]) ++Le;
            const long long t0 = fr.h_ft_off[L], t1 = fr.h_ft_off[Le];
            const int s0 = fr.h_level_off[L], s1 = fr.h_level_off[Le];
            if (t1 > t0) {
                HIPCHK(c, hipMemsetAsync(fr.sweep_cnt + 2 * (size_t)s0, 0, (size_t)(s1 - s0) * 2 * sizeof(int), c->stream));
#define FWD_LAUNCH(RM_) hipLaunchKernelGGL(k_sweep_wide_fwd<RM_>, dim3((unsigned)(t1 - t0)), dim3(256), 0, c->stream, fd, (const int*)fr.level_nodes, \
                                   (const SweepTask*)(fr.ftasks + t0), s0, (const int*)fr.slot_of, fr.sweep_cnt, v, y)
                if (c->opt.sweep_read_mode == 0) FWD_LAUNCH(0); else if (c->opt.sweep_read_mode == 1) FWD_LAUNCH(1); else FWD_LAUNCH(2);
#undef FWD_LAUNCH
            }
            for (int k = L; k < Le; ++k) { mark(); mark(); }        // the profile books the whole run under its first level
            L = Le - 1;
            continue;
        }
        if (fr.h_level_wide[L]) {
            if (join_xinv(c)) return 1;
            const int nct = (maxnp + 127) / 128, nrt = (maxnb + 127) / 128;
            if (fr.w_mode) {
                const int nx = nct * (nct + 1) / 2;
                FOR_FRONT_CHUNKS(cnt, off, n)
                    hipLaunchKernelGGL(k_sweep_fwd_w, dim3(nx + nrt * nct, n), dim3(256), 0, c->stream, fd, lev, off, nx, v, y);
                mark(); mark();
                continue;
            }
            FOR_FRONT_CHUNKS(cnt, off, n)
                hipLaunchKernelGGL(k_sweep_gemv_n<true>, dim3(nct * (nct + 1) / 2, n), dim3(256), 0, c->stream, fd, lev, off, (const double*)v, y);
            mark();
            if (maxnb > 0)
                FOR_FRONT_CHUNKS(cnt, off, n)
                    hipLaunchKernelGGL(k_sweep_gemv_n<false>, dim3(nrt * nct, n), dim3(256), 0, c->stream, fd, lev, off, (const double*)y, v);
        } else {
            const size_t shm = (size_t)(maxnp + SMALL_PART) * sizeof(double);
            hipLaunchKernelGGL(k_front_fwd_small, dim3(cnt), dim3(256), shm, c->stream, fd, lev, v, y);
            mark();
        }
        mark();
    }
    HIPCHK(c, hipGetLastError());
    return 0;
}

// backward over levels l1-1 ... l0: y (in tmp) is consumed in place (running right-hand side), x lands in v.  After the
// forward sweep nothing in v is alive (every entry is the pivot of a front behind us), so the sweep from the root starts
// by zeroing it: the wide levels accumulate x with atomics.
static int frontal_bwd(femo_ctx* c, double* v, int l0, int l1, std::vector<hipEvent_t>* marks = nullptr) {
    auto& fr = c->fr;
    const FrontDev fd = front_dev(c);
    double* y = c->tmp;
    auto mark = [&]() { if (marks) { hipEvent_t e; hipEventCreate(&e); hipEventRecord(e, c->stream); marks->push_back(e); } };
    if (l1 == fr.nlevels) HIPCHK(c, hipMemsetAsync(v, 0, (size_t)c->ndof * sizeof(double), c->stream));
    mark();
    for (int L = l1 - 1; L >= l0; --L) {
        const int b = fr.h_level_off[L], cnt = fr.h_level_off[L + 1] - b;
        const int* lev = fr.level_nodes + b;
        const int maxnp = fr.h_level_maxnp[L], maxnb = fr.h_level_maxnb[L];
        if (maxnp == 0) continue;
        if (fr.h_level_wide[L] && c->opt.sweep_fuse && !fr.w_mode) {
            if (join_xinv(c)) return 1;
            int Lb = L;                                            // the run of consecutive wide levels from L down to Lb
            while (Lb > l0 && fr.h_level_wide[Lb - 1]) --Lb;
            const long long t0 = fr.h_bt_begin[L], t1 = fr.h_bt_end[Lb];
            const int s0 = fr.h_level_off[Lb], s1 = fr.h_level_off[L + 1];
            if (t1 > t0) {
                int mxnb = 128;
                for (int k = Lb; k <= L; ++k) mxnb = std::max(mxnb, fr.h_level_maxnb[k]);
                HIPCHK(c, hipMemsetAsync(fr.sweep_cnt + 2 * (size_t)s0, 0, (size_t)(s1 - s0) * 2 * sizeof(int), c->stream));
#define BWD_LAUNCH(RM_) hipLaunchKernelGGL(k_sweep_wide_bwd<RM_>, dim3((unsigned)(t1 - t0)), dim3(256), (size_t)mxnb * sizeof(double), c->stream, fd, \
                                   (const int*)fr.level_nodes, (const SweepTask*)(fr.btasks + t0), s1, (const int*)fr.slot_of, fr.sweep_cnt, y, v)
                if (c->opt.sweep_read_mode == 0) BWD_LAUNCH(0); else if (c->opt.sweep_read_mode == 1) BWD_LAUNCH(1); else BWD_LAUNCH(2);
#undef BWD_LAUNCH
            }
            for (int k = Lb; k <= L; ++k) { mark(); mark(); }
            L = Lb;
            continue;
        }
        if (fr.h_level_wide[L]) {
            if (join_xinv(c)) return 1;
            const int nct = (maxnp + 127) / 128, nrt = (maxnb + 127) / 128;
            if (fr.w_mode) {
                const int nx = nct * (nct + 1) / 2, nbb = maxnb > 0 ? (maxnp + BB_COLS - 1) / BB_COLS : 0;
                FOR_FRONT_CHUNKS(cnt, off, n)
                    if (c->opt.sweep_butterfly & 2)
                        hipLaunchKernelGGL(k_sweep_bwd_w<true>, dim3(nx + nbb, n), dim3(256), (size_t)maxnb * sizeof(double), c->stream, fd, lev, off, nx, (const double*)y, v);
                    else
                        hipLaunchKernelGGL(k_sweep_bwd_w<false>, dim3(nx + nbb, n), dim3(256), (size_t)maxnb * sizeof(double), c->stream, fd, lev, off, nx, (const double*)y, v);
                mark(); mark();
                continue;
            }
            if (maxnb > 0) {
                if (maxnb >= c->opt.bnd_tiled_nb) {
                    FOR_FRONT_CHUNKS(cnt, off, n)
                        hipLaunchKernelGGL(k_sweep_gemv_t<false>, dim3(nrt * nct, n), dim3(256), 0, c->stream, fd, lev, off, (const double*)v, y);
                } else {
                    FOR_FRONT_CHUNKS(cnt, off, n)
                        if (c->opt.sweep_butterfly & 2)
                            hipLaunchKernelGGL(k_sweep_bnd_cols<true>, dim3((maxnp + BB_COLS - 1) / BB_COLS, n), dim3(256), (size_t)maxnb * sizeof(double), c->stream,
                                               fd, lev, off, y, (const double*)v);
                        else
                            hipLaunchKernelGGL(k_sweep_bnd_cols<false>, dim3((maxnp + BB_COLS - 1) / BB_COLS, n), dim3(256), (size_t)maxnb * sizeof(double), c->stream,
                                               fd, lev, off, y, (const double*)v);
                }
            }
            mark();
            FOR_FRONT_CHUNKS(cnt, off, n)
                hipLaunchKernelGGL(k_sweep_gemv_t<true>, dim3(nct * (nct + 1) / 2, n), dim3(256), 0, c->stream, fd, lev, off, (const double*)y, v);
        } else {
            const size_t shm = (size_t)(maxnp + maxnb + SMALL_PART) * sizeof(double);
            if (c->opt.sweep_butterfly & 1) hipLaunchKernelGGL(k_front_bwd_small<true>, dim3(cnt), dim3(256), shm, c->stream, fd, lev, y, v);
            else hipLaunchKernelGGL(k_front_bwd_small<false>, dim3(cnt), dim3(256), shm, c->stream, fd, lev, y, v);
            mark();
        }
        mark();
    }
    HIPCHK(c, hipGetLastError());
    return 0;
}

static int frontal_solve(femo_ctx* c, double* v) {
    const bool eq = c->opt.equilibrate && c->eq;                    // the factor is that of D K D:  K^-1 ~ D (L L^T)^-1 D
    if (eq) hipLaunchKernelGGL(k_mul, dim3(vec_grid(c->ndof)), dim3(256), 0, c->stream, v, (const double*)c->eq, (int64_t)c->ndof);
    if (int rc = frontal_fwd(c, v, 0, c->fr.nlevels)) return rc;
    if (int rc = frontal_bwd(c, v, 0, c->fr.nlevels)) return rc;
    if (eq) hipLaunchKernelGGL(k_mul, dim3(vec_grid(c->ndof)), dim3(256), 0, c->stream, v, (const double*)c->eq, (int64_t)c->ndof);
    return 0;
}

// ---- several right-hand sides through ONE pair of sweeps (sweeps_multi.h): V and Y hold NR interleaved vectors.  Same schedule as
// frontal_fwd / frontal_bwd in its default form (level by level; the fused and the W forms are single-vector experiments).
template <int NR>
static int frontal_solve_multi(femo_ctx* c, double* V, double* Y, std::vector<hipEvent_t>* marks = nullptr) {
    auto mark = [&]() { if (marks) { hipEvent_t e; hipEventCreate(&e); hipEventRecord(e, c->stream); marks->push_back(e); } };
    auto& fr = c->fr;
    const FrontDev fd = front_dev(c);
    const size_t bytes = (size_t)c->ndof * NR * sizeof(double);
    HIPCHK(c, hipMemsetAsync(Y, 0, bytes, c->stream));
    mark();
    for (int L = 0; L < fr.nlevels; ++L) {
        const int b = fr.h_level_off[L], cnt = fr.h_level_off[L + 1] - b;
        const int* lev = fr.level_nodes + b;
        const int maxnp = fr.h_level_maxnp[L], maxnb = fr.h_level_maxnb[L];
        if (maxnp == 0) { mark(); continue; }
        if (fr.h_level_wide[L]) {
            if (join_xinv(c)) return 1;
            const int nct = (maxnp + 127) / 128, nrt = (maxnb + 127) / 128;
            FOR_FRONT_CHUNKS(cnt, off, n)
                hipLaunchKernelGGL((k_sweep_gemv_n_m<true, NR>), dim3(nct * (nct + 1) / 2, n), dim3(256), 0, c->stream, fd, lev, off, (const double*)V, Y);
            if (maxnb > 0)
                FOR_FRONT_CHUNKS(cnt, off, n)
                    hipLaunchKernelGGL((k_sweep_gemv_n_m<false, NR>), dim3(nrt * nct, n), dim3(256), 0, c->stream, fd, lev, off, (const double*)Y, V);
        } else {
            const size_t shm = (size_t)(maxnp + SMALL_PART) * NR * sizeof(double);
            hipLaunchKernelGGL(k_front_fwd_small_m<NR>, dim3(cnt), dim3(256), shm, c->stream, fd, lev, V, Y);
        }
        mark();
    }
    HIPCHK(c, hipMemsetAsync(V, 0, bytes, c->stream));
    mark();
    for (int L = fr.nlevels - 1; L >= 0; --L) {
        const int b = fr.h_level_off[L], cnt = fr.h_level_off[L + 1] - b;
        const int* lev = fr.level_nodes + b;
        const int maxnp = fr.h_level_maxnp[L], maxnb = fr.h_level_maxnb[L];
        if (maxnp == 0) { mark(); continue; }
        if (fr.h_level_wide[L]) {
            if (join_xinv(c)) return 1;
            const int nct = (maxnp + 127) / 128, nrt = (maxnb + 127) / 128;
            if (maxnb > 0) {
                if (maxnb >= c->opt.bnd_tiled_nb || (size_t)maxnb * NR * sizeof(double) > 60 * 1024) {
                    FOR_FRONT_CHUNKS(cnt, off, n)
                        hipLaunchKernelGGL((k_sweep_gemv_t_m<false, NR>), dim3(nrt * nct, n), dim3(256), 0, c->stream, fd, lev, off, (const double*)V, Y);
                } else {
                    FOR_FRONT_CHUNKS(cnt, off, n)
                        hipLaunchKernelGGL(k_sweep_bnd_cols_m<NR>, dim3((maxnp + BB_COLS - 1) / BB_COLS, n), dim3(256), (size_t)maxnb * NR * sizeof(double),
                                           c->stream, fd, lev, off, Y, (const double*)V);
                }
            }
            FOR_FRONT_CHUNKS(cnt, off, n)
                hipLaunchKernelGGL((k_sweep_gemv_t_m<true, NR>), dim3(nct * (nct + 1) / 2, n), dim3(256), 0, c->stream, fd, lev, off, (const double*)Y, V);
        } else {
            const size_t shm = (size_t)(maxnp + maxnb + SMALL_PART) * NR * sizeof(double);
            hipLaunchKernelGGL(k_front_bwd_small_m<NR>, dim3(cnt), dim3(256), shm, c->stream, fd, lev, (const double*)Y, V);
        }
        mark();
    }
    HIPCHK(c, hipGetLastError());
    return 0;
}

static double* mr_io_buffer(femo_ctx* c, size_t doubles) {
    if (doubles > c->mr_io_cap) {
        if (c->mr_io) { hipStreamSynchronize(c->stream); hipFree(c->mr_io); c->mr_io = nullptr; c->mr_io_cap = 0; }
        if (hipMalloc((void**)&c->mr_io, doubles * sizeof(double)) != hipSuccess) { c->err = "out of device memory"; return nullptr; }
        c->mr_io_cap = doubles;
    }
    return c->mr_io;
}

static int mr_alloc(femo_ctx* c) {
    if (c->mr_v) return 0;
    const size_t n = (size_t)c->ndof;
    HIPCHK(c, hipMalloc((void**)&c->mr_v, 4 * n * sizeof(double)));
    HIPCHK(c, hipMalloc((void**)&c->mr_y, 4 * n * sizeof(double)));
    HIPCHK(c, hipMalloc((void**)&c->mr_work, 20 * n * sizeof(double)));
    HIPCHK(c, hipMalloc((void**)&c->mr_scal, 32 * sizeof(double)));
    HIPCHK(c, hipHostMalloc((void**)&c->mr_scal_host, 32 * sizeof(double)));
    return 0;
}

// z_r = (L L^T)^-1 r_r for g = 1 .. 4 vectors at once: 3 vectors ride as 4 (the fourth lane carries zeros)
static int frontal_apply_group(femo_ctx* c, int g, double* const* rin, double* const* zout) {
    const int64_t n = c->ndof;
    if (g == 1) {
        HIPCHK(c, hipMemcpyAsync(zout[0], rin[0], (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        return frontal_solve(c, zout[0]);
    }
    const int NRk = g == 2 ? 2 : 4;
    VecPtrs src, dst;
    for (int r = 0; r < 4; ++r) { src.p[r] = rin[std::min(r, g - 1)]; dst.p[r] = zout[std::min(r, g - 1)]; }
    const int vg = vec_grid(n);
    if (NRk == 2) {
        hipLaunchKernelGGL(k_interleave<2>, dim3(vg), dim3(256), 0, c->stream, src, c->mr_v, n);
        if (int rc = frontal_solve_multi<2>(c, c->mr_v, c->mr_y)) return rc;
        hipLaunchKernelGGL(k_deinterleave<2>, dim3(vg), dim3(256), 0, c->stream, (const double*)c->mr_v, dst, n);
    } else {
        // (g == 3: lane 3 repeats vector 2 and is written back to the same place with the same values)
        hipLaunchKernelGGL(k_interleave<4>, dim3(vg), dim3(256), 0, c->stream, src, c->mr_v, n);
        if (int rc = frontal_solve_multi<4>(c, c->mr_v, c->mr_y)) return rc;
        hipLaunchKernelGGL(k_deinterleave<4>, dim3(vg), dim3(256), 0, c->stream, (const double*)c->mr_v, dst, n);
    }
    HIPCHK(c, hipGetLastError());
    return 0;
}

// PCG with the multifrontal preconditioner for g <= 4 right-hand sides: independent recurrences (own alpha, beta and stopping test
// per right-hand side -- the iterates are those of g separate solves), the preconditioner applied to all residuals in ONE pair of
// sweeps, one host synchronisation per iteration.  B[r] is overwritten (masked), X[r] receives the solution (zero initial guess).
static int pcg_frontal_group(femo_ctx* c, int g, double* const* B, double* const* X, int32_t* iters, double* relres) {
    const int64_t n = c->ndof;
    const int vg = vec_grid(n);
    const unsigned char* mask = c->has_mask ? c->mask : nullptr;
    if (mr_alloc(c)) return 1;
    if (!c->fr.factored)
        if (int rc = frontal_factorize(c)) return rc;
    double *R[4], *Z[4], *P[4], *AP[4];
    for (int r = 0; r < g; ++r) {
        double* base = c->mr_work + (size_t)r * 5 * n;
        R[r] = base; Z[r] = base + n; P[r] = base + 2 * n; AP[r] = base + 3 * n;
    }
    double bb[4] = {0, 0, 0, 0}, rr[4] = {0, 0, 0, 0};
    HIPCHK(c, hipMemsetAsync(c->mr_scal, 0, 32 * sizeof(double), c->stream));
    for (int r = 0; r < g; ++r) {
        if (mask) hipLaunchKernelGGL(k_mask_zero, dim3(vg), dim3(256), 0, c->stream, B[r], mask, n);
        hipLaunchKernelGGL(k_dot, dim3(red_grid(n)), dim3(256), 0, c->stream, (const double*)B[r], (const double*)B[r], n, c->mr_scal + 8 * r + 3);
        hipLaunchKernelGGL(k_fill, dim3(vg), dim3(256), 0, c->stream, X[r], 0.0, n);
        HIPCHK(c, hipMemcpyAsync(R[r], B[r], (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    }
    HIPCHK(c, hipMemcpyAsync(c->mr_scal_host, c->mr_scal, 32 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    bool active[4] = {false, false, false, false};
    int its[4] = {0, 0, 0, 0};
    for (int r = 0; r < g; ++r) { bb[r] = rr[r] = c->mr_scal_host[8 * r + 3]; active[r] = bb[r] > 0; }
    int k = 0;
    auto any_active = [&]() { for (int r = 0; r < g; ++r) if (active[r]) return true; return false; };
    while (any_active() && k < c->maxit) {
        // one application of the factor for every residual (those of converged right-hand sides ride along: the bytes are the factor's)
        if (int rc = frontal_apply_group(c, g, R, Z)) return rc;
        for (int r = 0; r < g; ++r) {
            if (!active[r]) continue;
            double* sc = c->mr_scal + 8 * r;
            HIPCHK(c, hipMemsetAsync(sc + 1, 0, 3 * sizeof(double), c->stream));
            hipLaunchKernelGGL(k_dot, dim3(red_grid(n)), dim3(256), 0, c->stream, (const double*)R[r], (const double*)Z[r], n, sc + 1);
            hipLaunchKernelGGL(k_pcgf_direction, dim3(vg), dim3(256), 0, c->stream, P[r], (const double*)Z[r], (const double*)sc, k == 0 ? 1 : 0, n);
            hipLaunchKernelGGL(k_fill, dim3(vg), dim3(256), 0, c->stream, AP[r], 0.0, n);
            if (op_apply(c, P[r], AP[r], nullptr, nullptr, nullptr, true, c->op_aK, c->op_aM)) return 1;
            if (mask) hipLaunchKernelGGL(k_mask_zero, dim3(vg), dim3(256), 0, c->stream, AP[r], mask, n);
            hipLaunchKernelGGL(k_dot, dim3(red_grid(n)), dim3(256), 0, c->stream, (const double*)P[r], (const double*)AP[r], n, sc + 2);
            hipLaunchKernelGGL(k_pcgf_update, dim3(red_grid(n)), dim3(256), 0, c->stream, X[r], R[r], (const double*)P[r], (const double*)AP[r], sc, n);
        }
        HIPCHK(c, hipMemcpyAsync(c->mr_scal_host, c->mr_scal, 32 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        ++k;
        for (int r = 0; r < g; ++r) {
            if (!active[r]) continue;
            const double pAp = c->mr_scal_host[8 * r + 2];
            rr[r] = c->mr_scal_host[8 * r + 3];
            its[r] = k;
            if (!(pAp > 0)) return fail(c, "PCG broke down: p.Ap <= 0 (preconditioner or operator not positive definite)");
            if (!(rr[r] > c->rtol * c->rtol * bb[r])) active[r] = false;
        }
    }
    int rc = 0;
    for (int r = 0; r < g; ++r) {
        if (iters) iters[r] = its[r];
        const double rel = bb[r] > 0 ? sqrt(rr[r] / bb[r]) : 0.0;
        if (relres) relres[r] = rel;
        if (active[r] && c->opt.strict) {
            char buf[200];
            snprintf(buf, sizeof buf, "PCG (multifrontal preconditioner, %d right-hand sides) did not converge: relative residual %.3e after %d iterations "
                     "(rtol %.1e, maxit %d)", g, rel, k, c->rtol, c->maxit);
            c->err = buf;
            rc = 4;
        }
    }
    return rc;
}

// The preconditioner application of the PCG loop (always on c->z: ~56 dependent launches whose arguments depend on the plan and the
// options only) captured once as a HIP graph and replayed (option "sweep_graph").
static int frontal_solve_z(femo_ctx* c) {
    auto& fr = c->fr;
    if (!c->opt.sweep_graph) return frontal_solve(c, c->z);
    if (join_xinv(c)) return 1;                            // the cross-stream join stays outside the capture
    const long long graph_key = 2 * c->opt_version + (fr.w_mode ? 1 : 0);
    if (!fr.sweep_graph || fr.sweep_graph_key != graph_key) {
        if (fr.sweep_graph) { hipGraphExecDestroy(fr.sweep_graph); fr.sweep_graph = nullptr; }
        hipGraph_t g = nullptr;
        HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        const int rc = frontal_solve(c, c->z);
        const hipError_t e = hipStreamEndCapture(c->stream, &g);
        if (rc) { if (g) hipGraphDestroy(g); return rc; }
        HIPCHK(c, e);
        HIPCHK(c, hipGraphInstantiate(&fr.sweep_graph, g, nullptr, nullptr, 0));
        hipGraphDestroy(g);
        fr.sweep_graph_key = graph_key;
    }
    HIPCHK(c, hipGraphLaunch(fr.sweep_graph, c->stream));
    return 0;
}

// option "stale_factor": remember the fields the factor just made belongs to
static int snapshot_fields(femo_ctx* c) {
    if (c->opt.stale_factor <= 0) { c->fr.snap_valid = false; return 0; }
    double* cur[5] = {c->h, c->E, c->nu, c->rho, c->has_uhat ? c->uhat : nullptr};
    const int64_t len[5] = {c->nT, c->nT, c->nT, c->nT, 3 * (int64_t)c->nn};
    for (int i = 0; i < 5; ++i) {
        if (!cur[i]) { if (c->fr.snap[i]) { hipFree(c->fr.snap[i]); c->fr.snap[i] = nullptr; } continue; }
        if (!c->fr.snap[i]) HIPCHK(c, hipMalloc((void**)&c->fr.snap[i], (size_t)len[i] * sizeof(double)));
        HIPCHK(c, hipMemcpyAsync(c->fr.snap[i], cur[i], (size_t)len[i] * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    }
    c->fr.snap_valid = true;
    return 0;
}

// PCG preconditioned by the multifrontal factorisation (a handful of iterations)
static int pcg_frontal(femo_ctx* c, double* b, double* x, bool zero_guess, int32_t* iters, double* relres) {
    const int64_t n = c->ndof;
    const int vg = vec_grid(n);
    const unsigned char* mask = c->has_mask ? c->mask : nullptr;
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    // option "stale_factor": the factor of an earlier design preconditions this solve; refreshed below if the iteration drags
    bool stale = !c->fr.factored && c->opt.stale_factor > 0 && c->fr.have_factor && c->fr.snap_valid;
    if (stale) {
        // how far the design has moved from the one the factor belongs to: beyond stale_rel the iterations a kept factor needs cost
        // more than a factorisation (profiles/r5_stale_factor.txt), so it is refreshed at once
        double* cur[5] = {c->h, c->E, c->nu, c->rho, c->has_uhat ? c->uhat : nullptr};
        const int64_t len[5] = {c->nT, c->nT, c->nT, c->nT, 3 * (int64_t)c->nn};
        for (int i = 0; i < 5 && stale; ++i) {
            if (!cur[i] && !c->fr.snap[i]) continue;                                  // mesh motion absent then and now
            if (!cur[i] || !c->fr.snap[i]) { stale = false; break; }                  // switched on or off since: a different operator
            HIPCHK(c, hipMemsetAsync(c->scal + 5, 0, 2 * sizeof(double), c->stream));
            hipLaunchKernelGGL(k_sq_change, dim3(red_grid(len[i])), dim3(256), 0, c->stream, (const double*)cur[i], (const double*)c->fr.snap[i], len[i], c->scal + 5);
            HIPCHK(c, hipMemcpyAsync(c->scal_host + 5, c->scal + 5, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            // a field is measured against itself; the mesh motion against the cell size (what changes the operator is grad uhat ~
            // |delta uhat| / h_K, and a relative change of a displacement field says nothing): sum |delta|^2 <= rel^2 nn mean(h_K)^2
            const double d2 = c->scal_host[5], r2 = i == 4 ? (double)c->nn * c->hK_mean * c->hK_mean : c->scal_host[6];
            if (!(d2 <= c->opt.stale_rel * c->opt.stale_rel * r2)) stale = false;
        }
    }
    int factor_state = stale ? 1 : 0;
    if (stale) { c->fr.t_assemble_ms = 0; c->fr.t_factor_ms = 0; }
    if (!c->fr.factored && !stale) {
        // option "sweep_ahead": a cold solve from a zero guess applies the factor to z = b first, and b is known now -- hand it to the
        // factorisation, which starts its forward sweep through the lower levels beside the chain of the top ones
        const bool ahead = zero_guess && c->opt.sweep_ahead > 0 && c->fr.nlevels > c->opt.sweep_ahead + 1 && !c->opt.equilibrate &&
                           c->opt.sweep_w == 0 && !c->opt.sweep_fuse;
        c->fr.ahead_levels = 0;
        if (ahead) {
            if (mask) hipLaunchKernelGGL(k_mask_zero, dim3(vg), dim3(256), 0, c->stream, b, mask, n);
            HIPCHK(c, hipMemcpyAsync(c->z, b, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
            c->fr.ahead_vec = c->z;
        }
        const int rc = frontal_factorize(c);
        c->fr.ahead_vec = nullptr;
        if (rc) { c->fr.ahead_levels = 0; return rc; }
    }
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    auto dot = [&](const double* a, const double* bb, double* out) -> int {
        HIPCHK(c, hipMemsetAsync(c->scal + 7, 0, sizeof(double), c->stream));
        hipLaunchKernelGGL(k_dot, dim3(red_grid(n)), dim3(256), 0, c->stream, a, bb, n, c->scal + 7);
        HIPCHK(c, hipMemcpyAsync(c->scal_host + 7, c->scal + 7, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        *out = c->scal_host[7];
        return 0;
    };
    if (mask) hipLaunchKernelGGL(k_mask_zero, dim3(vg), dim3(256), 0, c->stream, b, mask, n);
    double bb = 0, rr = 0, pAp = 0;
    if (dot(b, b, &bb)) return 1;
    // r = b - A x
    if (zero_guess) {
        hipLaunchKernelGGL(k_fill, dim3(vg), dim3(256), 0, c->stream, x, 0.0, n);
        HIPCHK(c, hipMemcpyAsync(c->r, b, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    } else {
        if (mask) hipLaunchKernelGGL(k_mask_zero, dim3(vg), dim3(256), 0, c->stream, x, mask, n);
        hipLaunchKernelGGL(k_fill, dim3(vg), dim3(256), 0, c->stream, c->Ap, 0.0, n);
        if (op_apply(c, x, c->Ap, nullptr, nullptr, nullptr, true, c->op_aK, c->op_aM)) return 1;
        HIPCHK(c, hipMemcpyAsync(c->r, b, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        hipLaunchKernelGGL(k_axpby, dim3(vg), dim3(256), 0, c->stream, c->r, -1.0, c->Ap, 1.0, n);
        if (mask) hipLaunchKernelGGL(k_mask_zero, dim3(vg), dim3(256), 0, c->stream, c->r, mask, n);
    }
    if (zero_guess) rr = bb;                       // r = b
    else if (dot(c->r, c->r, &rr)) return 1;
    int k = 0, napply = 0;
    const double target = c->rtol * c->rtol * bb;
    // device scalars: [0] r.z of the previous iteration, [1] r.z, [2] p.Ap, [3] r.r -- one host synchronisation per iteration
    int k_restart = 0;                             // iteration at which the search directions start afresh
    if (c->fr.ahead_levels > 0 && !(bb > 0 && rr > target && k < c->maxit)) {      // nothing to iterate: the started sweep must not outlive this call
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_a[2], 0));
        c->fr.ahead_levels = 0;
    }
    while (bb > 0 && rr > target && k < c->maxit) {
        // (never end a solve unconverged on a kept factor: a fresh one needs two iterations, so the refresh comes no later than maxit - 2)
        if (stale && k >= std::min(c->opt.stale_factor, std::max(c->maxit - 2, 0))) {
            // the kept factor is too far from this operator: factorise the current one and restart from the iterate reached so far
            if (int rc = frontal_factorize(c)) return rc;
            stale = false; factor_state = 2; k_restart = k;
            hipLaunchKernelGGL(k_fill, dim3(vg), dim3(256), 0, c->stream, c->Ap, 0.0, n);
            if (op_apply(c, x, c->Ap, nullptr, nullptr, nullptr, true, c->op_aK, c->op_aM)) return 1;
            ++napply;
            HIPCHK(c, hipMemcpyAsync(c->r, b, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
            hipLaunchKernelGGL(k_axpby, dim3(vg), dim3(256), 0, c->stream, c->r, -1.0, c->Ap, 1.0, n);
            if (mask) hipLaunchKernelGGL(k_mask_zero, dim3(vg), dim3(256), 0, c->stream, c->r, mask, n);
            if (dot(c->r, c->r, &rr)) return 1;
            if (!(rr > target)) break;
        }
        HIPCHK(c, hipMemsetAsync(c->scal + 1, 0, 3 * sizeof(double), c->stream));
        if (k == 0 && c->fr.ahead_levels > 0) {
            // z = b has been swept through the levels [0, ahead_levels) beside the factorisation of the top: join, finish the sweeps
            const int la = c->fr.ahead_levels;
            c->fr.ahead_levels = 0;
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_a[2], 0));
            if (frontal_fwd(c, c->z, la, c->fr.nlevels)) return 1;
            if (frontal_bwd(c, c->z, 0, c->fr.nlevels)) return 1;
        } else {
        c->fr.ahead_levels = 0;
        HIPCHK(c, hipMemcpyAsync(c->z, c->r, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        if (frontal_solve_z(c)) return 1;
        }
        hipLaunchKernelGGL(k_dot, dim3(red_grid(n)), dim3(256), 0, c->stream, c->r, c->z, n, c->scal + 1);
        hipLaunchKernelGGL(k_pcgf_direction, dim3(vg), dim3(256), 0, c->stream, c->p, c->z, c->scal, k == k_restart ? 1 : 0, n);
        hipLaunchKernelGGL(k_fill, dim3(vg), dim3(256), 0, c->stream, c->Ap, 0.0, n);
        if (op_apply(c, c->p, c->Ap, nullptr, nullptr, nullptr, true, c->op_aK, c->op_aM)) return 1;
        ++napply;
        if (mask) hipLaunchKernelGGL(k_mask_zero, dim3(vg), dim3(256), 0, c->stream, c->Ap, mask, n);
        hipLaunchKernelGGL(k_dot, dim3(red_grid(n)), dim3(256), 0, c->stream, c->p, c->Ap, n, c->scal + 2);
        hipLaunchKernelGGL(k_pcgf_update, dim3(red_grid(n)), dim3(256), 0, c->stream, x, c->r, c->p, c->Ap, c->scal, n);
        HIPCHK(c, hipMemcpyAsync(c->scal_host, c->scal, 4 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        pAp = c->scal_host[2]; rr = c->scal_host[3];
        if (!(pAp > 0)) return fail(c, "PCG broke down: p.Ap <= 0 (preconditioner or operator not positive definite)");
        if (!(rr == rr)) return fail(c, "PCG broke down (NaN residual)");
        ++k;
    }
    HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float t_loop = 0;
    hipEventElapsedTime(&t_loop, c->ev[1], c->ev[2]);
    c->timing[0] = c->fr.t_assemble_ms + c->fr.t_factor_ms; c->timing[1] = t_loop; c->timing[2] = c->timing[0] + t_loop;
    if (factor_state == 2) c->timing[1] = std::max(0.0, (double)t_loop - c->timing[0]);      // the refresh happened inside the loop's event pair
    c->timing[2] = c->timing[0] + c->timing[1];
    c->timing[3] = factor_state; c->timing[4] = napply;
    return finish_solve(c, "PCG (multifrontal preconditioner)", k, rr, bb, target, iters, relres);
}

// Right-preconditioned BiCGStab on the same operator and preconditioners (Jacobi or the multifrontal factor).  The
// operator of this path is symmetric positive definite, so conjugate gradients are the default; BiCGStab is the
// second Krylov method the north star names and costs two operator / preconditioner applications per iteration.
static int bicgstab(femo_ctx* c, double* b, double* x, bool zero_guess, int32_t* iters, double* relres) {
    const int64_t n = c->ndof;
    const int vg = vec_grid(n);
    const unsigned char* mask = c->has_mask ? c->mask : nullptr;
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    if (c->precond == 2) {
        if (!c->fr.factored)
            if (int rc = frontal_factorize(c)) return rc;
    } else {
        if (c->op_aM != 0.0 || c->op_aK != 1.0) return fail(c, "the Jacobi preconditioner only handles the static operator; use preconditioner 2");
        if (refresh_diag(c)) return 1;
    }
    for (int i = 0; i < 5; ++i)
        if (!c->bi[i]) HIPCHK(c, hipMalloc((void**)&c->bi[i], (size_t)n * sizeof(double)));
    double *rh = c->bi[0], *v = c->bi[1], *s = c->bi[2], *t = c->bi[3], *y = c->bi[4], *r = c->r, *p = c->p, *z = c->z;
    auto dot = [&](const double* a, const double* bb, double* out) -> int {
        HIPCHK(c, hipMemsetAsync(c->scal + 7, 0, sizeof(double), c->stream));
        hipLaunchKernelGGL(k_dot, dim3(red_grid(n)), dim3(256), 0, c->stream, a, bb, n, c->scal + 7);
        HIPCHK(c, hipMemcpyAsync(c->scal_host + 7, c->scal + 7, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        *out = c->scal_host[7];
        return 0;
    };
    auto copy = [&](double* dst, const double* src) { return hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream); };
    auto apply = [&](const double* in, double* out) -> int {               // out = A in (masked rows zero)
        hipLaunchKernelGGL(k_fill, dim3(vg), dim3(256), 0, c->stream, out, 0.0, n);
        if (op_apply(c, in, out, nullptr, nullptr, nullptr, true, c->op_aK, c->op_aM)) return 1;
        if (mask) hipLaunchKernelGGL(k_mask_zero, dim3(vg), dim3(256), 0, c->stream, out, mask, n);
        return 0;
    };
    auto precond = [&](const double* in, double* out) -> int {             // out = M^-1 in
        HIPCHK(c, copy(out, in));
        if (c->precond == 2) return frontal_solve(c, out);
        hipLaunchKernelGGL(k_mul, dim3(vg), dim3(256), 0, c->stream, out, c->dinv, n);
        return 0;
    };
    if (mask) hipLaunchKernelGGL(k_mask_zero, dim3(vg), dim3(256), 0, c->stream, b, mask, n);
    double bb = 0, rr = 0;
    if (dot(b, b, &bb)) return 1;
    if (zero_guess) {
        hipLaunchKernelGGL(k_fill, dim3(vg), dim3(256), 0, c->stream, x, 0.0, n);
        HIPCHK(c, copy(r, b));
    } else {
        if (mask) hipLaunchKernelGGL(k_mask_zero, dim3(vg), dim3(256), 0, c->stream, x, mask, n);
        if (apply(x, t)) return 1;
        HIPCHK(c, copy(r, b));
        hipLaunchKernelGGL(k_axpby, dim3(vg), dim3(256), 0, c->stream, r, -1.0, t, 1.0, n);
    }
    HIPCHK(c, copy(rh, r));
    if (dot(r, r, &rr)) return 1;
    double rho = 1.0, alpha = 1.0, omega = 1.0;
    hipLaunchKernelGGL(k_fill, dim3(vg), dim3(256), 0, c->stream, v, 0.0, n);
    hipLaunchKernelGGL(k_fill, dim3(vg), dim3(256), 0, c->stream, p, 0.0, n);
    int k = 0, napply = 0;
    const double target = c->rtol * c->rtol * bb;
    while (bb > 0 && rr > target && k < c->maxit) {
        double rho_new = 0, rhv = 0, ts = 0, tt = 0;
        if (dot(rh, r, &rho_new)) return 1;
        if (rho_new == 0.0 || omega == 0.0) return fail(c, "BiCGStab broke down (rho or omega vanished)");
        const double beta = (rho_new / rho) * (alpha / omega);
        hipLaunchKernelGGL(k_axpby, dim3(vg), dim3(256), 0, c->stream, p, -omega, v, 1.0, n);       // p -= omega v
        hipLaunchKernelGGL(k_axpby, dim3(vg), dim3(256), 0, c->stream, p, 1.0, r, beta, n);          // p = r + beta p
        if (precond(p, y) || apply(y, v)) return 1;
        ++napply;
        if (dot(rh, v, &rhv)) return 1;
        if (rhv == 0.0) return fail(c, "BiCGStab broke down (r_hat . v = 0)");
        alpha = rho_new / rhv;
        HIPCHK(c, copy(s, r));
        hipLaunchKernelGGL(k_axpby, dim3(vg), dim3(256), 0, c->stream, s, -alpha, v, 1.0, n);        // s = r - alpha v
        hipLaunchKernelGGL(k_axpby, dim3(vg), dim3(256), 0, c->stream, x, alpha, y, 1.0, n);         // x += alpha y
        if (dot(s, s, &rr)) return 1;
        ++k;
        rho = rho_new;
        if (rr <= target) break;
        if (precond(s, z) || apply(z, t)) return 1;
        ++napply;
        if (dot(t, s, &ts) || dot(t, t, &tt)) return 1;
        if (tt == 0.0) return fail(c, "BiCGStab broke down (t = 0)");
        omega = ts / tt;
        hipLaunchKernelGGL(k_axpby, dim3(vg), dim3(256), 0, c->stream, x, omega, z, 1.0, n);         // x += omega z
        HIPCHK(c, copy(r, s));
        hipLaunchKernelGGL(k_axpby, dim3(vg), dim3(256), 0, c->stream, r, -omega, t, 1.0, n);        // r = s - omega t
        if (dot(r, r, &rr)) return 1;
        if (!(rr == rr)) return fail(c, "BiCGStab broke down (NaN residual)");
    }
    HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float t_loop = 0;
    hipEventElapsedTime(&t_loop, c->ev[1], c->ev[2]);
    c->timing[0] = c->precond == 2 ? c->fr.t_assemble_ms + c->fr.t_factor_ms : 0.0; c->timing[1] = t_loop;
    c->timing[2] = c->timing[0] + t_loop; c->timing[3] = 0; c->timing[4] = napply;
    return finish_solve(c, "BiCGStab", k, rr, bb, target, iters, relres);
}

static int solve_dispatch(femo_ctx* c, double* b, double* x, bool zero_guess, int32_t* iters, double* relres) {
    if (c->krylov == 1) return bicgstab(c, b, x, zero_guess, iters, relres);
    return c->precond == 2 ? pcg_frontal(c, b, x, zero_guess, iters, relres) : pcg(c, b, x, zero_guess, iters, relres);
}

// ------------------------------------------------------------------------------------------ C ABI
extern "C" {

int femo_version(void) { return FEMO_VERSION; }

int femo_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* femo_last_error(const femo_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

static int alloc_d(femo_ctx* c, double** p, int64_t n) {
    HIPCHK(c, hipMalloc((void**)p, std::max<int64_t>(n, 1) * sizeof(double)));
    HIPCHK(c, hipMemsetAsync(*p, 0, std::max<int64_t>(n, 1) * sizeof(double), c->stream));
    return 0;
}

static int create_impl(femo_ctx* c, const double* xyz, const int32_t* cells, const int32_t* cell_p2, int nquad) {
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamCreate(&c->stream));
    HIPCHK(c, hipStreamCreate(&c->stream2));
    for (int i = 0; i < 2; ++i) HIPCHK(c, hipEventCreateWithFlags(&c->ev_la[i], hipEventDisableTiming));
    for (int i = 0; i < 2; ++i) HIPCHK(c, hipEventCreateWithFlags(&c->ev_sp[i], hipEventDisableTiming));
    HIPCHK(c, hipStreamCreate(&c->stream3));
    for (int i = 0; i < 2; ++i) HIPCHK(c, hipEventCreateWithFlags(&c->ev_x[i], hipEventDisableTiming));
    HIPCHK(c, hipStreamCreate(&c->stream_g));
    for (int i = 0; i < 2; ++i) HIPCHK(c, hipEventCreateWithFlags(&c->ev_g[i], hipEventDisableTiming));
    HIPCHK(c, hipStreamCreate(&c->stream_a));      // (a lowest-priority stream here made the factorisation 6 ms SLOWER: 13.4 -> 19.4 ms)
    for (int i = 0; i < 3; ++i) HIPCHK(c, hipEventCreateWithFlags(&c->ev_a[i], hipEventDisableTiming));
    {
        // The diagonal look-ahead (factorize_fronts) runs the bulk of a panel's work on stream_m beside the next diagonal
        // block, whose one workgroup per front needs a whole CU's LDS: stream_m leaves 32 of the 256 CUs alone (bits whose
        // index / 8 is a multiple of 8 -- four CUs per XCD if the mask interleaves the XCDs, eight in every other XCD if
        // it runs through them one after the other), so those workgroups start at once.  Without the mask the schedule
        // is still correct; the option falls back to the plain super-panel order when the stream cannot be made.
        uint32_t cumask[8];
        for (int w = 0; w < 8; ++w) cumask[w] = (w & 1) ? 0xffffffffu : 0xffffff00u;
        if (hipExtStreamCreateWithCUMask(&c->stream_m, 8, cumask) != hipSuccess) { (void)hipGetLastError(); c->stream_m = nullptr; }
        for (int i = 0; i < 2; ++i) HIPCHK(c, hipEventCreateWithFlags(&c->ev_da[i], hipEventDisableTiming));
    }
    for (int i = 0; i < 4; ++i) HIPCHK(c, hipEventCreate(&c->ev[i]));
    const int nel = c->nel, nvc = c->nvc, npc = c->npc;
    // SoA connectivity
    std::vector<int> soa_c((size_t)nvc * nel), soa_p((size_t)npc * nel);
    std::vector<double> hK(nel);
    for (int e = 0; e < nel; ++e) {
        for (int b = 0; b < nvc; ++b) soa_c[(size_t)b * nel + e] = cells[(size_t)e * nvc + b];
        for (int a = 0; a < npc; ++a) soa_p[(size_t)a * nel + e] = cell_p2[(size_t)e * npc + a];
        double d = 0.0;
        for (int i = 0; i < nvc; ++i)
            for (int j = i + 1; j < nvc; ++j) {
                const double* xi = xyz + 3 * (size_t)cells[(size_t)e * nvc + i];
                const double* xj = xyz + 3 * (size_t)cells[(size_t)e * nvc + j];
                const double dd = sqrt((xi[0] - xj[0]) * (xi[0] - xj[0]) + (xi[1] - xj[1]) * (xi[1] - xj[1]) +
                                       (xi[2] - xj[2]) * (xi[2] - xj[2]));
                d = std::max(d, dd);
            }
        hK[e] = d;
    }
    HIPCHK(c, hipMalloc((void**)&c->xyz, (size_t)c->nn * 3 * sizeof(double)));
    HIPCHK(c, hipMemcpy(c->xyz, xyz, (size_t)c->nn * 3 * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(c, hipMalloc((void**)&c->cells, soa_c.size() * sizeof(int)));
    HIPCHK(c, hipMemcpy(c->cells, soa_c.data(), soa_c.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(c, hipMalloc((void**)&c->cellp2, soa_p.size() * sizeof(int)));
    HIPCHK(c, hipMemcpy(c->cellp2, soa_p.data(), soa_p.size() * sizeof(int), hipMemcpyHostToDevice));
    {   // Morton order of the element centroids (30 bits per axis on the bounding box)
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        std::vector<double> cen((size_t)nel * 3, 0.0);
        for (int e = 0; e < nel; ++e)
            for (int k = 0; k < 3; ++k) {
                double s = 0.0;
                for (int b = 0; b < nvc; ++b) s += xyz[3 * (size_t)cells[(size_t)e * nvc + b] + k];
                s /= nvc;
                cen[3 * (size_t)e + k] = s;
                lo[k] = std::min(lo[k], s); hi[k] = std::max(hi[k], s);
            }
        std::vector<std::pair<unsigned long long, int>> key(nel);
        auto spread = [](unsigned long long v) {
            unsigned long long r = 0;
            for (int b = 0; b < 21; ++b) r |= ((v >> b) & 1ull) << (3 * b);
            return r;
        };
        for (int e = 0; e < nel; ++e) {
            unsigned long long code = 0;
            for (int k = 0; k < 3; ++k) {
                const double ext = hi[k] - lo[k];
                const unsigned long long q = ext > 0 ? (unsigned long long)((cen[3 * (size_t)e + k] - lo[k]) / ext * 2097151.0) : 0ull;
                code |= spread(q) << k;
            }
            key[e] = {code, e};
        }
        std::sort(key.begin(), key.end());
        std::vector<int> eo(nel);
        for (int e = 0; e < nel; ++e) eo[e] = key[e].second;
        HIPCHK(c, hipMalloc((void**)&c->eorder, (size_t)nel * sizeof(int)));
        HIPCHK(c, hipMemcpy(c->eorder, eo.data(), (size_t)nel * sizeof(int), hipMemcpyHostToDevice));
        // inverted connectivity in slot order
        std::vector<int> off(c->nP2 + 1, 0), ent((size_t)nel * npc);
        for (size_t i = 0; i < (size_t)nel * npc; ++i) off[cell_p2[i] + 1]++;
        for (int p = 0; p < c->nP2; ++p) off[p + 1] += off[p];
        std::vector<int> cur(off.begin(), off.end() - 1);
        for (int slot = 0; slot < nel; ++slot) {
            const int e = eo[slot];
            for (int a = 0; a < npc; ++a) ent[cur[cell_p2[(size_t)e * npc + a]]++] = slot * npc + a;
        }
        HIPCHK(c, hipMalloc((void**)&c->n2e_off, off.size() * sizeof(int)));
        HIPCHK(c, hipMemcpy(c->n2e_off, off.data(), off.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(c, hipMalloc((void**)&c->n2e_ent, ent.size() * sizeof(int)));
        HIPCHK(c, hipMemcpy(c->n2e_ent, ent.data(), ent.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(c, hipMalloc((void**)&c->ybuf, (size_t)nel * YSTRIDE * sizeof(double)));
    }
    HIPCHK(c, hipMalloc((void**)&c->hK, (size_t)nel * sizeof(double)));
    HIPCHK(c, hipMemcpy(c->hK, hK.data(), (size_t)nel * sizeof(double), hipMemcpyHostToDevice));
    { double sum = 0; for (double v : hK) sum += v; c->hK_mean = sum / nel; }
    Tables T;
    c->nquad = nquad;
    build_tables(c->quad, nquad, T, 0, c->cg1, c->cr);
    HIPCHK(c, hipMalloc((void**)&c->tab, sizeof(Tables)));
    HIPCHK(c, hipMemcpy(c->tab, &T, sizeof(Tables), hipMemcpyHostToDevice));
    c->tab_nq = T.nq;
    Tables TS_;
    // quadrature_degree 4 (rm_shell_model.py:200-205): 3 x 3 Gauss on quadrilaterals, the 6-point rule of degree 4 on triangles
    build_tables(c->quad, c->quad ? 3 : 4, TS_, 0, c->cg1, c->cr);
    HIPCHK(c, hipMalloc((void**)&c->tab_s, sizeof(Tables)));
    HIPCHK(c, hipMemcpy(c->tab_s, &TS_, sizeof(Tables), hipMemcpyHostToDevice));
    c->nT = c->ewm ? nel : c->nn;
    c->nF = c->ewp ? nel : c->nn;
    if (alloc_d(c, &c->h, c->nT) || alloc_d(c, &c->E, c->nT) || alloc_d(c, &c->nu, c->nT) || alloc_d(c, &c->rho, c->nT) ||
        alloc_d(c, &c->f, 3 * c->nF) || alloc_d(c, &c->uhat, 3 * (int64_t)c->nn))
        return 1;
    double** vecs[] = {&c->w, &c->lam, &c->r, &c->z, &c->p, &c->Ap, &c->dinv, &c->b, &c->tmp};
    for (auto v : vecs)
        if (alloc_d(c, v, c->ndof)) return 1;
    HIPCHK(c, hipMalloc((void**)&c->scal, 8 * sizeof(double)));
    HIPCHK(c, hipHostMalloc((void**)&c->scal_host, 8 * sizeof(double)));
    HIPCHK(c, hipMalloc((void**)&c->mask, (size_t)c->ndof));
    HIPCHK(c, hipMemset(c->mask, 0, (size_t)c->ndof));
    // FEA.add_input initial values (rm_shell_model.py:209-214): thickness 1e-3, others 1, uhat 0
    const int vgT = vec_grid(c->nT);
    hipLaunchKernelGGL(k_fill, dim3(vgT), dim3(256), 0, c->stream, c->h, 1e-3, c->nT);
    hipLaunchKernelGGL(k_fill, dim3(vgT), dim3(256), 0, c->stream, c->E, 1.0, c->nT);
    hipLaunchKernelGGL(k_fill, dim3(vgT), dim3(256), 0, c->stream, c->nu, 1.0, c->nT);
    hipLaunchKernelGGL(k_fill, dim3(vgT), dim3(256), 0, c->stream, c->rho, 1.0, c->nT);
    hipLaunchKernelGGL(k_fill, dim3(vec_grid(3 * c->nF)), dim3(256), 0, c->stream, c->f, 1.0, 3 * c->nF);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int femo_create(femo_ctx** out, int device, int32_t nn, int32_t nel, int32_t nvc, int32_t nP2, const double* xyz,
                const int32_t* cells, const int32_t* cell_p2, int elementwise_material, int elementwise_pressure, int nquad) {
    return femo_create_ghost(out, device, nn, nel, nvc, nP2, xyz, cells, cell_p2, elementwise_material, elementwise_pressure,
                             nquad, 0);
}

int femo_create_ghost(femo_ctx** out, int device, int32_t nn, int32_t nel, int32_t nvc, int32_t nP2, const double* xyz,
                      const int32_t* cells, const int32_t* cell_p2, int elementwise_material, int elementwise_pressure,
                      int nquad, int32_t nghost) {
    return femo_create_element(out, device, nn, nel, nvc, nP2, xyz, cells, cell_p2, elementwise_material, elementwise_pressure, nquad,
                               nghost, 0);
}

int femo_create_element(femo_ctx** out, int device, int32_t nn, int32_t nel, int32_t nvc, int32_t nP2, const double* xyz,
                        const int32_t* cells, const int32_t* cell_p2, int elementwise_material, int elementwise_pressure,
                        int nquad, int32_t nghost, int element) {
    if (!out) return 2;
    if (nghost < 0) { g_create_error = "nghost must be >= 0"; return 2; }
    *out = nullptr;
    if (nvc != 3 && nvc != 4) { g_create_error = "nvc must be 3 (triangles) or 4 (quads)"; return 2; }
    if (nn <= 0 || nel <= 0 || !xyz || !cells || !cell_p2) { g_create_error = "empty mesh or null pointer"; return 2; }
    if (nvc == 4 && (nquad < 2 || nquad > 6)) { g_create_error = "nquad must be in 2..6"; return 2; }
    // triangles: nquad is the degree of the symmetric rule (triangle_rule); 0 = the default, degree 6
    if (nvc == 3 && nquad == 0) nquad = 6;
    if (nvc == 3 && nquad != 4 && nquad != 6 && nquad != 9 && nquad != 12) {
        g_create_error = "triangles: nquad is the degree of the symmetric rule -- 4 (6 points), 6 (12, the default 0), 9 (19) or 12 (33)";
        return 2;
    }
    // CG1CG1 (linear_shell_model.py:74-79): the caller's "P2 node" set is the vertex set itself and cell_p2 holds nvc entries per
    // cell -- told apart by nP2 == nn (a CG2CG1 mesh always has nP2 = nn + edges [+ cells] > nn)
    const bool cg1 = nP2 == nn;
    const int npc = cg1 ? nvc : (nvc == 4 ? 9 : 6);
    // element 1 = CG2CR1 (linear_shell_model.py:68-73): displacement on the P2 nodes, rotation on the EDGE MIDPOINTS (Crouzeix-Raviart);
    // triangles only, as in the reference.  The rotation nodes are the P2 nodes nn .. nP2 - 1 (cell_p2 lists a cell's edge midpoints behind
    // its vertices), so the state vector is [u(P2 nodes) | theta(edges)] with 3 nP2 + 3 (nP2 - nn) entries
    if (element != 0 && element != 1) { g_create_error = "element: 0 (CG2CG1 / CG1CG1 by the node count) or 1 (CG2CR1)"; return 2; }
    const bool cr = element == 1;
    if (cr && (nvc != 3 || cg1)) { g_create_error = "Invalid element type: CG2CR1 is defined on triangles with P2 displacement"; return 2; }
    for (int64_t i = 0; i < (int64_t)nel * nvc; ++i)
        if (cells[i] < 0 || cells[i] >= nn) { g_create_error = "cells refers to a vertex outside 0..nn-1"; return 2; }
    for (int64_t i = 0; i < (int64_t)nel * npc; ++i)
        if (cell_p2[i] < 0 || cell_p2[i] >= nP2) { g_create_error = "cell_p2 refers to a node outside 0..nP2-1"; return 2; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_create_error = "no HIP device available (libfemo_hip needs an MI355X; there is no CPU fallback)"; return 3; }
    if (device < 0 || device >= ndev) { g_create_error = "device index out of range"; return 2; }
    femo_ctx* c = new femo_ctx();
    c->device = device;
    c->nn = nn; c->nel = nel; c->nvc = nvc; c->npc = npc; c->nP2 = nP2;
    c->quad = nvc == 4;
    c->cg1 = cg1; c->cr = cr;
    c->nrot = cr ? nP2 - nn : nn;
    c->ndof_u = 3 * nP2; c->ndof = 3 * nP2 + 3 * c->nrot + nghost; c->nghost = nghost; c->ld = 3 * npc + 3 * nvc;
    c->ewm = elementwise_material != 0; c->ewp = elementwise_pressure != 0;
    if (create_impl(c, xyz, cells, cell_p2, nquad)) {
        g_create_error = c->err;
        femo_destroy(c);
        return 1;
    }
    *out = c;
    return 0;
}

void femo_destroy(femo_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipDeviceSynchronize();                   // stream2 / stream3 may still hold work that reads the buffers freed below
    if (c->gdir) hipFree(c->gdir);
    void* nptrs[] = {c->nm.W, c->nm.Fh, c->nm.wdot, c->nm.Fsw, c->nm.mu0, c->nm.mu1, c->nm.Lam, c->nm.Gh};
    for (void* p : nptrs)
        if (p) hipFree(p);
    void* mptrs[] = {c->mr_v, c->mr_y, c->mr_work, c->mr_scal, c->mr_io};
    for (void* p : mptrs)
        if (p) hipFree(p);
    if (c->mr_scal_host) hipHostFree(c->mr_scal_host);
    void* dptrs[] = {c->di.top_idx, c->di.sel, c->di.wdot, c->di.topbuf, c->di.topsave, c->di.gloc};
    for (void* p : dptrs)
        if (p) hipFree(p);
    for (double* p : c->fp)
        if (p) hipFree(p);
    if (c->eq) hipFree(c->eq);
    void* ptrs[] = {c->bi[0], c->bi[1], c->bi[2], c->bi[3], c->bi[4], c->ctag, c->gradbuf, c->csr_perm, c->csr_dest, c->csr_rowptr, c->csr_colidx, c->csr_vals, c->csr_ke, c->xyz, c->cells, c->cellp2, c->eorder, c->n2e_off, c->n2e_ent, c->ybuf, c->hK, c->tab, c->tab_s, c->tab_pre, c->h, c->E, c->nu, c->rho, c->f, c->uhat, c->fcell, c->fledge,
                    c->funode, c->fvnode, c->fM2, c->fM1, c->frnode, c->fMR, c->mask, c->w, c->lam, c->r, c->z, c->p, c->Ap, c->dinv, c->b, c->tmp,
                    c->scal};
    for (void* p : ptrs)
        if (p) hipFree(p);
    void* fptrs[] = {c->fr.nf, c->fr.npiv, c->fr.dofs, c->fr.upmap, c->fr.parent, c->fr.left, c->fr.right, c->fr.level_nodes,
                     c->fr.elem_front, c->fr.elem_map, c->fr.info, c->fr.poff, c->fr.soff, c->fr.doff, c->fr.linvoff, c->fr.P, c->fr.S, c->fr.Linv,
                     c->fr.xoff, c->fr.X, c->fr.Xtmp, c->fr.Swork, c->fr.cinv0, c->fr.cinv1, c->fr.slot_of, c->fr.fel_off, c->fr.fel, c->fr.sweep_cnt, c->fr.ftasks,
                     c->fr.btasks, c->fr.snap[0], c->fr.snap[1], c->fr.snap[2], c->fr.snap[3], c->fr.snap[4]};
    for (void* p : fptrs)
        if (p) hipFree(p);
    if (c->fr.sweep_graph) hipGraphExecDestroy(c->fr.sweep_graph);
    if (c->scal_host) hipHostFree(c->scal_host);
    for (int i = 0; i < 4; ++i)
        if (c->ev[i]) hipEventDestroy(c->ev[i]);
    for (int i = 0; i < 2; ++i) {
        if (c->ev_la[i]) hipEventDestroy(c->ev_la[i]);
        if (c->ev_sp[i]) hipEventDestroy(c->ev_sp[i]);
    }
    for (int i = 0; i < 2; ++i)
        if (c->ev_x[i]) hipEventDestroy(c->ev_x[i]);
    for (int i = 0; i < 2; ++i)
        if (c->ev_g[i]) hipEventDestroy(c->ev_g[i]);
    if (c->stream_g) hipStreamDestroy(c->stream_g);
    for (int i = 0; i < 3; ++i)
        if (c->ev_a[i]) hipEventDestroy(c->ev_a[i]);
    if (c->stream_a) hipStreamDestroy(c->stream_a);
    for (int i = 0; i < 2; ++i)
        if (c->ev_da[i]) hipEventDestroy(c->ev_da[i]);
    if (c->stream_m) hipStreamDestroy(c->stream_m);
    if (c->stream3) hipStreamDestroy(c->stream3);
    if (c->stream2) hipStreamDestroy(c->stream2);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

int64_t femo_ndof(const femo_ctx* c) { return c ? c->ndof : -1; }

static double* field_ptr(const femo_ctx* c, const char* name, int64_t* n) {
    const std::string s(name ? name : "");
    if (s == "thickness") { *n = c->nT; return c->h; }
    if (s == "E") { *n = c->nT; return c->E; }
    if (s == "nu") { *n = c->nT; return c->nu; }
    if (s == "density") { *n = c->nT; return c->rho; }
    if (s == "F_solid") { *n = 3 * c->nF; return c->f; }
    if (s == "uhat") { *n = 3 * (int64_t)c->nn; return c->uhat; }
    if (s == "dirichlet" && c->gdir) { *n = c->ndof; return c->gdir; }
    *n = -1;
    return nullptr;
}

int64_t femo_field_size(const femo_ctx* c, const char* name) {
    int64_t n;
    field_ptr(c, name, &n);
    return n;
}

int femo_set_penalty_facets(femo_ctx* c, int32_t nf, const int32_t* cl, double beta) {
    HIPCHK(c, hipSetDevice(c->device));
    if (nf < 0 || (nf > 0 && !cl)) return fail(c, "bad facet list");
    void* old[] = {c->fcell, c->fledge, c->funode, c->fvnode, c->fM2, c->fM1, c->frnode, c->fMR};
    for (void* p : old)
        if (p) hipFree(p);
    c->fcell = c->fledge = c->funode = c->fvnode = c->frnode = nullptr;
    c->fM2 = c->fM1 = c->fMR = nullptr;
    c->nf = 0;
    c->beta = beta;
    c->penalty_dirty = true;
    operator_changed(c);
    if (nf == 0) return 0;
    std::vector<int> hc((size_t)c->nvc * c->nel), hp((size_t)c->npc * c->nel);
    HIPCHK(c, hipMemcpy(hc.data(), c->cells, hc.size() * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(hp.data(), c->cellp2, hp.size() * sizeof(int), hipMemcpyDeviceToHost));
    std::vector<int> cell(nf), le(nf), un(3 * (size_t)nf), vn(2 * (size_t)nf);
    for (int i = 0; i < nf; ++i) {
        const int e = cl[2 * i], k = cl[2 * i + 1];
        if (e < 0 || e >= c->nel || k < 0 || k >= c->nvc) return fail(c, "facet (cell, local edge) out of range");
        cell[i] = e; le[i] = k;
        const int kb = (k + 1) % c->nvc;
        un[3 * i] = hp[(size_t)k * c->nel + e];
        un[3 * i + 1] = hp[(size_t)(c->cg1 ? k : c->nvc + k) * c->nel + e];      // CG1CG1: no mid-edge node (its rows of M2 are empty)
        un[3 * i + 2] = hp[(size_t)kb * c->nel + e];
        vn[2 * i] = hc[(size_t)k * c->nel + e];
        vn[2 * i + 1] = hc[(size_t)kb * c->nel + e];
    }
    HIPCHK(c, hipMalloc((void**)&c->fcell, nf * sizeof(int)));
    HIPCHK(c, hipMalloc((void**)&c->fledge, nf * sizeof(int)));
    HIPCHK(c, hipMalloc((void**)&c->funode, 3 * (size_t)nf * sizeof(int)));
    HIPCHK(c, hipMalloc((void**)&c->fvnode, 2 * (size_t)nf * sizeof(int)));
    HIPCHK(c, hipMalloc((void**)&c->fM2, 9 * (size_t)nf * sizeof(double)));
    HIPCHK(c, hipMalloc((void**)&c->fM1, 4 * (size_t)nf * sizeof(double)));
    HIPCHK(c, hipMemcpy(c->fcell, cell.data(), nf * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->fledge, le.data(), nf * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->funode, un.data(), un.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->fvnode, vn.data(), vn.size() * sizeof(int), hipMemcpyHostToDevice));
    if (c->cr) {
        // CG2CR1: the three rotation nodes (edge midpoints) of the facet's cell and room for the 3 x 3 rotation block
        std::vector<int> rn(3 * (size_t)nf);
        for (int i = 0; i < nf; ++i)
            for (int a = 0; a < 3; ++a) rn[3 * i + a] = hp[(size_t)(c->nvc + a) * c->nel + cl[2 * i]] - c->nn;
        HIPCHK(c, hipMalloc((void**)&c->frnode, rn.size() * sizeof(int)));
        HIPCHK(c, hipMemcpy(c->frnode, rn.data(), rn.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(c, hipMalloc((void**)&c->fMR, 9 * (size_t)nf * sizeof(double)));
    }
    c->nf = nf;
    return 0;
}

int femo_set_strong_dofs(femo_ctx* c, int32_t n, const int32_t* dofs) {
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<unsigned char> m((size_t)c->ndof, 0);
    for (int i = 0; i < n; ++i) {
        if (dofs[i] < 0 || dofs[i] >= c->ndof) return fail(c, "strong-BC dof out of range");
        m[dofs[i]] = 1;
    }
    HIPCHK(c, hipMemcpy(c->mask, m.data(), m.size(), hipMemcpyHostToDevice));
    c->has_mask = n > 0;
    operator_changed(c);
    return 0;
}

int femo_set_field(femo_ctx* c, const char* name, const double* v, int64_t n) {
    HIPCHK(c, hipSetDevice(c->device));
    if (name && std::string(name) == "dirichlet" && !c->gdir) {
        HIPCHK(c, hipMalloc((void**)&c->gdir, (size_t)c->ndof * sizeof(double)));
        HIPCHK(c, hipMemset(c->gdir, 0, (size_t)c->ndof * sizeof(double)));
    }
    int64_t len;
    double* d = field_ptr(c, name, &len);
    if (!d) return fail(c, std::string("unknown field '") + (name ? name : "") + "'");
    if (!v) return fail(c, "null values");
    if (n == 1 && len != 1) {   // update(): a length-1 array broadcasts (fea/utils_dolfinx.py:327-330)
        hipLaunchKernelGGL(k_fill, dim3(vec_grid(len)), dim3(256), 0, c->stream, d, v[0], len);
        HIPCHK(c, hipStreamSynchronize(c->stream));
    } else if (n == len) {
        HIPCHK(c, hipMemcpy(d, v, (size_t)len * sizeof(double), hipMemcpyHostToDevice));
    } else {
        char buf[160];
        snprintf(buf, sizeof buf, "field '%s' has length %lld, got %lld", name, (long long)len, (long long)n);
        return fail(c, buf);
    }
    if (d == c->uhat) {
        bool any = false;
        for (int64_t i = 0; i < n; ++i)
            if (v[i] != 0.0) { any = true; break; }
        c->has_uhat = any;
        c->penalty_dirty = true;
    }
    if (d == c->gdir) {
        bool any = false;
        for (int64_t i = 0; i < n; ++i)
            if (v[i] != 0.0) { any = true; break; }
        c->has_g = any;
        return 0;                                   // the prescribed values enter the right-hand side only
    }
    // the load never enters the operator; the density only through the inertia term aM M
    if (d != c->f && (d != c->rho || c->op_aM != 0.0)) operator_changed(c, true);
    return 0;
}

int femo_get_field(femo_ctx* c, const char* name, double* v, int64_t n) {
    HIPCHK(c, hipSetDevice(c->device));
    int64_t len;
    double* d = field_ptr(c, name, &len);
    if (!d) return fail(c, "unknown field");
    if (n != len) return fail(c, "length mismatch");
    HIPCHK(c, hipMemcpy(v, d, (size_t)len * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int femo_set_state(femo_ctx* c, const double* w) {
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy(c->w, w, (size_t)c->ndof * sizeof(double), hipMemcpyHostToDevice));
    return 0;
}

int femo_get_state(femo_ctx* c, double* w) {
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy(w, c->w, (size_t)c->ndof * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

// y = A x on device vectors (x is modified on masked rows only through a copy in tmp)
static int full_apply_dev(femo_ctx* c, const double* x, double* y) {
    const int64_t n = c->ndof;
    const int vg = vec_grid(n);
    hipLaunchKernelGGL(k_fill, dim3(vg), dim3(256), 0, c->stream, y, 0.0, n);
    const double* xin = x;
    if (c->has_mask) {
        HIPCHK(c, hipMemcpyAsync(c->tmp, x, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        hipLaunchKernelGGL(k_mask_zero, dim3(vg), dim3(256), 0, c->stream, c->tmp, c->mask, n);
        xin = c->tmp;
    }
    if (op_apply(c, xin, y, nullptr, nullptr, nullptr, true)) return 1;
    if (c->has_mask) hipLaunchKernelGGL(k_mask_identity, dim3(vg), dim3(256), 0, c->stream, y, x, c->mask, n);
    HIPCHK(c, hipGetLastError());
    return 0;
}

int femo_apply_K(femo_ctx* c, const double* x, double* y) {
    HIPCHK(c, hipSetDevice(c->device));
    const size_t bytes = (size_t)c->ndof * sizeof(double);
    HIPCHK(c, hipMemcpy(c->z, x, bytes, hipMemcpyHostToDevice));
    if (full_apply_dev(c, c->z, c->r)) return 1;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(y, c->r, bytes, hipMemcpyDeviceToHost));
    return 0;
}

int femo_load_vector(femo_ctx* c, double* F) {
    HIPCHK(c, hipSetDevice(c->device));
    if (load_vector_dev(c, c->b)) return 1;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(F, c->b, (size_t)c->ndof * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int femo_residual(femo_ctx* c, const double* w, double* r) {
    HIPCHK(c, hipSetDevice(c->device));
    const int64_t n = c->ndof;
    const double* wd = c->w;
    if (w) {
        HIPCHK(c, hipMemcpy(c->z, w, n * sizeof(double), hipMemcpyHostToDevice));
        wd = c->z;
    }
    if (full_apply_dev(c, wd, c->r)) return 1;
    if (load_vector_dev(c, c->b)) return 1;
    hipLaunchKernelGGL(k_axpby, dim3(vec_grid(n)), dim3(256), 0, c->stream, c->r, -1.0, c->b, 1.0, n);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(r, c->r, n * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int femo_diagonal(femo_ctx* c, double* d) {
    HIPCHK(c, hipSetDevice(c->device));
    c->jacobi_dirty = true;
    if (refresh_diag(c)) return 1;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<double> inv((size_t)c->ndof);
    HIPCHK(c, hipMemcpy(inv.data(), c->dinv, inv.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < inv.size(); ++i) d[i] = 1.0 / inv[i];
    return 0;
}

int femo_force_to_pressure(femo_ctx* c, const double* force, double* pressure, double rtol, int32_t maxit, int32_t* iters, double* relres) {
    HIPCHK(c, hipSetDevice(c->device));
    const int64_t n = 3 * (int64_t)c->nn;
    for (int i = 0; i < 6; ++i)
        if (!c->fp[i]) HIPCHK(c, hipMalloc((void**)&c->fp[i], (size_t)n * sizeof(double)));
    double *x = c->fp[0], *r = c->fp[1], *z = c->fp[2], *p = c->fp[3], *Ap = c->fp[4], *dg = c->fp[5];
    const int vg = vec_grid(n), eg = nblk(c->nel, 128);
    const MeshDev m = mesh_dev(c);
    auto apply = [&](const double* in, double* out, double* diag) {
        if (c->quad) hipLaunchKernelGGL(k_vmass_apply<4>, dim3(eg), dim3(128), 0, c->stream, m, in, out, diag);
        else hipLaunchKernelGGL(k_vmass_apply<3>, dim3(eg), dim3(128), 0, c->stream, m, in, out, diag);
    };
    auto dot = [&](const double* a, const double* b, double* out) -> int {
        HIPCHK(c, hipMemsetAsync(c->scal + 7, 0, sizeof(double), c->stream));
        hipLaunchKernelGGL(k_dot, dim3(red_grid(n)), dim3(256), 0, c->stream, a, b, n, c->scal + 7);
        HIPCHK(c, hipMemcpyAsync(c->scal_host + 7, c->scal + 7, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        *out = c->scal_host[7];
        return 0;
    };
    HIPCHK(c, hipMemcpyAsync(r, force, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(dg, 0, (size_t)n * sizeof(double), c->stream));
    HIPCHK(c, hipMemsetAsync(x, 0, (size_t)n * sizeof(double), c->stream));
    apply(nullptr, nullptr, dg);
    double bb = 0, rr = 0, rz = 0, rz_old = 0, pAp = 0;
    if (dot(r, r, &bb)) return 1;
    rr = bb;
    int k = 0;
    const double target = rtol * rtol * bb;
    while (bb > 0 && rr > target && k < maxit) {
        hipLaunchKernelGGL(k_div, dim3(vg), dim3(256), 0, c->stream, z, (const double*)r, (const double*)dg, n);
        if (dot(r, z, &rz)) return 1;
        if (k == 0) HIPCHK(c, hipMemcpyAsync(p, z, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        else hipLaunchKernelGGL(k_axpby, dim3(vg), dim3(256), 0, c->stream, p, 1.0, (const double*)z, rz / rz_old, n);     // p = z + beta p
        HIPCHK(c, hipMemsetAsync(Ap, 0, (size_t)n * sizeof(double), c->stream));
        apply(p, Ap, nullptr);
        if (dot(p, Ap, &pAp)) return 1;
        if (!(pAp > 0)) return fail(c, "force -> pressure: the mass matrix is not positive definite (degenerate cells?)");
        const double alpha = rz / pAp;
        hipLaunchKernelGGL(k_axpby, dim3(vg), dim3(256), 0, c->stream, x, alpha, (const double*)p, 1.0, n);
        hipLaunchKernelGGL(k_axpby, dim3(vg), dim3(256), 0, c->stream, r, -alpha, (const double*)Ap, 1.0, n);
        if (dot(r, r, &rr)) return 1;
        rz_old = rz;
        ++k;
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(pressure, x, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (iters) *iters = k;
    if (relres) *relres = bb > 0 ? sqrt(rr / bb) : 0.0;
    if (rr > target && c->opt.strict) {
        char buf[160];
        snprintf(buf, sizeof buf, "force -> pressure: PCG stopped at maxit = %d with relative residual %.3e", (int)maxit, sqrt(rr / bb));
        c->err = buf;
        return 4;
    }
    return 0;
}

int femo_element_matrices(femo_ctx* c, int32_t first, int32_t count, double* Ke) {
    HIPCHK(c, hipSetDevice(c->device));
    if (first < 0 || count < 0 || first + count > c->nel) return fail(c, "element range out of bounds");
    if (count == 0) return 0;
    double* d = nullptr;
    const size_t bytes = (size_t)count * c->ld * c->ld * sizeof(double);
    HIPCHK(c, hipMalloc((void**)&d, bytes));
    ELEM_LAUNCH_S(c, k_element_matrices, NOEXTRA, count, 64, QPOINT_LDS(c), mesh_dev(c), fields_dev(c), c->tab, first, count, d);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(Ke, d, bytes, hipMemcpyDeviceToHost);
    hipFree(d);
    HIPCHK(c, e);
    return 0;
}

int femo_set_solver(femo_ctx* c, int preconditioner, double rtol, int32_t maxit, int32_t check_every) {
    if (preconditioner != 0 && preconditioner != 2) return fail(c, "preconditioner must be 0 (Jacobi) or 2 (multifrontal Cholesky)");
    if (preconditioner == 2 && !c->fr.ready) return fail(c, "preconditioner 2 needs femo_set_frontal_plan first");
    if (!(rtol > 0) || maxit < 1 || check_every < 1) return fail(c, "bad solver parameters");
    c->precond = preconditioner; c->rtol = rtol; c->maxit = maxit; c->check_every = check_every;
    return 0;
}

int femo_set_option(femo_ctx* c, const char* key, double value) {
    const std::string k(key ? key : "");
    const int v = (int)value;
    auto& o = c->opt;
    ++c->opt_version;                        // captured launch sequences are stale
    if (k == "trailing") { if (v < 0 || v > 2) return fail(c, "trailing: 0 auto, 1 left-looking, 2 right-looking"); o.trailing = v; }
    else if (k == "left_min") o.left_min = v;
    else if (k == "left_max") o.left_max = v;
    else if (k == "lookahead") o.lookahead = v != 0;
    else if (k == "lookahead_cnt") o.lookahead_cnt = v;
    else if (k == "super_panel") { if (v < 0) return fail(c, "super_panel: a column count (rounded down to whole 128-column panels; < 256 switches it off)"); o.super_panel = v; }
    else if (k == "super_panel_cnt") o.super_panel_cnt = v;
    else if (k == "super_panel_ahead") o.super_panel_ahead = v != 0;
    else if (k == "diag_ahead") o.diag_ahead = v != 0;
    else if (k == "rows_fine_wg") o.rows_fine_wg = v;
    else if (k == "narrow_fine_wg") o.narrow_fine_wg = v;
    else if (k == "rows_preload_wg") o.rows_preload_wg = v;
    else if (k == "narrow_split") { if (v < 1 || v > 32) return fail(c, "narrow_split: 1..32 slices of the K range"); o.narrow_split = v; }
    else if (k == "narrow_split_wg") o.narrow_split_wg = v;
    else if (k == "split_cnt") o.split_cnt = v;
    else if (k == "diag_v1_cnt") o.diag_v1_cnt = v;
    else if (k == "sweep_graph") o.sweep_graph = v != 0;
    else if (k == "fuse_rows") o.fuse_rows = v != 0;
    else if (k == "fuse_rows_cnt") o.fuse_rows_cnt = v;
    else if (k == "fuse_rows_np") o.fuse_rows_np = v;
    else if (k == "split_groups") { if (v < 2) return fail(c, "split_groups: at least 2"); o.split_groups = v; }
    else if (k == "super_tiles") o.super_tiles = v != 0;
    else if (k == "super_tiles_min") o.super_tiles_min = v;
    else if (k == "fused_schur") o.fused_schur = v != 0;
    else if (k == "precond_nquad") { if (v != 0 && (v < 2 || v > 6)) return fail(c, "precond_nquad: 0 (the operator's rule) or 2..6"); o.precond_nquad = v; operator_changed(c); }
    else if (k == "equilibrate") { if (v < 0 || v > 2) return fail(c, "equilibrate: 0 off, 1 diag^-1/2, 2 nearest powers of two"); o.equilibrate = v; operator_changed(c); }
    else if (k == "grid_chunk") { if (v < 1 || v > 65535) return fail(c, "grid_chunk must be in 1..65535"); o.grid_chunk = v; }
    else if (k == "wide_np" || k == "wide_cnt") {
        if (c->fr.ready) return fail(c, "wide_np / wide_cnt shape the plan: set them before femo_set_frontal_plan");
        if (v < 0) return fail(c, "negative threshold");
        (k == "wide_np" ? o.wide_np : o.wide_cnt) = v;
    }
    else if (k == "strict") o.strict = v != 0;
    else if (k == "allow_pivot_repair") o.allow_pivot_repair = v != 0;
    else if (k == "stress_regularization") { if (value < 0) return fail(c, "stress_regularization: a coefficient >= 0 (the reference's is 0.5e3)"); c->stress_reg = value; }
    else if (k == "profile_verbose") o.profile_verbose = v != 0;
    else if (k == "big_tiles") o.big_tiles = v != 0;
    else if (k == "big_min_wg") o.big_min_wg = v;
    else if (k == "strip_cnt") o.strip_cnt = v;
    else if (k == "strip_kmax") o.strip_kmax = v;
    else if (k == "strip_depth") { if (v < 1 || v > 2) return fail(c, "strip_depth: 1..2 blocks of look-ahead"); o.strip_depth = v; }
    else if (k == "diag_v1") { if (v < 0 || v > 3) return fail(c, "diag_v1: 0 auto, 1 round-2 kernel, 2 overlapped kernel, 3 the rule before the LDS diet"); o.diag_v1 = v; }
    else if (k == "profile") c->fr.profile = v != 0;      // event pair around every factorisation launch until switched off (femo_factorize_profile_get)
    else if (k == "bnd_tiled_nb") o.bnd_tiled_nb = v;
    else if (k == "sweep_butterfly") o.sweep_butterfly = v;
    else if (k == "sweep_fuse") o.sweep_fuse = v != 0;
    else if (k == "assemble_fc") { if (v < 0 || v > 2) return fail(c, "assemble_fc: 0 never, 1 where it pays, 2 always"); o.assemble_fc = v; }
    else if (k == "sweep_w") { if ((v != 0) != (o.sweep_w != 0)) { o.sweep_w = v != 0; operator_changed(c); } }
    else if (k == "multi_rhs") { o.multi_rhs = v != 0; }
    else if (k == "diag_t") { o.diag_t = v; }
    else if (k == "apply_lanes") { if (v != 0 && v != 4 && v != 5) return fail(c, "apply_lanes: 0 or 4 (a quad of lanes per element), or 5"); o.apply_lanes = v; }
    else if (k == "sweep_ahead") { if (v < 0) return fail(c, "sweep_ahead: number of top levels left to the solve (0: off)"); o.sweep_ahead = v; }
    else if (k == "stale_rel") { if (!(value >= 0)) return fail(c, "stale_rel: a relative change >= 0"); o.stale_rel = value; }
    else if (k == "stale_factor") { if (v < 0) return fail(c, "stale_factor: PCG iterations a kept factor is given before the factorisation is refreshed (0: never keep)"); o.stale_factor = v; }
    else if (k == "sweep_read_mode") { if (v < 0 || v > 2) return fail(c, "sweep_read_mode: 0 returning atomic, 1 agent-scope load, 2 plain load (experiment)"); o.sweep_read_mode = v; }
    else if (k == "swork_slots") { if (c->fr.ready || v < 1) return fail(c, "swork_slots >= 1, before femo_set_frontal_plan"); o.swork_slots = v; }
    else if (k == "xinv_small_cnt") o.xinv_small_cnt = v;
    else return fail(c, "unknown option '" + k + "'");
    return 0;
}

int femo_set_krylov(femo_ctx* c, int method) {
    if (method != 0 && method != 1) return fail(c, "Krylov method must be 0 (conjugate gradients) or 1 (BiCGStab)");
    c->krylov = method;
    return 0;
}

int femo_solve_state(femo_ctx* c, int zero_guess, int32_t* iters, double* relres) {
    HIPCHK(c, hipSetDevice(c->device));
    if (load_vector_dev(c, c->b)) return 1;
    return solve_dispatch(c, c->b, c->w, zero_guess != 0, iters, relres);
}

int femo_solve_linear(femo_ctx* c, const double* rhs, double* x, int32_t* iters, double* relres) {
    HIPCHK(c, hipSetDevice(c->device));
    const size_t bytes = (size_t)c->ndof * sizeof(double);
    HIPCHK(c, hipMemcpy(c->b, rhs, bytes, hipMemcpyHostToDevice));
    if (int rc = solve_dispatch(c, c->b, c->lam, true, iters, relres)) return rc;
    HIPCHK(c, hipMemcpy(x, c->lam, bytes, hipMemcpyDeviceToHost));
    return 0;
}

// whole-mesh integrals; restricted to the selected sub-domain only when asked (tip_disp, area: the reference's dxx(i))
static MeshDev mesh_dev_all(const femo_ctx* c) { MeshDev m = mesh_dev(c); m.csel = -1; return m; }

static int functionals_dev(femo_ctx* c, double* out3, int nout = 3, bool subdomain = false) {
    HIPCHK(c, hipMemsetAsync(c->scal, 0, 8 * sizeof(double), c->stream));
    ELEM_LAUNCH(c, k_functionals, NOEXTRA, nblk(c->nel, EB), EB, subdomain ? mesh_dev(c) : mesh_dev_all(c), fields_dev(c), c->tab, c->w, c->scal);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->scal_host, c->scal, 8 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < nout; ++i) out3[i] = c->scal_host[i];
    return 0;
}

// reference area the stress aggregate is divided by: of the whole mesh or of the selected sub-domain
static double& stress_alpha_ref(femo_ctx* c) { return c->csel < 0 ? c->stress_alpha : c->alpha_tag[c->csel]; }

// int (m vm)^rho J dx over the cells and (first call) the reference area alpha
static int pnorm_dev(femo_ctx* c, double out2[2]) {
    HIPCHK(c, hipMemsetAsync(c->scal, 0, 8 * sizeof(double), c->stream));
    ELEM_LAUNCH(c, k_pnorm, NOEXTRA, nblk(c->nel, EB), EB, mesh_dev(c), fields_dev(c), c->tab_s, 0, c->stress_m, c->stress_rho, 1.0, c->stress_reg,
                c->w, (double*)nullptr, c->scal);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->scal_host, c->scal, 8 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    out2[0] = c->scal_host[0]; out2[1] = c->scal_host[1];
    if (stress_alpha_ref(c) < 0) stress_alpha_ref(c) = out2[1];
    return 0;
}

int femo_functional(femo_ctx* c, const char* name, double* value) {
    HIPCHK(c, hipSetDevice(c->device));
    const std::string s(name ? name : "");
    if (s == "compliance" || s == "mass" || s == "volume" || s == "regularization") {
        double v[4];
        if (functionals_dev(c, v, 4)) return 1;
        *value = s == "mass" ? v[2] : s == "volume" ? v[3] : s == "regularization" ? v[1] : v[0] + v[1];
        return 0;
    }
    if (s == "tip_disp" || s == "area") {          // over the selected sub-domain (rm_shell_pde.py:95-96, 104-105)
        double v[5];
        if (functionals_dev(c, v, 5, true)) return 1;
        *value = s == "area" ? v[4] : 0.5 * v[0];
        return 0;
    }
    if (s == "elastic_energy") {
        const int64_t n = c->ndof;
        HIPCHK(c, hipMemsetAsync(c->scal, 0, 8 * sizeof(double), c->stream));
        hipLaunchKernelGGL(k_fill, dim3(vec_grid(n)), dim3(256), 0, c->stream, c->r, 0.0, n);
        if (op_apply(c, c->w, c->r, c->scal + 7, nullptr, nullptr, false)) return 1;
        HIPCHK(c, hipMemcpyAsync(c->scal_host, c->scal, 8 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        *value = 0.5 * c->scal_host[7];
        return 0;
    }
    if (s == "pnorm_stress") {
        double v[2];
        if (pnorm_dev(c, v)) return 1;
        *value = v[0] / stress_alpha_ref(c);
        return 0;
    }
    if (s.rfind("sum_stress_", 0) == 0) {         // over the selected sub-domain (rm_shell_pde.py:130-150)
        static const char* comps[6] = {"x", "y", "z", "xy", "xz", "yz"};
        int k = -1;
        for (int i = 0; i < 6; ++i) if (s.substr(11) == comps[i]) k = i;
        if (k < 0) return fail(c, "unknown stress component in '" + s + "' (x, y, z, xy, xz, yz)");
        HIPCHK(c, hipMemsetAsync(c->scal, 0, 8 * sizeof(double), c->stream));
        ELEM_LAUNCH(c, k_stress_sums, NOEXTRA, nblk(c->nel, EB), EB, mesh_dev(c), fields_dev(c), c->tab, c->w, c->scal);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(c->scal_host, c->scal, 8 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        *value = c->scal_host[k];
        return 0;
    }
    return fail(c, "unknown functional '" + s + "'");
}

// out (3 nn) += scale * d/d uhat of: mode 0 lam.(K w - F) [+ penalty], 1 int u.u J, 2 mass, 3 elastic energy
static int shape_gradient_dev(femo_ctx* c, int mode, const double* w, const double* lam, double scale, double* out) {
    const MeshDev m = mesh_dev(c);
    const FieldsDev f = fields_dev(c);
    const int nthreads = c->nel * 3 * c->nvc;
    const Tables* tb = mode == 4 ? c->tab_s : c->tab;
#define SHAPE_LAUNCH(NPC, NVC, QUAD)                                                                                              \
    hipLaunchKernelGGL((k_shape_gradient<NPC, NVC, QUAD>), dim3(nblk(nthreads, 128)), dim3(128), 0, c->stream, m, f, tb, mode, w, lam, \
                       scale, c->stress_m, c->stress_rho, c->stress_reg, out)
    if (c->cg1) { if (c->quad) SHAPE_LAUNCH(4, 4, true); else SHAPE_LAUNCH(3, 3, false); }
    else        { if (c->quad) SHAPE_LAUNCH(9, 4, true); else SHAPE_LAUNCH(6, 3, false); }
#undef SHAPE_LAUNCH
    if (mode == 0 && c->nf > 0) {
        if (c->has_g) {           // the penalty term is P(uhat) (w - g)
            hipLaunchKernelGGL(k_lincomb3, dim3(vec_grid(c->ndof)), dim3(256), 0, c->stream, c->tmp, 1.0, w, -1.0, (const double*)c->gdir, 0.0,
                               (const double*)nullptr, (int64_t)c->ndof);
            w = c->tmp;
        }
        const int nt = c->nf * 3 * c->nvc;
#define SHAPE_PENALTY(NVC, QUAD, CG1)                                                                                                  \
    hipLaunchKernelGGL((k_shape_gradient_penalty<NVC, QUAD, CG1>), dim3(nblk(nt, 64)), dim3(64), 0, c->stream, m, f, facet_dev(c), c->beta, \
                       w, lam, scale, out)
        if (c->cg1) { if (c->quad) SHAPE_PENALTY(4, true, true); else SHAPE_PENALTY(3, false, true); }
        else        { if (c->quad) SHAPE_PENALTY(4, true, false); else SHAPE_PENALTY(3, false, false); }
#undef SHAPE_PENALTY
    }
    HIPCHK(c, hipGetLastError());
    return 0;
}

// gradient of a functional into a device buffer `out` (length n, zero-filled here)
static int dfunctional_dev(femo_ctx* c, const std::string& fn, const std::string& wrt, double* out, int64_t n) {
    int64_t len;
    if (wrt == "disp_solid") len = c->ndof;
    else if (!field_ptr(c, wrt.c_str(), &len)) return fail(c, "unknown argument '" + wrt + "'");
    if (len != n) return fail(c, "gradient buffer has the wrong length for '" + wrt + "'");
    hipLaunchKernelGGL(k_fill, dim3(vec_grid(n)), dim3(256), 0, c->stream, out, 0.0, n);
    const MeshDev m = mesh_dev(c);
    const FieldsDev f = fields_dev(c);
    const int g = nblk(c->nel, EB);
    if (wrt == "uhat") {
        if (fn == "regularization") return 0;                 // integrates over the reference configuration only
        const int mode = fn == "compliance" ? 1 : fn == "mass" ? 2 : fn == "elastic_energy" ? 3 : fn == "pnorm_stress" ? 4 : -1;
        if (mode < 0) return fail(c, "unknown functional '" + fn + "'");
        if (mode == 4) {
            if (stress_alpha_ref(c) < 0) { double v[2]; if (pnorm_dev(c, v)) return 1; hipLaunchKernelGGL(k_fill, dim3(vec_grid(n)), dim3(256), 0, c->stream, out, 0.0, n); }
            return shape_gradient_dev(c, 4, c->w, nullptr, 1.0 / stress_alpha_ref(c), out);
        }
        return shape_gradient_dev(c, mode, c->w, nullptr, 1.0, out);
    }
    if (fn == "tip_disp") {               // 0.5 int u.u J over the selected sub-domain
        if (wrt == "disp_solid") ELEM_LAUNCH(c, k_dcompliance_du, NOEXTRA, g, EB, m, f, c->tab, c->w, out, 0.5);
    } else if (fn == "area") {
    } else if (fn == "regularization") {         // the thickness term of the compliance (its only explicit thickness dependence)
        if (wrt == "thickness") ELEM_LAUNCH(c, k_field_grad, NOEXTRA, g, EB, m, f, c->tab, 0, out);
    } else if (fn == "compliance") {
        if (wrt == "disp_solid") ELEM_LAUNCH(c, k_dcompliance_du, NOEXTRA, g, EB, mesh_dev_all(c), f, c->tab, c->w, out, 1.0);
        else if (wrt == "thickness") ELEM_LAUNCH(c, k_field_grad, NOEXTRA, g, EB, m, f, c->tab, 0, out);
    } else if (fn == "mass") {
        if (wrt == "thickness") ELEM_LAUNCH(c, k_field_grad, NOEXTRA, g, EB, m, f, c->tab, 1, out);
        else if (wrt == "density") ELEM_LAUNCH(c, k_field_grad, NOEXTRA, g, EB, m, f, c->tab, 2, out);
    } else if (fn == "volume") {
        if (wrt == "thickness") ELEM_LAUNCH(c, k_field_grad, NOEXTRA, g, EB, m, f, c->tab, 3, out);
    } else if (fn == "elastic_energy") {
        if (wrt == "disp_solid") { if (op_apply(c, c->w, out, nullptr, nullptr, nullptr, false)) return 1; }
        else if (wrt == "thickness") ELEM_LAUNCH(c, k_dRdfield_T, COMMA_H, g, EB, m, f, c->tab, c->w, c->w, 0.5, out);
        else if (wrt == "E") ELEM_LAUNCH(c, k_dRdfield_T, COMMA_E, g, EB, m, f, c->tab, c->w, c->w, 0.5, out);
        else if (wrt == "nu") ELEM_LAUNCH(c, k_dRdfield_T, COMMA_NU, g, EB, m, f, c->tab, c->w, c->w, 0.5, out);
    } else if (fn == "pnorm_stress") {
        if (stress_alpha_ref(c) < 0) { double v[2]; if (pnorm_dev(c, v)) return 1; hipLaunchKernelGGL(k_fill, dim3(vec_grid(n)), dim3(256), 0, c->stream, out, 0.0, n); }
        const int mode = wrt == "disp_solid" ? 1 : wrt == "thickness" ? 2 : wrt == "E" ? 3 : wrt == "nu" ? 4 : 0;
        if (mode)
            ELEM_LAUNCH(c, k_pnorm, NOEXTRA, g, EB, m, f, c->tab_s, mode, c->stress_m, c->stress_rho, 1.0 / stress_alpha_ref(c), c->stress_reg, c->w, out,
                        (double*)nullptr);
    } else {
        return fail(c, "unknown functional '" + fn + "'");
    }
    HIPCHK(c, hipGetLastError());
    return 0;
}

static int dRdarg_T_dev(femo_ctx* c, const std::string& arg, const double* lam, double scale, double* out, int64_t n) {
    int64_t len;
    if (!field_ptr(c, arg.c_str(), &len)) return fail(c, "unknown argument '" + arg + "'");
    if (len != n) return fail(c, "buffer has the wrong length for '" + arg + "'");
    const MeshDev m = mesh_dev(c);
    const FieldsDev f = fields_dev(c);
    const int g = nblk(c->nel, EB);
    if (arg == "thickness") ELEM_LAUNCH(c, k_dRdfield_T, COMMA_H, g, EB, m, f, c->tab, c->w, lam, scale, out);
    else if (arg == "E") ELEM_LAUNCH(c, k_dRdfield_T, COMMA_E, g, EB, m, f, c->tab, c->w, lam, scale, out);
    else if (arg == "nu") ELEM_LAUNCH(c, k_dRdfield_T, COMMA_NU, g, EB, m, f, c->tab, c->w, lam, scale, out);
    else if (arg == "F_solid") ELEM_LAUNCH(c, k_dRdf_T, NOEXTRA, g, EB, m, f, c->tab, lam, -scale, out);
    else if (arg == "density") { /* R does not depend on density */ }
    else if (arg == "uhat") { if (shape_gradient_dev(c, 0, c->w, lam, scale, out)) return 1; }
    else return fail(c, "(dR/d" + arg + ")^T is not implemented in this build");
    HIPCHK(c, hipGetLastError());
    return 0;
}

int femo_dfunctional(femo_ctx* c, const char* name, const char* wrt, double* out, int64_t n) {
    HIPCHK(c, hipSetDevice(c->device));
    double* d = nullptr;
    HIPCHK(c, hipMalloc((void**)&d, std::max<int64_t>(n, 1) * sizeof(double)));
    int rc = dfunctional_dev(c, name ? name : "", wrt ? wrt : "", d, n);
    if (!rc) {
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) e = hipMemcpy(out, d, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { c->err = hipGetErrorString(e); rc = 1; }
    }
    hipFree(d);
    return rc;
}

int femo_dRdarg_T(femo_ctx* c, const char* arg, const double* lambda, double* out, int64_t n) {
    HIPCHK(c, hipSetDevice(c->device));
    double* d = nullptr;
    HIPCHK(c, hipMalloc((void**)&d, std::max<int64_t>(n, 1) * sizeof(double)));
    hipMemsetAsync(d, 0, std::max<int64_t>(n, 1) * sizeof(double), c->stream);
    hipMemcpyAsync(c->lam, lambda, (size_t)c->ndof * sizeof(double), hipMemcpyHostToDevice, c->stream);
    int rc = dRdarg_T_dev(c, arg ? arg : "", c->lam, 1.0, d, n);
    if (!rc) {
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) e = hipMemcpy(out, d, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { c->err = hipGetErrorString(e); rc = 1; }
    }
    hipFree(d);
    return rc;
}

int femo_total_gradient(femo_ctx* c, const char* functional, const char* arg, double* out, int64_t n, int32_t* iters,
                        double* relres) {
    HIPCHK(c, hipSetDevice(c->device));
    const std::string fn(functional ? functional : ""), a(arg ? arg : "");
    double* d = nullptr;
    HIPCHK(c, hipMalloc((void**)&d, std::max<int64_t>(n, 1) * sizeof(double)));
    int rc = dfunctional_dev(c, fn, "disp_solid", c->b, c->ndof);          // dJ/dw
    if (!rc) rc = solve_dispatch(c, c->b, c->lam, true, iters, relres);     // lambda = K^-1 dJ/dw
    if (!rc) rc = dfunctional_dev(c, fn, a, d, n);                          // dJ/d arg (zero-fills d)
    if (!rc) rc = dRdarg_T_dev(c, a, c->lam, -1.0, d, n);                   // - (dR/d arg)^T lambda
    if (!rc) {
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) e = hipMemcpy(out, d, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { c->err = hipGetErrorString(e); rc = 1; }
    }
    hipFree(d);
    return rc;
}


// Several linear solves with the state operator at once.  With the multifrontal preconditioner the right-hand sides travel through the
// triangular sweeps in groups of up to four (sweeps_multi.h: the factor bytes are read once per group); any other solver setting, and
// the experimental sweep forms that rewrite the factor, take them one after the other.
static bool multi_sweeps_apply(const femo_ctx* c) {
    return c->precond == 2 && c->krylov == 0 && c->fr.ready && c->opt.sweep_w == 0 && !c->opt.equilibrate && c->opt.multi_rhs != 0;
}

// B: nrhs device vectors (overwritten), X: nrhs device vectors
static int solve_multi_dev(femo_ctx* c, int nrhs, double* const* B, double* const* X, int32_t* iters, double* relres) {
    int rc_all = 0;
    float t_fac = 0;
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    for (int r0 = 0; r0 < nrhs;) {
        const int g = multi_sweeps_apply(c) ? std::min(4, nrhs - r0) : 1;
        int rc;
        if (g == 1) rc = solve_dispatch(c, B[r0], X[r0], true, iters ? iters + r0 : nullptr, relres ? relres + r0 : nullptr);
        else rc = pcg_frontal_group(c, g, B + r0, X + r0, iters ? iters + r0 : nullptr, relres ? relres + r0 : nullptr);
        if (rc && rc != 4) return rc;
        if (rc) rc_all = rc;
        r0 += g;
    }
    (void)t_fac;
    return rc_all;
}

int femo_solve_linear_multi(femo_ctx* c, int32_t nrhs, const double* rhs, double* x, int32_t* iters, double* relres) {
    HIPCHK(c, hipSetDevice(c->device));
    if (nrhs < 1 || !rhs || !x) return fail(c, "femo_solve_linear_multi: nrhs >= 1 right-hand sides, one vector after the other");
    const size_t n = (size_t)c->ndof;
    double* Bd = mr_io_buffer(c, 2 * (size_t)nrhs * n);
    if (!Bd) return 1;
    double* Xd = Bd + (size_t)nrhs * n;
    std::vector<double*> B(nrhs), X(nrhs);
    for (int r = 0; r < nrhs; ++r) { B[r] = Bd + r * n; X[r] = Xd + r * n; }
    int rc = 0;
    if (hipMemcpy(Bd, rhs, (size_t)nrhs * n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) rc = fail(c, "copy of the right-hand sides failed");
    if (!rc) rc = solve_multi_dev(c, nrhs, B.data(), X.data(), iters, relres);
    if (!rc || rc == 4) {
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) e = hipMemcpy(x, Xd, (size_t)nrhs * n * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { c->err = hipGetErrorString(e); rc = 1; }
    }
    return rc;
}

// Total derivatives of SEVERAL functionals of the state with respect to one argument: the adjoint right-hand sides dJ_i/dw are known
// together, so their solves share the sweeps (the reference solves one adjoint per output: state_operation.py:188-220 called once per
// registered output of `disp_solid`, rm_shell_model.py:221-253).  subdomains[i] restricts functional i to a tagged sub-domain
// (femo_set_cell_tags; -1 or a NULL array: the whole mesh) -- the reference's pnorm_stress_<tag>.  out: nfun x n, row i = d J_i / d arg.
int femo_total_gradients(femo_ctx* c, int32_t nfun, const char* const* functionals, const int32_t* subdomains, const char* arg, double* out,
                         int64_t n, int32_t* iters, double* relres) {
    HIPCHK(c, hipSetDevice(c->device));
    if (nfun < 1 || !functionals || !out) return fail(c, "femo_total_gradients: nfun >= 1 functional names");
    const std::string a(arg ? arg : "");
    const size_t nd = (size_t)c->ndof;
    const int keep_sel = c->csel;
    for (int i = 0; i < nfun; ++i)
        if (subdomains && (subdomains[i] < -1 || subdomains[i] >= c->ntags)) return fail(c, "unknown sub-domain");
    double* Bd = mr_io_buffer(c, (size_t)nfun * (2 * nd + (size_t)std::max<int64_t>(n, 1)));
    if (!Bd) return 1;
    double *Xd = Bd + (size_t)nfun * nd, *Gd = Xd + (size_t)nfun * nd;
    std::vector<double*> B(nfun), X(nfun);
    for (int i = 0; i < nfun; ++i) { B[i] = Bd + i * nd; X[i] = Xd + i * nd; }
    int rc = 0;
    for (int i = 0; i < nfun && !rc; ++i) {
        c->csel = subdomains ? subdomains[i] : -1;
        rc = dfunctional_dev(c, functionals[i] ? functionals[i] : "", "disp_solid", B[i], c->ndof);      // dJ_i/dw
    }
    if (!rc) { rc = solve_multi_dev(c, nfun, B.data(), X.data(), iters, relres); }                       // lambda_i = K^-1 dJ_i/dw
    for (int i = 0; i < nfun && !rc; ++i) {
        c->csel = subdomains ? subdomains[i] : -1;
        double* g = Gd + (size_t)i * n;
        rc = dfunctional_dev(c, functionals[i], a, g, n);                                                // dJ_i/d arg (zero-fills g)
        if (!rc) rc = dRdarg_T_dev(c, a, X[i], -1.0, g, n);                                              // - (dR/d arg)^T lambda_i
    }
    c->csel = keep_sel;
    if (!rc) {
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) e = hipMemcpy(out, Gd, (size_t)nfun * n * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { c->err = hipGetErrorString(e); rc = 1; }
    }
    return rc;
}

int femo_set_frontal_plan(femo_ctx* c, int32_t ntree, int32_t nlevels, const int32_t* nf, const int32_t* npiv,
                          const int64_t* front_off, const int64_t* dof_off, const int32_t* front_dofs, const int32_t* up_map,
                          const int32_t* parent, const int32_t* left, const int32_t* right, const int32_t* level_off,
                          const int32_t* level_nodes, const int32_t* elem_front, const int32_t* elem_map) {
    HIPCHK(c, hipSetDevice(c->device));
    auto& fr = c->fr;
    if (fr.ready) return fail(c, "frontal plan already set for this context");
    const int WIDE_NP = std::max(1, c->opt.wide_np), WIDE_CNT = std::max(0, c->opt.wide_cnt);
    if (ntree < 1 || nlevels < 1) return fail(c, "empty frontal plan");
    fr.ntree = ntree; fr.nlevels = nlevels;
    fr.h_nf.assign(nf, nf + ntree); fr.h_npiv.assign(npiv, npiv + ntree);
    fr.h_level_off.assign(level_off, level_off + nlevels + 1);
    fr.h_level_nodes.assign(level_nodes, level_nodes + ntree);
    if (level_off[0] != 0 || level_off[nlevels] != ntree) return fail(c, "level_off does not cover all fronts");
    for (int L = 0; L < nlevels; ++L)
        if (level_off[L + 1] < level_off[L]) return fail(c, "level_off must be non-decreasing");
    {
        std::vector<char> seen(ntree, 0);                 // every front is factorised exactly once
        for (int i = 0; i < ntree; ++i) {
            if (level_nodes[i] < 0 || level_nodes[i] >= ntree) return fail(c, "level_nodes out of range");
            if (seen[level_nodes[i]]++) return fail(c, "level_nodes lists a front twice");
        }
    }
    // inside a level the fronts are kept in order of pivot count: the diagonal-block kernel is launched per class of
    // equal sub-block count (its LDS footprint, hence its occupancy, depends on it)
    for (int L = 0; L < nlevels; ++L)
        std::stable_sort(fr.h_level_nodes.begin() + level_off[L], fr.h_level_nodes.begin() + level_off[L + 1],
                         [&](int a, int b) { return npiv[a] < npiv[b]; });
    level_nodes = fr.h_level_nodes.data();
    long long piv_total = 0;
    std::vector<long long> linvoff(ntree + 1, 0);
    fr.max_nf = 0;
    for (int t = 0; t < ntree; ++t) {
        if (npiv[t] < 0 || nf[t] < npiv[t] || front_off[t + 1] - front_off[t] != (long long)nf[t] * nf[t] ||
            dof_off[t + 1] - dof_off[t] != nf[t])
            return fail(c, "inconsistent frontal plan (sizes / offsets)");
        piv_total += npiv[t];
        linvoff[t + 1] = linvoff[t] + (long long)((npiv[t] + NB - 1) / NB) * NB * NB;
        fr.max_nf = std::max(fr.max_nf, nf[t]);
    }
    if (piv_total != c->ndof) return fail(c, "frontal plan does not eliminate every DOF exactly once");
    if ((size_t)(fr.max_nf + NB) * sizeof(double) > 150 * 1024) return fail(c, "largest front does not fit the LDS solve kernels");
    const long long ndofs_total = dof_off[ntree];
    for (long long i = 0; i < ndofs_total; ++i)
        if (front_dofs[i] < 0 || front_dofs[i] >= c->ndof) return fail(c, "front_dofs out of range");
    for (int t = 0; t < ntree; ++t) {
        const int p = parent[t];
        if (p < -1 || p >= ntree) return fail(c, "parent out of range");
        for (long long i = dof_off[t] + npiv[t]; i < dof_off[t + 1]; ++i)
            if (p < 0 || up_map[i] < 0 || up_map[i] >= nf[p]) return fail(c, "up_map out of range");
    }
    for (long long e = 0; e < c->nel; ++e) {
        const int t = elem_front[e];
        if (t < 0 || t >= ntree) return fail(c, "elem_front out of range");
        for (int i = 0; i < c->ld; ++i)
            if (elem_map[e * c->ld + i] < 0 || elem_map[e * c->ld + i] >= nf[t]) return fail(c, "elem_map out of range");
    }
    fr.h_level_maxnp.assign(nlevels, 0); fr.h_level_maxnb.assign(nlevels, 0);
    for (int L = 0; L < nlevels; ++L)
        for (int i = level_off[L]; i < level_off[L + 1]; ++i) {
            const int t = level_nodes[i];
            if (t < 0 || t >= ntree) return fail(c, "level_nodes out of range");
            fr.h_level_maxnp[L] = std::max(fr.h_level_maxnp[L], npiv[t]);
            fr.h_level_maxnb[L] = std::max(fr.h_level_maxnb[L], nf[t] - npiv[t]);
        }
    fr.h_level_wide.assign(nlevels, 0);
    for (int L = 0; L < nlevels; ++L)
        fr.h_level_wide[L] = fr.h_level_maxnp[L] > WIDE_NP || level_off[L + 1] - level_off[L] <= WIDE_CNT;
    // Storage of the fronts.  Pivot columns (the factor): one nf x npiv panel per front, for good.  Schur complements: an
    // arena.  The block of front t is written at t's level and read once, at its parent's level (by the extend-add or by
    // the rank-k updates that gather their columns from the children, any time during that level); all
    // blocks whose parents share a level form one region, alive from the lowest level of its fronts to that parent level,
    // and regions are placed first-fit so that two regions alive at the same time never overlap.
    std::vector<long long> poff(ntree + 1, 0), soff(ntree, 0);
    {
        std::vector<int> level_of(ntree, 0);
        for (int L = 0; L < nlevels; ++L)
            for (int i = level_off[L]; i < level_off[L + 1]; ++i) level_of[level_nodes[i]] = L;
        for (int t = 0; t < ntree; ++t) poff[t + 1] = poff[t] + (long long)ldp_of(nf[t]) * npiv[t];   // even leading dimension: 16-byte loads of row pairs
        std::vector<long long> rsize(nlevels + 1, 0);                 // region nlevels: fronts without a parent (never read)
        std::vector<int> rstart(nlevels + 1, nlevels);
        auto region_of = [&](int t) { return parent[t] >= 0 ? level_of[parent[t]] : nlevels; };
        for (int t = 0; t < ntree; ++t) {
            const long long nb = nf[t] - npiv[t];
            const int R = region_of(t);
            if (parent[t] >= 0 && level_of[parent[t]] <= level_of[t]) return fail(c, "inconsistent frontal plan (a parent is not above its child)");
            soff[t] = rsize[R];
            rsize[R] += nb * nb;
            rstart[R] = std::min(rstart[R], level_of[t]);
        }
        std::vector<long long> rbase(nlevels + 1, 0);
        std::vector<int> order;
        for (int R = 0; R <= nlevels; ++R)
            if (rsize[R] > 0) order.push_back(R);
        std::sort(order.begin(), order.end(), [&](int a, int b) { return rstart[a] != rstart[b] ? rstart[a] < rstart[b] : a < b; });
        std::vector<int> placed;
        long long arena = 0;
        for (int R : order) {
            // lifetimes: [rstart[R], R]; busy address intervals of the placed regions whose lifetime overlaps this one
            std::vector<std::pair<long long, long long>> busy;
            for (int Q : placed)
                if (rstart[Q] <= R && rstart[R] <= Q) busy.push_back({rbase[Q], rbase[Q] + rsize[Q]});
            std::sort(busy.begin(), busy.end());
            long long at = 0;
            for (auto& iv : busy) {
                if (at + rsize[R] <= iv.first) break;
                at = std::max(at, iv.second);
            }
            rbase[R] = at;
            arena = std::max(arena, at + rsize[R]);
            placed.push_back(R);
        }
        for (int t = 0; t < ntree; ++t) soff[t] += rbase[region_of(t)];
        fr.p_doubles = poff[ntree];
        fr.s_doubles = arena;
        fr.h_soff = soff;
    }
    fr.linv_doubles = linvoff[ntree];
#define UPI(dst, src, n) do { HIPCHK(c, hipMalloc((void**)&dst, std::max<size_t>((size_t)(n), 1) * sizeof(*dst))); \
        HIPCHK(c, hipMemcpy(dst, src, (size_t)(n) * sizeof(*dst), hipMemcpyHostToDevice)); } while (0)
    UPI(fr.nf, nf, ntree); UPI(fr.npiv, npiv, ntree); UPI(fr.parent, parent, ntree); UPI(fr.left, left, ntree);
    UPI(fr.right, right, ntree); UPI(fr.level_nodes, level_nodes, ntree); UPI(fr.dofs, front_dofs, ndofs_total);
    UPI(fr.upmap, up_map, ndofs_total); UPI(fr.elem_front, elem_front, c->nel); UPI(fr.elem_map, elem_map, (size_t)c->nel * c->ld);
    HIPCHK(c, hipMalloc((void**)&fr.poff, (ntree + 1) * sizeof(long long)));
    HIPCHK(c, hipMemcpy(fr.poff, poff.data(), (ntree + 1) * sizeof(long long), hipMemcpyHostToDevice));
    HIPCHK(c, hipMalloc((void**)&fr.soff, std::max(ntree, 1) * sizeof(long long)));
    HIPCHK(c, hipMemcpy(fr.soff, soff.data(), ntree * sizeof(long long), hipMemcpyHostToDevice));
    HIPCHK(c, hipMalloc((void**)&fr.doff, (ntree + 1) * sizeof(long long)));
    HIPCHK(c, hipMemcpy(fr.doff, dof_off, (ntree + 1) * sizeof(long long), hipMemcpyHostToDevice));
    HIPCHK(c, hipMalloc((void**)&fr.linvoff, (ntree + 1) * sizeof(long long)));
    HIPCHK(c, hipMemcpy(fr.linvoff, linvoff.data(), (ntree + 1) * sizeof(long long), hipMemcpyHostToDevice));
    {
        // X = L11^-1 of every front of the wide levels (leading dimension ldx_of(npiv)); Xtmp: scratch of the same shape
        std::vector<long long> xoff(ntree + 1, 0);
        std::vector<char> wide(ntree, 0);
        for (int L = 0; L < nlevels; ++L)
            if (fr.h_level_wide[L])
                for (int i = level_off[L]; i < level_off[L + 1]; ++i) wide[level_nodes[i]] = 1;
        for (int t = 0; t < ntree; ++t) {
            const long long ld = wide[t] ? ldx_of(npiv[t]) : 0;
            xoff[t + 1] = xoff[t] + ld * ld;
        }
        fr.x_doubles = xoff[ntree];
        HIPCHK(c, hipMalloc((void**)&fr.xoff, (ntree + 1) * sizeof(long long)));
        HIPCHK(c, hipMemcpy(fr.xoff, xoff.data(), (ntree + 1) * sizeof(long long), hipMemcpyHostToDevice));
        HIPCHK(c, hipMalloc((void**)&fr.X, std::max<size_t>((size_t)fr.x_doubles, 1) * sizeof(double)));
        HIPCHK(c, hipMalloc((void**)&fr.Xtmp, std::max<size_t>((size_t)fr.x_doubles, 1) * sizeof(double)));
        // scratch for the diagonal-block inverses of the other levels (needed only between k_diag_block and k_panel_rows)
        int max_cnt = 0;
        for (int L = 0; L < nlevels; ++L)
            if (!fr.h_level_wide[L]) max_cnt = std::max(max_cnt, level_off[L + 1] - level_off[L]);
        fr.swork_slots = std::max(1, std::min(max_cnt, std::max(1, c->opt.swork_slots)));
        HIPCHK(c, hipMalloc((void**)&fr.Swork, (size_t)fr.swork_slots * SPD * SPD * sizeof(double)));
    }
    {
        // row maps of the extend-add gather: for every row of a front, the row of each child's front that lands there
        std::vector<int> inv0((size_t)ndofs_total, -1), inv1((size_t)ndofs_total, -1);
        for (int t = 0; t < ntree; ++t) {
            const int p = parent[t];
            if (p < 0) continue;
            std::vector<int>& inv = left[p] == t ? inv0 : inv1;
            if (left[p] != t && right[p] != t) return fail(c, "inconsistent frontal plan (parent / child links)");
            for (int k = npiv[t]; k < nf[t]; ++k) {
                const int pr = up_map[dof_off[t] + k];
                if (pr < 0 || pr >= nf[p]) return fail(c, "inconsistent frontal plan (up_map out of range)");
                inv[(size_t)dof_off[p] + pr] = k;
            }
        }
        UPI(fr.cinv0, inv0.data(), ndofs_total); UPI(fr.cinv1, inv1.data(), ndofs_total);
    }
    {
        // task tables of the fused wide sweeps (k_sweep_wide_fwd / _bwd): per wide level the first-phase tiles of its fronts (in level
        // order), then the second-phase tiles; forward table by ascending level, backward table by descending level
        std::vector<int> slot_of(ntree, 0);
        for (int i = 0; i < ntree; ++i) slot_of[level_nodes[i]] = i;
        std::vector<SweepTask> ft, bt;
        fr.h_ft_off.assign(nlevels + 1, 0); fr.h_bt_begin.assign(nlevels, 0); fr.h_bt_end.assign(nlevels, 0);
        for (int L = 0; L < nlevels; ++L) {
            fr.h_ft_off[L] = (long long)ft.size();
            // (first-phase tiles of ALL fronts of the level, then the second-phase tiles: with a front's two phases next to each other
            //  the workgroups in flight are the waiting second-phase tiles of a few fronts while the other fronts' first phase queues)
            if (fr.h_level_wide[L]) {
                for (int i = level_off[L]; i < level_off[L + 1]; ++i)
                    for (int k = 0; k < sweep_xtiles(npiv[level_nodes[i]]); ++k) ft.push_back({i, k});
                for (int i = level_off[L]; i < level_off[L + 1]; ++i) {
                    const int t = level_nodes[i];
                    for (int k = 0; k < sweep_ltiles(npiv[t], nf[t] - npiv[t]); ++k) ft.push_back({i, (1 << 24) | k});
                }
            }
        }
        fr.h_ft_off[nlevels] = (long long)ft.size();
        for (int L = nlevels - 1; L >= 0; --L) {
            fr.h_bt_begin[L] = (long long)bt.size();
            if (fr.h_level_wide[L]) {
                for (int i = level_off[L]; i < level_off[L + 1]; ++i) {
                    const int t = level_nodes[i];
                    for (int k = 0; k < sweep_bblocks(npiv[t], nf[t] - npiv[t]); ++k) bt.push_back({i, k});
                }
                for (int i = level_off[L]; i < level_off[L + 1]; ++i)
                    for (int k = 0; k < sweep_xtiles(npiv[level_nodes[i]]); ++k) bt.push_back({i, (1 << 24) | k});
            }
            fr.h_bt_end[L] = (long long)bt.size();
        }
        if (sweep_xtiles(fr.max_nf) >= (1 << 24)) return fail(c, "front too large for the fused sweep task table");
        UPI(fr.slot_of, slot_of.data(), ntree);
        // front-centric assembly: the elements of every level-0 front
        const int cnt0 = level_off[1] - level_off[0];
        std::vector<int> fel_off(cnt0 + 1, 0), fel((size_t)std::max<long long>(c->nel, 1));
        fr.fc_ok = true;
        for (long long e = 0; e < c->nel && fr.fc_ok; ++e) {
            const int sl = slot_of[elem_front[e]] - level_off[0];
            if (sl < 0 || sl >= cnt0) fr.fc_ok = false; else ++fel_off[sl + 1];
        }
        if (fr.fc_ok) {
            for (int i = 0; i < cnt0; ++i) fel_off[i + 1] += fel_off[i];
            std::vector<int> fill(fel_off.begin(), fel_off.end() - 1);
            for (long long e = 0; e < c->nel; ++e) fel[fill[slot_of[elem_front[e]] - level_off[0]]++] = (int)e;
            UPI(fr.fel_off, fel_off.data(), cnt0 + 1); UPI(fr.fel, fel.data(), c->nel);
        }
        UPI(fr.ftasks, ft.data(), ft.size()); UPI(fr.btasks, bt.data(), bt.size());
        HIPCHK(c, hipMalloc((void**)&fr.sweep_cnt, (size_t)std::max(ntree, 1) * 2 * sizeof(int)));
    }
    HIPCHK(c, hipMalloc((void**)&fr.P, (size_t)std::max<long long>(fr.p_doubles, 1) * sizeof(double)));
    HIPCHK(c, hipMemset(fr.P, 0, (size_t)std::max<long long>(fr.p_doubles, 1) * sizeof(double)));   // once: the upper triangles of L11 are never written
    // two doubles of padding: the gathering updates read Sc[0] of a child without a Schur block (or of the front itself when a
    // child is missing) as their "safe address", which may be the very end of the arena
    HIPCHK(c, hipMalloc((void**)&fr.S, (size_t)(std::max<long long>(fr.s_doubles, 1) + 2) * sizeof(double)));
    HIPCHK(c, hipMalloc((void**)&fr.Linv, (size_t)std::max<long long>(fr.linv_doubles, 1) * sizeof(double)));
    HIPCHK(c, hipMalloc((void**)&fr.info, sizeof(int)));
    // dynamic LDS of the one-workgroup-per-front sweeps: forward maxnp + SMALL_PART, backward maxnp + maxnb + SMALL_PART
    // doubles, where the two maxima of a level may come from different fronts
    int max_sweep = fr.max_nf + SMALL_PART;
    for (int L = 0; L < nlevels; ++L) max_sweep = std::max(max_sweep, fr.h_level_maxnp[L] + fr.h_level_maxnb[L] + SMALL_PART);
    if ((size_t)max_sweep * sizeof(double) > 150 * 1024) return fail(c, "largest front does not fit the LDS solve kernels");
    if ((size_t)max_sweep * sizeof(double) > 48 * 1024) {
        const int bytes = (int)(max_sweep * sizeof(double));
        HIPCHK(c, hipFuncSetAttribute((const void*)k_front_fwd_small, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        HIPCHK(c, hipFuncSetAttribute((const void*)k_front_bwd_small<false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        HIPCHK(c, hipFuncSetAttribute((const void*)k_front_bwd_small<true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        HIPCHK(c, hipFuncSetAttribute((const void*)k_sweep_bnd_cols<false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        HIPCHK(c, hipFuncSetAttribute((const void*)k_sweep_bnd_cols<true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        HIPCHK(c, hipFuncSetAttribute((const void*)k_sweep_bwd_w<false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        HIPCHK(c, hipFuncSetAttribute((const void*)k_sweep_bwd_w<true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        HIPCHK(c, hipFuncSetAttribute((const void*)k_sweep_wide_bwd<0>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        HIPCHK(c, hipFuncSetAttribute((const void*)k_sweep_wide_bwd<1>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        HIPCHK(c, hipFuncSetAttribute((const void*)k_sweep_wide_bwd<2>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    }
    HIPCHK(c, hipFuncSetAttribute((const void*)k_diag_block, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(diag_block_lds_blocks(NBO / NB) * sizeof(blk32))));
    HIPCHK(c, hipFuncSetAttribute((const void*)k_panel_rows_preload, hipFuncAttributeMaxDynamicSharedMemorySize, PANEL_ROWS_PRELOAD_LDS));
    HIPCHK(c, hipFuncSetAttribute((const void*)k_trailing_big<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * sizeof(double) * 16 * LSTRB)));
    HIPCHK(c, hipFuncSetAttribute((const void*)k_trailing_big<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * sizeof(double) * 16 * LSTRB)));
    HIPCHK(c, hipFuncSetAttribute((const void*)k_diag_block2<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(diag_block2_lds_blocks(NBO / NB) * sizeof(blk32))));
    HIPCHK(c, hipFuncSetAttribute((const void*)k_diag_block2<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(diag_block2_lds_blocks(NBO / NB) * sizeof(blk32))));
    fr.ready = true;
    if (fr.sweep_graph) { hipGraphExecDestroy(fr.sweep_graph); fr.sweep_graph = nullptr; }     // captured for another plan
    fr.factored = false;
    return 0;
}

int femo_factorize(femo_ctx* c) {
    HIPCHK(c, hipSetDevice(c->device));
    return frontal_factorize(c);
}

/* out6: [0] element-matrix assembly into fronts (ms), [1] numeric factorisation (ms), [2] front storage (GB),
 *       [3] factor flops (GFLOP, from the plan sizes), [4] number of non-positive pivots repaired, [5] fronts */
/* Run one factorisation with a HIP event pair around every kernel launch and report, per kernel class
 * (0 panel rows, 1 diagonal blocks, 2 trailing, 3 extend_add, 4 front_assemble, 5 memset): total ms and launches;
 * also the algorithmic flop counts (lower triangles only) of what the launches of classes 2, 0, 1 execute.
 * out32 = ms[8], calls[8], flops[8], bytes[8], indexed by class (class 6: inversion of L11; class 7: the subset of class 2 whose
 * flops / compulsory bytes lie above the chip's ridge of 9.8 flop per byte, i.e. the launches the matrix cores can bound). */
int femo_factorize_profile(femo_ctx* c, double* out32) {
    HIPCHK(c, hipSetDevice(c->device));
    c->fr.profile = true;
    int rc = frontal_factorize(c);
    c->fr.profile = false;
    if (rc) return rc;
    for (int i = 0; i < 8; ++i) {
        out32[i] = c->fr.prof_ms[i]; out32[8 + i] = (double)c->fr.prof_calls[i];
        out32[16 + i] = c->fr.prof_flops[i]; out32[24 + i] = c->fr.prof_bytes[i];
    }
    return 0;
}

// one application of the factor with HIP events between the launches of both sweeps:
// out[4 L + 0 / 1] = forward sweep of level L, first / second launch (X b, then L21 y; levels with one workgroup per front
// have only the first), out[4 L + 2 / 3] = backward sweep (L21^T x, then X^T s).  ms; n >= 4 nlevels
int femo_sweep_profile(femo_ctx* c, double* out, int64_t n) {
    HIPCHK(c, hipSetDevice(c->device));
    auto& fr = c->fr;
    if (!fr.ready) return fail(c, "no frontal plan");
    if (n < 4 * (int64_t)fr.nlevels) return fail(c, "output too small: 4 * nlevels doubles");
    if (!fr.factored)
        if (int rc = frontal_factorize(c)) return rc;
    hipLaunchKernelGGL(k_fill, dim3(vec_grid(c->ndof)), dim3(256), 0, c->stream, c->z, 1.0, (int64_t)c->ndof);
    std::vector<hipEvent_t> mf, mb;
    int rc = frontal_fwd(c, c->z, 0, fr.nlevels, &mf);
    if (!rc) rc = frontal_bwd(c, c->z, 0, fr.nlevels, &mb);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // marks: one before the first level, then two per level (after the first launch, after the level)
    if (!rc && (int)mf.size() == 1 + 2 * fr.nlevels && (int)mb.size() == 1 + 2 * fr.nlevels) {
        for (int L = 0; L < fr.nlevels; ++L) {
            float a = 0, b = 0;
            hipEventElapsedTime(&a, mf[2 * L], mf[2 * L + 1]); hipEventElapsedTime(&b, mf[2 * L + 1], mf[2 * L + 2]);
            out[4 * L] = a; out[4 * L + 1] = b;
            const int k = fr.nlevels - 1 - L;               // the backward sweep visits the levels in reverse
            hipEventElapsedTime(&a, mb[2 * k], mb[2 * k + 1]); hipEventElapsedTime(&b, mb[2 * k + 1], mb[2 * k + 2]);
            out[4 * L + 2] = a; out[4 * L + 3] = b;
        }
    } else if (!rc) {
        rc = fail(c, "internal: unexpected number of sweep marks");
    }
    for (auto e : mf) hipEventDestroy(e);
    for (auto e : mb) hipEventDestroy(e);
    return rc;
}

// the same for the sweeps with nrhs = 2 or 4 interleaved vectors (sweeps_multi.h): out[2 L] = forward sweep of level L, out[2 L + 1] =
// backward sweep (ms); n >= 2 nlevels
int femo_sweep_profile_multi(femo_ctx* c, int32_t nrhs, double* out, int64_t n) {
    HIPCHK(c, hipSetDevice(c->device));
    auto& fr = c->fr;
    if (!fr.ready) return fail(c, "no frontal plan");
    if (nrhs != 2 && nrhs != 4) return fail(c, "nrhs: 2 or 4");
    if (n < 2 * (int64_t)fr.nlevels) return fail(c, "output too small: 2 * nlevels doubles");
    if (!fr.factored)
        if (int rc = frontal_factorize(c)) return rc;
    if (mr_alloc(c)) return 1;
    hipLaunchKernelGGL(k_fill, dim3(vec_grid((int64_t)c->ndof * nrhs)), dim3(256), 0, c->stream, c->mr_v, 1.0, (int64_t)c->ndof * nrhs);
    if (join_xinv(c)) return 1;
    std::vector<hipEvent_t> mk;
    int rc = nrhs == 2 ? frontal_solve_multi<2>(c, c->mr_v, c->mr_y, &mk) : frontal_solve_multi<4>(c, c->mr_v, c->mr_y, &mk);
    if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(c, "synchronisation failed");
    if (!rc && (int)mk.size() == 2 * fr.nlevels + 2) {
        for (int L = 0; L < fr.nlevels; ++L) {
            float a = 0, b = 0;
            hipEventElapsedTime(&a, mk[L], mk[L + 1]);
            const int k = fr.nlevels + 1 + (fr.nlevels - 1 - L);          // the backward sweep visits the levels in reverse
            hipEventElapsedTime(&b, mk[k], mk[k + 1]);
            out[2 * L] = a; out[2 * L + 1] = b;
        }
    } else if (!rc) {
        rc = fail(c, "internal: unexpected number of sweep marks");
    }
    for (auto e : mk) hipEventDestroy(e);
    return rc;
}

int femo_frontal_info(const femo_ctx* c, double* out6) {
    const auto& fr = c->fr;
    double fl = 0;
    for (int t = 0; t < fr.ntree; ++t) {
        const double p = fr.h_npiv[t], n = fr.h_nf[t];
        fl += p * n * n - p * p * n + p * p * p / 3.0;
    }
    out6[0] = fr.t_assemble_ms; out6[1] = fr.t_factor_ms; out6[2] = (fr.p_doubles + fr.s_doubles) * 8.0 / 1e9; out6[3] = fl / 1e9;
    out6[4] = fr.pivots_fixed; out6[5] = fr.ntree;
    return 0;
}


// ---- building blocks for the element-partitioned (multi-GPU) driver, femo_alpha_amd/parallel.py -------------
static double* vec_by_id(femo_ctx* c, int id) {
    double* v[] = {c->w, c->lam, c->r, c->z, c->p, c->Ap, c->b};
    return (id >= 0 && id < 7) ? v[id] : nullptr;
}

void* femo_vec_ptr(femo_ctx* c, int32_t id) { return vec_by_id(c, id); }

void* femo_stream_ptr(femo_ctx* c) { return (void*)c->stream; }

int femo_sync(femo_ctx* c) {
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int femo_op_apply_vec(femo_ctx* c, int32_t src, int32_t dst) {
    HIPCHK(c, hipSetDevice(c->device));
    double *x = vec_by_id(c, src), *y = vec_by_id(c, dst);
    if (!x || !y || x == y) return fail(c, "bad vector ids");
    hipLaunchKernelGGL(k_fill, dim3(vec_grid(c->ndof)), dim3(256), 0, c->stream, y, 0.0, (int64_t)c->ndof);
    if (op_apply(c, x, y, nullptr, nullptr, nullptr, true)) return 1;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int femo_load_vec(femo_ctx* c, int32_t dst) {
    HIPCHK(c, hipSetDevice(c->device));
    double* y = vec_by_id(c, dst);
    if (!y) return fail(c, "bad vector id");
    if (load_vector_dev(c, y)) return 1;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int femo_factorize_range(femo_ctx* c, int32_t l0, int32_t l1, int assemble) {
    HIPCHK(c, hipSetDevice(c->device));
    return frontal_factorize_range(c, l0, l1, assemble != 0);
}

int femo_frontal_sweep(femo_ctx* c, int32_t vec, int32_t l0, int32_t l1, int backward) {
    HIPCHK(c, hipSetDevice(c->device));
    double* v = vec_by_id(c, vec);
    if (!v) return fail(c, "bad vector id");
    if (l0 < 0 || l1 > c->fr.nlevels || l0 > l1) return fail(c, "bad level range");
    if (int rc = backward ? frontal_bwd(c, v, l0, l1) : frontal_fwd(c, v, l0, l1)) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// contiguous copy of the Schur complement (trailing nb x nb block, column-major; the lower triangle is what is maintained)
int femo_front_schur_get(femo_ctx* c, int32_t front, void* dst_dev, int64_t capacity_doubles) {
    HIPCHK(c, hipSetDevice(c->device));
    auto& fr = c->fr;
    if (!fr.ready || front < 0 || front >= fr.ntree) return fail(c, "bad front id");
    const int nf = fr.h_nf[front], np = fr.h_npiv[front], nb = nf - np;
    if ((int64_t)nb * nb > capacity_doubles) return fail(c, "destination too small for the Schur complement");
    if (nb == 0) return 0;
    HIPCHK(c, hipMemcpyAsync(dst_dev, fr.S + fr.h_soff[front], (size_t)nb * nb * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// overwrite the whole dense block of a front without pivots (a remote subtree's Schur complement)
int femo_front_block_set(femo_ctx* c, int32_t front, const void* src_dev) {
    HIPCHK(c, hipSetDevice(c->device));
    auto& fr = c->fr;
    if (!fr.ready || front < 0 || front >= fr.ntree) return fail(c, "bad front id");
    if (fr.h_npiv[front] != 0) return fail(c, "only fronts without pivots can be overwritten");
    const int nf = fr.h_nf[front];
    HIPCHK(c, hipMemcpyAsync(fr.S + fr.h_soff[front], src_dev, (size_t)nf * nf * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// local partial sums for the stored state: out3 = { int u.u J dx, regularisation, mass } over this context's cells
int femo_functionals_partial(femo_ctx* c, double* out3) {
    HIPCHK(c, hipSetDevice(c->device));
    return functionals_dev(c, out3);
}

// gradient pieces on device vectors (no host copies): dst(vector id) = d functional / d disp_solid
int femo_dfunctional_vec(femo_ctx* c, const char* name, int32_t dst) {
    HIPCHK(c, hipSetDevice(c->device));
    double* y = vec_by_id(c, dst);
    if (!y) return fail(c, "bad vector id");
    if (int rc = dfunctional_dev(c, name ? name : "", "disp_solid", y, c->ndof)) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// out (host, field length) = scale * (dR/d arg)^T lambda(vector id) + d functional / d arg, local cells only
int femo_field_gradient_vec(femo_ctx* c, const char* functional, const char* arg, int32_t lam, double* out, int64_t n) {
    HIPCHK(c, hipSetDevice(c->device));
    double* l = vec_by_id(c, lam);
    if (!l) return fail(c, "bad vector id");
    double* d = nullptr;
    HIPCHK(c, hipMalloc((void**)&d, std::max<int64_t>(n, 1) * sizeof(double)));
    int rc = dfunctional_dev(c, functional ? functional : "", arg ? arg : "", d, n);
    if (!rc) rc = dRdarg_T_dev(c, arg ? arg : "", l, -1.0, d, n);
    if (!rc) {
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) e = hipMemcpy(out, d, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { c->err = hipGetErrorString(e); rc = 1; }
    }
    hipFree(d);
    return rc;
}



// ---- element-partitioned PCG: everything between two collectives is ONE call that only enqueues work on the context's
// stream (no host synchronisation); the caller issues the all-reduce on femo_dist_ptr(ctx, 0) / (ctx, 1) on the same stream.
// Per iteration: precond_fwd -> all-reduce(topbuf[0 .. ntop]) -> read -> precond_rest -> all-reduce(scal[1]) ->
// direction_apply -> all-reduce(topbuf[0 .. ntop]) -> update.  topbuf[ntop] carries the scalar that rides along with the
// replicated entries: this rank's share of r.r (first collective) or of p.Ap (second).
int femo_dist_setup(femo_ctx* c, int32_t ntop, const int32_t* top_idx, int32_t nranks, int32_t n_local_levels, int32_t nsel,
                    const int32_t* sel) {
    HIPCHK(c, hipSetDevice(c->device));
    auto& d = c->di;
    if (d.ready) return fail(c, "femo_dist_setup was already called for this context");
    if (ntop < 0 || nranks < 1 || nsel < 0 || (ntop > 0 && !top_idx) || (nsel > 0 && !sel)) return fail(c, "bad arguments");
    if (!c->fr.ready || n_local_levels < 0 || n_local_levels > c->fr.nlevels) return fail(c, "n_local_levels outside the frontal plan");
    for (int i = 0; i < ntop; ++i)
        if (top_idx[i] < 0 || top_idx[i] >= c->ndof) return fail(c, "replicated entry outside the vector");
    d.ntop = ntop; d.nranks = nranks; d.nl = n_local_levels; d.nsel = nsel;
    std::vector<double> w((size_t)c->ndof, 1.0);
    for (int i = 0; i < ntop; ++i) w[top_idx[i]] = 1.0 / nranks;
    HIPCHK(c, hipMalloc((void**)&d.wdot, (size_t)c->ndof * sizeof(double)));
    HIPCHK(c, hipMemcpy(d.wdot, w.data(), (size_t)c->ndof * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(c, hipMalloc((void**)&d.top_idx, (size_t)std::max(ntop, 1) * sizeof(int)));
    if (ntop) HIPCHK(c, hipMemcpy(d.top_idx, top_idx, (size_t)ntop * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(c, hipMalloc((void**)&d.topbuf, (size_t)(ntop + 2) * sizeof(double)));
    HIPCHK(c, hipMemset(d.topbuf, 0, (size_t)(ntop + 2) * sizeof(double)));
    HIPCHK(c, hipMalloc((void**)&d.topsave, (size_t)std::max(ntop, 1) * sizeof(double)));
    HIPCHK(c, hipMalloc((void**)&d.sel, (size_t)std::max(nsel, 1) * sizeof(int)));
    if (nsel) HIPCHK(c, hipMemcpy(d.sel, sel, (size_t)nsel * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(c, hipMalloc((void**)&d.gloc, (size_t)std::max<int64_t>(std::max<int64_t>(c->nT, 3 * c->nF), 1) * sizeof(double)));
    d.ready = true;
    return 0;
}

void* femo_dist_ptr(femo_ctx* c, int32_t which) { return which == 0 ? (void*)c->di.topbuf : which == 1 ? (void*)c->scal : nullptr; }

#define DIST_READY(c) do { HIPCHK(c, hipSetDevice((c)->device)); if (!(c)->di.ready) return fail(c, "call femo_dist_setup first"); } while (0)

int femo_dist_pack(femo_ctx* c, int32_t vec) {
    DIST_READY(c);
    double* v = vec_by_id(c, vec);
    if (!v) return fail(c, "bad vector id");
    if (c->di.ntop) hipLaunchKernelGGL(k_gather_idx, dim3(nblk(c->di.ntop, 256)), dim3(256), 0, c->stream, c->di.topbuf, (const double*)v, (const int*)c->di.top_idx, c->di.ntop);
    HIPCHK(c, hipGetLastError());
    return 0;
}

int femo_dist_unpack(femo_ctx* c, int32_t vec) {
    DIST_READY(c);
    double* v = vec_by_id(c, vec);
    if (!v) return fail(c, "bad vector id");
    if (c->di.ntop) hipLaunchKernelGGL(k_scatter_idx, dim3(nblk(c->di.ntop, 256)), dim3(256), 0, c->stream, v, (const double*)c->di.topbuf, (const int*)c->di.top_idx, c->di.ntop);
    HIPCHK(c, hipGetLastError());
    return 0;
}

// x = 0, r = b; this rank's share of b.b goes into scal[3] (it travels with the first collective of the first iteration)
int femo_dist_pcg_start(femo_ctx* c, int32_t b, int32_t x) {
    DIST_READY(c);
    double *vb = vec_by_id(c, b), *vx = vec_by_id(c, x);
    if (!vb || !vx || vb == vx || vb == c->r || vb == c->z || vb == c->p || vb == c->Ap || vx == c->r || vx == c->z || vx == c->p || vx == c->Ap)
        return fail(c, "bad vector ids (2..5 are the solver's work space)");
    const int64_t n = c->ndof;
    HIPCHK(c, hipMemsetAsync(c->scal, 0, 8 * sizeof(double), c->stream));
    HIPCHK(c, hipMemsetAsync(vx, 0, n * sizeof(double), c->stream));
    HIPCHK(c, hipMemcpyAsync(c->r, vb, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    hipLaunchKernelGGL(k_wdot, dim3(red_grid(n)), dim3(256), 0, c->stream, (const double*)vb, (const double*)vb, (const double*)c->di.wdot, n, c->scal + 3);
    HIPCHK(c, hipGetLastError());
    return 0;
}

// z = r; forward sweep over this rank's subtree; topbuf = what it added to the replicated entries, topbuf[ntop] = share of r.r
int femo_dist_precond_fwd(femo_ctx* c) {
    DIST_READY(c);
    auto& d = c->di;
    if (!c->fr.factored) return fail(c, "the factorisation is stale");
    const int64_t n = c->ndof;
    HIPCHK(c, hipMemcpyAsync(c->z, c->r, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    if (d.ntop) hipLaunchKernelGGL(k_gather_idx, dim3(nblk(d.ntop, 256)), dim3(256), 0, c->stream, d.topsave, (const double*)c->z, (const int*)d.top_idx, d.ntop);
    if (int rc = frontal_fwd(c, c->z, 0, d.nl)) return rc;
    if (d.ntop) hipLaunchKernelGGL(k_top_delta, dim3(nblk(d.ntop, 256)), dim3(256), 0, c->stream, d.topbuf, (const double*)c->z, (const double*)d.topsave, (const int*)d.top_idx, d.ntop);
    hipLaunchKernelGGL(k_copy_scalar, dim3(1), dim3(64), 0, c->stream, d.topbuf + d.ntop, (const double*)(c->scal + 3));
    HIPCHK(c, hipGetLastError());
    return 0;
}

// after the collective: out2 = { global r.r, global p.Ap of the previous iteration }  (the one host synchronisation per iteration)
int femo_dist_read(femo_ctx* c, double* out2) {
    DIST_READY(c);
    HIPCHK(c, hipMemcpyAsync(c->scal_host, c->di.topbuf + c->di.ntop, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->scal_host + 1, c->scal + 2, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    out2[0] = c->scal_host[0]; out2[1] = c->scal_host[1];
    return 0;
}

// replicated entries of z <- saved + summed delta; the replicated top of the tree forward and backward; this rank's subtree
// backward; scal[1] = share of r.z
int femo_dist_precond_rest(femo_ctx* c) {
    DIST_READY(c);
    auto& d = c->di;
    const int64_t n = c->ndof;
    if (d.ntop) hipLaunchKernelGGL(k_top_restore, dim3(nblk(d.ntop, 256)), dim3(256), 0, c->stream, c->z, (const double*)d.topsave, (const double*)d.topbuf, (const int*)d.top_idx, d.ntop);
    if (int rc = frontal_fwd(c, c->z, d.nl, c->fr.nlevels)) return rc;
    if (int rc = frontal_bwd(c, c->z, d.nl, c->fr.nlevels)) return rc;
    if (int rc = frontal_bwd(c, c->z, 0, d.nl)) return rc;
    HIPCHK(c, hipMemsetAsync(c->scal + 1, 0, 3 * sizeof(double), c->stream));
    hipLaunchKernelGGL(k_wdot, dim3(red_grid(n)), dim3(256), 0, c->stream, (const double*)c->r, (const double*)c->z, (const double*)d.wdot, n, c->scal + 1);
    HIPCHK(c, hipGetLastError());
    return 0;
}

// p = z + (r.z / previous r.z) p; Ap = (local operator) p; topbuf = Ap on the replicated entries, topbuf[ntop] = p . Ap_local
// (p^T A p is the sum over the ranks of p^T A_local p: the scalar needs no collective of its own)
int femo_dist_direction_apply(femo_ctx* c, int first) {
    DIST_READY(c);
    auto& d = c->di;
    const int64_t n = c->ndof;
    const int vg = vec_grid(n);
    hipLaunchKernelGGL(k_pcgf_direction, dim3(vg), dim3(256), 0, c->stream, c->p, (const double*)c->z, (const double*)c->scal, first ? 1 : 0, n);
    HIPCHK(c, hipMemsetAsync(c->Ap, 0, n * sizeof(double), c->stream));
    if (op_apply(c, c->p, c->Ap, nullptr, nullptr, nullptr, true, c->op_aK, c->op_aM)) return 1;
    if (c->has_mask) hipLaunchKernelGGL(k_mask_zero, dim3(vg), dim3(256), 0, c->stream, c->Ap, c->mask, n);
    hipLaunchKernelGGL(k_dot, dim3(red_grid(n)), dim3(256), 0, c->stream, (const double*)c->p, (const double*)c->Ap, n, c->scal + 2);
    if (d.ntop) hipLaunchKernelGGL(k_gather_idx, dim3(nblk(d.ntop, 256)), dim3(256), 0, c->stream, d.topbuf, (const double*)c->Ap, (const int*)d.top_idx, d.ntop);
    hipLaunchKernelGGL(k_copy_scalar, dim3(1), dim3(64), 0, c->stream, d.topbuf + d.ntop, (const double*)(c->scal + 2));
    HIPCHK(c, hipGetLastError());
    return 0;
}

// after the collective: Ap's replicated entries and the global p.Ap are in topbuf; x += alpha p, r -= alpha Ap, scal[3] = share of r.r
int femo_dist_update(femo_ctx* c, int32_t x) {
    DIST_READY(c);
    auto& d = c->di;
    double* vx = vec_by_id(c, x);
    if (!vx) return fail(c, "bad vector id");
    const int64_t n = c->ndof;
    if (d.ntop) hipLaunchKernelGGL(k_scatter_idx, dim3(nblk(d.ntop, 256)), dim3(256), 0, c->stream, c->Ap, (const double*)d.topbuf, (const int*)d.top_idx, d.ntop);
    hipLaunchKernelGGL(k_copy_scalar, dim3(1), dim3(64), 0, c->stream, c->scal + 2, (const double*)(d.topbuf + d.ntop));
    hipLaunchKernelGGL(k_pcgf_update_w, dim3(red_grid(n)), dim3(256), 0, c->stream, vx, c->r, (const double*)c->p, (const double*)c->Ap, (const double*)d.wdot, c->scal, n);
    HIPCHK(c, hipGetLastError());
    return 0;
}

// gglob (device, nglob doubles, zeroed by the caller) [sel] = - (dR/d arg)^T lambda + d functional / d arg over this rank's cells
int femo_dist_gradient(femo_ctx* c, const char* functional, const char* arg, int32_t lam, void* gglob, int64_t nglob) {
    DIST_READY(c);
    auto& d = c->di;
    double* l = vec_by_id(c, lam);
    if (!l) return fail(c, "bad vector id");
    int64_t len;
    if (!field_ptr(c, arg ? arg : "", &len)) return fail(c, "unknown argument");
    if (len != d.nsel) return fail(c, "the gradient scatter map was set up for a field of another length");
    int rc = dfunctional_dev(c, functional ? functional : "", arg, d.gloc, len);
    if (!rc) rc = dRdarg_T_dev(c, arg, l, -1.0, d.gloc, len);
    if (rc) return rc;
    hipLaunchKernelGGL(k_scatter_idx, dim3(nblk(d.nsel, 256)), dim3(256), 0, c->stream, (double*)gglob, (const double*)d.gloc, (const int*)d.sel, d.nsel);
    HIPCHK(c, hipGetLastError());
    (void)nglob;
    return 0;
}

// packed lower triangle of a front's Schur complement (what the all-gather of the subtree roots carries), and back into a
// pivot-free stand-in front; both asynchronous on the context's stream
int femo_front_schur_pack(femo_ctx* c, int32_t front, void* dst_dev, int64_t capacity_doubles) {
    HIPCHK(c, hipSetDevice(c->device));
    auto& fr = c->fr;
    if (!fr.ready || front < 0 || front >= fr.ntree) return fail(c, "bad front id");
    const int nb = fr.h_nf[front] - fr.h_npiv[front];
    if ((int64_t)nb * (nb + 1) / 2 > capacity_doubles) return fail(c, "destination too small for the packed Schur complement");
    if (nb == 0) return 0;
    hipLaunchKernelGGL(k_tril_pack, dim3(std::min(8, nblk(nb, 256)), nb), dim3(256), 0, c->stream, (double*)dst_dev, (const double*)(fr.S + fr.h_soff[front]), nb);
    HIPCHK(c, hipGetLastError());
    return 0;
}

int femo_front_block_unpack(femo_ctx* c, int32_t front, const void* src_dev) {
    HIPCHK(c, hipSetDevice(c->device));
    auto& fr = c->fr;
    if (!fr.ready || front < 0 || front >= fr.ntree) return fail(c, "bad front id");
    if (fr.h_npiv[front] != 0) return fail(c, "only fronts without pivots can be overwritten");
    const int nf = fr.h_nf[front];
    if (nf == 0) return 0;
    hipLaunchKernelGGL(k_tril_unpack, dim3(std::min(8, nblk(nf, 256)), nf), dim3(256), 0, c->stream, fr.S + fr.h_soff[front], (const double*)src_dev, nf);
    HIPCHK(c, hipGetLastError());
    return 0;
}

// results of the instrumented factorisation(s) since the last assembly (option "profile"): same layout as femo_factorize_profile
int femo_factorize_profile_get(femo_ctx* c, double* out32) {
    for (int i = 0; i < 8; ++i) {
        out32[i] = c->fr.prof_ms[i]; out32[8 + i] = (double)c->fr.prof_calls[i];
        out32[16 + i] = c->fr.prof_flops[i]; out32[24 + i] = c->fr.prof_bytes[i];
    }
    return 0;
}

// ---- dynamic shell: operator A = aK K + aM M, device-vector building blocks (femo_alpha_amd/dynamic_rm_shell) ----
int femo_set_operator(femo_ctx* c, double aK, double aM) {
    if (aK != c->op_aK || aM != c->op_aM) { c->op_aK = aK; c->op_aM = aM; operator_changed(c); }
    return 0;
}

// The rule of the operator after creation (same meaning as femo_create's nquad): the host mirror raises the triangles' rule from
// degree 6 to 9 when a field makes the integrand non-polynomial (ShellContext.set_field); everything assembled or factorised with
// the old tables is stale
int femo_set_quadrature(femo_ctx* c, int32_t nquad) {
    HIPCHK(c, hipSetDevice(c->device));
    if (c->quad && (nquad < 2 || nquad > 6)) return fail(c, "nquad must be in 2..6");
    if (!c->quad && nquad != 4 && nquad != 6 && nquad != 9 && nquad != 12)
        return fail(c, "triangles: nquad is the degree of the symmetric rule -- 4, 6, 9 or 12");
    if (c->quad && c->nred * c->nred + nquad * nquad > MAXQ) return fail(c, "the rule does not fit beside the reduced strain rule");
    if (nquad == c->nquad) return 0;
    Tables T;
    build_tables(c->quad, nquad, T, c->nred, c->cg1, c->cr);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(c->tab, &T, sizeof(Tables), hipMemcpyHostToDevice));
    c->tab_nq = T.nq;
    c->nquad = nquad;
    operator_changed(c);
    return 0;
}

// Host only (no device needed): the tables build_tables makes for a rule, so that a test can hold them bit for bit against the
// oracle's (tests/test_cabi.py).  Arrays are sized for MAXQ = 36 points: w, wS [36]; N2 [36][9]; dN2 [36][9][2]; N1, NR [36][4]; dN1, dNR [36][4][2].
int femo_quadrature_tables(int32_t nvc, int32_t nquad, int32_t nred, int32_t cg1, int32_t cr, int32_t* npoints, double* w, double* wS,
                           double* N2, double* dN2, double* N1, double* dN1, double* NR, double* dNR) {
    if (nvc != 3 && nvc != 4) return 2;
    if (nvc == 4 && (nquad < 2 || nquad > 6 || nred < 0 || nred * nred + nquad * nquad > MAXQ)) return 2;
    if (nvc == 3 && (nred != 0 || (nquad != 4 && nquad != 6 && nquad != 9 && nquad != 12))) return 2;
    static Tables T;
    build_tables(nvc == 4, nquad, T, nred, cg1 != 0, cr != 0);
    if (npoints) *npoints = T.nq;
    if (w) memcpy(w, T.w, sizeof T.w);
    if (wS) memcpy(wS, T.wS, sizeof T.wS);
    if (N2) memcpy(N2, T.N2, sizeof T.N2);
    if (dN2) memcpy(dN2, T.dN2, sizeof T.dN2);
    if (N1) memcpy(N1, T.N1, sizeof T.N1);
    if (dN1) memcpy(dN1, T.dN1, sizeof T.dN1);
    if (NR) memcpy(NR, T.NR, sizeof T.NR);
    if (dNR) memcpy(dNR, T.dNR, sizeof T.dNR);
    return 0;
}

int femo_get_quadrature(femo_ctx* c, int32_t* nquad, int32_t* npoints) {
    if (nquad) *nquad = c->nquad;
    if (npoints) *npoints = c->tab_nq;
    return 0;
}

int femo_set_strain_quadrature(femo_ctx* c, int32_t nred) {
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->quad) return fail(c, "reduced strain quadrature is implemented for quadrilaterals only");
    if (nred < 0 || nred > 5 || nred * nred + c->nquad * c->nquad > MAXQ) return fail(c, "unsupported reduced rule");
    Tables T;
    build_tables(true, c->nquad, T, nred, c->cg1, c->cr);
    HIPCHK(c, hipMemcpy(c->tab, &T, sizeof(Tables), hipMemcpyHostToDevice));
    c->tab_nq = T.nq;
    c->nred = nred; operator_changed(c);
    return 0;
}

int femo_op_apply_vec2(femo_ctx* c, int32_t src, int32_t dst, double aK, double aM, int with_penalty) {
    HIPCHK(c, hipSetDevice(c->device));
    double *x = vec_by_id(c, src), *y = vec_by_id(c, dst);
    if (!x || !y || x == y) return fail(c, "bad vector ids");
    if (op_apply(c, x, y, nullptr, nullptr, nullptr, with_penalty != 0, aK, aM)) return 1;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int femo_solve_vec(femo_ctx* c, int32_t b, int32_t x, int zero_guess, int32_t* iters, double* relres) {
    HIPCHK(c, hipSetDevice(c->device));
    double *vb = vec_by_id(c, b), *vx = vec_by_id(c, x);
    if (!vb || !vx || vb == vx) return fail(c, "bad vector ids");
    if (vb == c->r || vb == c->z || vb == c->p || vb == c->Ap || vx == c->r || vx == c->z || vx == c->p || vx == c->Ap)
        return fail(c, "vectors 2..5 are the solver's work space");
    return solve_dispatch(c, vb, vx, zero_guess != 0, iters, relres);
}

int femo_vec_mask_zero(femo_ctx* c, int32_t id) {
    HIPCHK(c, hipSetDevice(c->device));
    double* v = vec_by_id(c, id);
    if (!v) return fail(c, "bad vector id");
    if (c->has_mask) hipLaunchKernelGGL(k_mask_zero, dim3(vec_grid(c->ndof)), dim3(256), 0, c->stream, v, c->mask, (int64_t)c->ndof);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// thickness-gradient accumulator on the device:  g += scale * y^T (dK/dh) x   or   g += scale * y^T (dM/dh) x
int femo_grad_reset(femo_ctx* c) {
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->gradbuf) HIPCHK(c, hipMalloc((void**)&c->gradbuf, (size_t)std::max<int64_t>(c->nT, 1) * sizeof(double)));
    HIPCHK(c, hipMemsetAsync(c->gradbuf, 0, (size_t)c->nT * sizeof(double), c->stream));
    return 0;
}

int femo_grad_add(femo_ctx* c, int kind, int32_t x, int32_t y, double scale) {
    HIPCHK(c, hipSetDevice(c->device));
    double *vx = vec_by_id(c, x), *vy = vec_by_id(c, y);
    if (!vx || !vy) return fail(c, "bad vector ids");
    if (!c->gradbuf) return fail(c, "call femo_grad_reset first");
    const MeshDev m = mesh_dev(c);
    const FieldsDev f = fields_dev(c);
    const int g = nblk(c->nel, EB);
    if (kind == 0) ELEM_LAUNCH(c, k_dRdfield_T, COMMA_H, g, EB, m, f, c->tab, vx, vy, scale, c->gradbuf);
    else if (kind == 1) ELEM_LAUNCH(c, k_dMdh_T, NOEXTRA, g, EB, m, f, c->tab, vx, vy, scale, c->gradbuf);
    else return fail(c, "kind must be 0 (stiffness) or 1 (inertia)");
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int femo_grad_get(femo_ctx* c, double* out, int64_t n) {
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->gradbuf || n != c->nT) return fail(c, "gradient accumulator not initialised or wrong length");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, c->gradbuf, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}



// ---- transient march in the library (BASELINE config 5): the reference's PlateSim.solve_dynamic_problem
// (dynamic_rm_shell/plate_sim.py:281-361: per step update_f, solveNonlinear_mod, wdot update) and the backward sweep of its
// adjoint (state_operation_dynamic.py:619-691), with history, force history and velocity resident in HBM.
//   w_mid = (w_old + w)/2,  wdot = 2/dt (w - w_old) - wdot_old,  wddot = (wdot - wdot_old)/dt      (plate_sim.py:131-140)
//   step:  (a M + K/2) w_i = F_i + M (a w_{i-1} + b wdot_{i-1}) - K/2 w_{i-1},   a = 2/dt^2, b = 2/dt
int femo_newmark_setup(femo_ctx* c, int32_t time_levels, double dt) {
    HIPCHK(c, hipSetDevice(c->device));
    auto& nm = c->nm;
    if (time_levels < 2 || !(dt > 0)) return fail(c, "femo_newmark_setup: need at least two time levels and dt > 0");
    void* old[] = {nm.W, nm.Fh, nm.wdot, nm.Fsw, nm.mu0, nm.mu1, nm.Lam, nm.Gh};
    for (void* p : old) if (p) hipFree(p);
    nm = femo_ctx::Newmark();
    const size_t n = (size_t)c->ndof;
    HIPCHK(c, hipMalloc((void**)&nm.W, (size_t)time_levels * n * sizeof(double)));
    HIPCHK(c, hipMemset(nm.W, 0, (size_t)time_levels * n * sizeof(double)));
    HIPCHK(c, hipMalloc((void**)&nm.wdot, n * sizeof(double)));
    HIPCHK(c, hipMalloc((void**)&nm.mu0, n * sizeof(double)));
    HIPCHK(c, hipMalloc((void**)&nm.mu1, n * sizeof(double)));
    nm.levels = time_levels; nm.dt = dt; nm.a = 2.0 / (dt * dt); nm.b = 2.0 / dt;
    if (0.5 != c->op_aK || nm.a != c->op_aM) { c->op_aK = 0.5; c->op_aM = nm.a; operator_changed(c); }
    nm.ready = true;
    return 0;
}

// pressure history, (levels_given x field length of F_solid) row-major; levels beyond the last one repeat it
int femo_newmark_set_forces(femo_ctx* c, const double* f_history, int32_t levels_given) {
    HIPCHK(c, hipSetDevice(c->device));
    auto& nm = c->nm;
    if (!nm.ready) return fail(c, "call femo_newmark_setup first");
    if (levels_given < 1 || !f_history) return fail(c, "empty force history");
    if (nm.Fh) { hipFree(nm.Fh); nm.Fh = nullptr; }
    const size_t bytes = (size_t)levels_given * 3 * c->nF * sizeof(double);
    HIPCHK(c, hipMalloc((void**)&nm.Fh, bytes));
    HIPCHK(c, hipMemcpy(nm.Fh, f_history, bytes, hipMemcpyHostToDevice));
    nm.flevels = levels_given;
    return 0;
}

// a load vector added to every step's right-hand side (the self weight of element-wise thickness as consistent nodal loads); null: none
int femo_newmark_set_constant_load(femo_ctx* c, const double* F) {
    HIPCHK(c, hipSetDevice(c->device));
    auto& nm = c->nm;
    if (!nm.ready) return fail(c, "call femo_newmark_setup first");
    nm.has_sw = F != nullptr;
    if (!F) return 0;
    if (!nm.Fsw) HIPCHK(c, hipMalloc((void**)&nm.Fsw, (size_t)c->ndof * sizeof(double)));
    HIPCHK(c, hipMemcpy(nm.Fsw, F, (size_t)c->ndof * sizeof(double), hipMemcpyHostToDevice));
    return 0;
}

void* femo_newmark_ptr(femo_ctx* c, int32_t which) { return which == 0 ? (void*)c->nm.W : which == 1 ? (void*)c->nm.wdot : which == 2 ? (void*)c->nm.Lam : nullptr; }

// March levels 1 .. nsteps from zero initial conditions.  reassemble != 0: the step operator is re-assembled and re-factorised
// before every solve, as the reference does (nonlinear_utils.py:210-233).  iters / relres: nsteps entries (may be null).
int femo_newmark_march(femo_ctx* c, int32_t nsteps, int reassemble, int32_t* iters, double* relres) {
    HIPCHK(c, hipSetDevice(c->device));
    auto& nm = c->nm;
    if (!nm.ready || !nm.Fh) return fail(c, "femo_newmark_march: set-up or force history missing");
    if (nsteps < 1 || nsteps >= nm.levels) return fail(c, "nsteps must be in 1 .. time_levels - 1");
    if (c->op_aK != 0.5 || c->op_aM != nm.a) return fail(c, "the operator was changed after femo_newmark_setup");
    const int64_t n = c->ndof;
    const int vg = vec_grid(n);
    const unsigned char* mask = c->has_mask ? c->mask : nullptr;
    HIPCHK(c, hipMemsetAsync(nm.W, 0, (size_t)n * sizeof(double), c->stream));
    if (nsteps + 1 < nm.levels)            // a shorter march than the last one: no stale levels behind it
        HIPCHK(c, hipMemsetAsync(nm.W + (size_t)(nsteps + 1) * n, 0, (size_t)(nm.levels - nsteps - 1) * n * sizeof(double), c->stream));
    HIPCHK(c, hipMemsetAsync(nm.wdot, 0, (size_t)n * sizeof(double), c->stream));
    for (int i = 1; i <= nsteps; ++i) {
        const double* w_old = nm.W + (size_t)(i - 1) * n;
        double* w_new = nm.W + (size_t)i * n;
        if (load_vector_dev(c, c->b, nm.Fh + (size_t)std::min(i, nm.flevels - 1) * 3 * c->nF)) return 1;
        if (op_apply(c, w_old, c->Ap, nullptr, nullptr, nullptr, false, -0.5, nm.a)) return 1;        // (a M - K/2) w_old
        if (op_apply(c, nm.wdot, c->z, nullptr, nullptr, nullptr, false, 0.0, nm.b)) return 1;        // b M wdot_old
        hipLaunchKernelGGL(k_newmark_rhs, dim3(vg), dim3(256), 0, c->stream, c->b, (const double*)(nm.has_sw ? nm.Fsw : nullptr),
                           (const double*)c->Ap, (const double*)c->z, mask, n);
        if (reassemble) operator_changed(c);
        int32_t it = 0; double rr = 0;
        if (int rc = solve_dispatch(c, c->b, w_new, true, &it, &rr)) return rc;
        if (iters) iters[i - 1] = it;
        if (relres) relres[i - 1] = rr;
        hipLaunchKernelGGL(k_newmark_wdot, dim3(vg), dim3(256), 0, c->stream, nm.wdot, (const double*)w_new, w_old, nm.b, n);
    }
    HIPCHK(c, hipMemcpyAsync(c->w, nm.W + (size_t)nsteps * n, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));   // the reference's self.w
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// history to the host, (time_levels x ndof) row-major (level-major); which: 0 displacement history, 2 adjoint history
int femo_newmark_get_history(femo_ctx* c, int32_t which, double* out) {
    HIPCHK(c, hipSetDevice(c->device));
    auto& nm = c->nm;
    const double* src = which == 0 ? nm.W : which == 2 ? nm.Lam : nullptr;
    if (!nm.ready || !src) return fail(c, "no such history");
    HIPCHK(c, hipMemcpy(out, src, (size_t)nm.levels * c->ndof * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int femo_newmark_set_history(femo_ctx* c, int32_t which, const double* H) {      // a history from outside (the caller's adjoint seed, restarts): level-major
    HIPCHK(c, hipSetDevice(c->device));
    auto& nm = c->nm;
    if (!nm.ready) return fail(c, "call femo_newmark_setup first");
    if (which != 0 && which != 2) return fail(c, "which: 0 displacement history, 2 adjoint history");
    const size_t hb = (size_t)nm.levels * c->ndof * sizeof(double);
    if (which == 2 && !nm.Lam) HIPCHK(c, hipMalloc((void**)&nm.Lam, hb));
    HIPCHK(c, hipMemcpy(which == 0 ? nm.W : nm.Lam, H, hb, hipMemcpyHostToDevice));
    return 0;
}

// Lambda solving (dR/dy)^T Lambda = G for the whole-history residual, by the backward two-vector recursion
//   mu_i = b M lam_{i+1} - mu_{i+1},   A lam_i = G_i + b mu_i + (a M - K/2) lam_{i+1} - b mu_{i+1},   lam_0 = G_0 + (a M - K/2) lam_1 - b mu_1
// (the O(T) form of state_operation_dynamic.py:619-691).  G: (levels x ndof) level-major, host; the result stays on the device
// (femo_newmark_get_history(ctx, 2, ..)) for femo_newmark_residual_T.
int femo_newmark_adjoint(femo_ctx* c, const double* G, int32_t levels) {
    HIPCHK(c, hipSetDevice(c->device));
    auto& nm = c->nm;
    if (!nm.ready) return fail(c, "call femo_newmark_setup first");
    if (levels < 1 || levels > nm.levels) return fail(c, "bad number of levels");
    const int64_t n = c->ndof;
    const int vg = vec_grid(n);
    const unsigned char* mask = c->has_mask ? c->mask : nullptr;
    const size_t hb = (size_t)nm.levels * n * sizeof(double);
    if (!nm.Lam) HIPCHK(c, hipMalloc((void**)&nm.Lam, hb));
    if (!nm.Gh) HIPCHK(c, hipMalloc((void**)&nm.Gh, hb));
    HIPCHK(c, hipMemsetAsync(nm.Lam, 0, hb, c->stream));
    HIPCHK(c, hipMemcpyAsync(nm.Gh, G, (size_t)levels * n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(nm.mu0, 0, (size_t)n * sizeof(double), c->stream));
    HIPCHK(c, hipMemsetAsync(c->lam, 0, (size_t)n * sizeof(double), c->stream));
    double *mu_next = nm.mu0, *mu_i = nm.mu1;
    const double* lam_next = c->lam;                               // zero above the last level
    for (int i = levels - 1; i >= 0; --i) {
        if (op_apply(c, lam_next, c->Ap, nullptr, nullptr, nullptr, false, 0.0, 1.0)) return 1;        // M lam_{i+1}
        if (op_apply(c, lam_next, c->z, nullptr, nullptr, nullptr, false, -0.5, nm.a)) return 1;       // (a M - K/2) lam_{i+1}
        double* Li = nm.Lam + (size_t)i * n;
        hipLaunchKernelGGL(k_newmark_adj, dim3(vg), dim3(256), 0, c->stream, c->b, mu_i, i == 0 ? Li : (double*)nullptr,
                           (const double*)(nm.Gh + (size_t)i * n), (const double*)c->Ap, (const double*)c->z, (const double*)mu_next, mask, nm.b, n);
        if (i == 0) break;
        if (int rc = solve_dispatch(c, c->b, Li, true, nullptr, nullptr)) return rc;
        lam_next = Li;
        std::swap(mu_next, mu_i);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// g_t (thickness length) = sum_i lam_i^T [ K'/2 (w_i + w_{i-1}) + M' (a (w_i - w_{i-1}) - b wdot_{i-1}) ],
// dF (levels x field length of F_solid, level-major; level 0 zero) = (dR_i/df)^T lam_i, for the resident displacement and
// adjoint histories (state_operation_dynamic.py:406-427: the thickness gradient re-assembled per level)
int femo_newmark_residual_T(femo_ctx* c, int32_t levels, double* g_t, double* dF) {
    HIPCHK(c, hipSetDevice(c->device));
    auto& nm = c->nm;
    if (!nm.ready || !nm.Lam) return fail(c, "femo_newmark_residual_T: no adjoint history (femo_newmark_adjoint)");
    if (levels < 1 || levels > nm.levels) return fail(c, "bad number of levels");
    const int64_t n = c->ndof;
    const int vg = vec_grid(n);
    const MeshDev m = mesh_dev(c);
    const FieldsDev f = fields_dev(c);
    const int g = nblk(c->nel, EB);
    if (!c->gradbuf) HIPCHK(c, hipMalloc((void**)&c->gradbuf, (size_t)std::max<int64_t>(c->nT, 1) * sizeof(double)));
    HIPCHK(c, hipMemsetAsync(c->gradbuf, 0, (size_t)c->nT * sizeof(double), c->stream));
    double* dFd = nullptr;
    const size_t fl = (size_t)3 * c->nF;
    HIPCHK(c, hipMalloc((void**)&dFd, (size_t)levels * fl * sizeof(double)));
    hipMemsetAsync(dFd, 0, (size_t)levels * fl * sizeof(double), c->stream);
    // the velocity recursion of the stored history is re-marched in a SCRATCH vector (mu1: free once the adjoint sweep is done) --
    // nm.wdot stays the velocity of the last level the march reached, which femo_newmark_ptr(ctx, 1) exposes
    double* wdv = nm.mu1;
    HIPCHK(c, hipMemsetAsync(wdv, 0, (size_t)n * sizeof(double), c->stream));
    for (int i = 1; i < levels; ++i) {
        const double *wi = nm.W + (size_t)i * n, *wo = nm.W + (size_t)(i - 1) * n, *li = nm.Lam + (size_t)i * n;
        hipLaunchKernelGGL(k_lincomb3, dim3(vg), dim3(256), 0, c->stream, c->p, 1.0, wi, 1.0, wo, 0.0, (const double*)nullptr, n);         // w_i + w_{i-1}
        hipLaunchKernelGGL(k_lincomb3, dim3(vg), dim3(256), 0, c->stream, c->z, nm.a, wi, -nm.a, wo, -nm.b, (const double*)wdv, n);     // a (w_i - w_{i-1}) - b wdot_{i-1}
        ELEM_LAUNCH(c, k_dRdfield_T, COMMA_H, g, EB, m, f, c->tab, c->p, li, 0.5, c->gradbuf);
        ELEM_LAUNCH(c, k_dMdh_T, NOEXTRA, g, EB, m, f, c->tab, c->z, li, 1.0, c->gradbuf);
        ELEM_LAUNCH(c, k_dRdf_T, NOEXTRA, g, EB, m, f, c->tab, li, -1.0, dFd + (size_t)i * fl);
        hipLaunchKernelGGL(k_newmark_wdot, dim3(vg), dim3(256), 0, c->stream, wdv, wi, wo, nm.b, n);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess && g_t) e = hipMemcpy(g_t, c->gradbuf, (size_t)c->nT * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess && dF) e = hipMemcpy(dF, dFd, (size_t)levels * fl * sizeof(double), hipMemcpyDeviceToHost);
    hipFree(dFd);
    HIPCHK(c, e);
    return 0;
}

// ---- forward mode of the transient operator (the "tangent linear model", state_operation_dynamic.py:228-329, 534-605).
// The whole-history Jacobian J = dR/dy couples a level to ALL earlier ones through the velocity recursion (the reference
// carries alternating-sign history sums with factors 2/dt, 4/dt); here the perturbed velocity is marched beside the perturbed
// displacement, O(T).  Dirichlet rows follow the reference's zeroRows: identity in dR/dw_i, zero elsewhere.
// out_i = [J dY]_i + (dR_i/dt) dthickness + (dR_i/df) dF_i.  Any of dY (levels x ndof), dthickness, dF (levels x F length) may be
// null.  The result lands in the adjoint-history buffer (femo_newmark_get_history(ctx, 2, ..)).
int femo_newmark_jvp(femo_ctx* c, int32_t levels, const double* dY, const double* dthickness, const double* dF) {
    HIPCHK(c, hipSetDevice(c->device));
    auto& nm = c->nm;
    if (!nm.ready) return fail(c, "call femo_newmark_setup first");
    if (levels < 1 || levels > nm.levels) return fail(c, "bad number of levels");
    const int64_t n = c->ndof;
    const int vg = vec_grid(n);
    const unsigned char* mask = c->has_mask ? c->mask : nullptr;
    const size_t hb = (size_t)nm.levels * n * sizeof(double);
    if (!nm.Lam) HIPCHK(c, hipMalloc((void**)&nm.Lam, hb));
    if (!nm.Gh) HIPCHK(c, hipMalloc((void**)&nm.Gh, hb));
    HIPCHK(c, hipMemsetAsync(nm.Lam, 0, hb, c->stream));
    if (dY) HIPCHK(c, hipMemcpyAsync(nm.Gh, dY, (size_t)levels * n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    double *dh = nullptr, *dFd = nullptr;
    const size_t fl = (size_t)3 * c->nF;
    if (dthickness) {
        HIPCHK(c, hipMalloc((void**)&dh, (size_t)std::max<int64_t>(c->nT, 1) * sizeof(double)));
        HIPCHK(c, hipMemcpyAsync(dh, dthickness, (size_t)c->nT * sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
    if (dF) {
        HIPCHK(c, hipMalloc((void**)&dFd, (size_t)levels * fl * sizeof(double)));
        HIPCHK(c, hipMemcpyAsync(dFd, dF, (size_t)levels * fl * sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
    const MeshDev m = mesh_dev(c);
    const FieldsDev f = fields_dev(c);
    const int g = nblk(c->nel, EB);
    double* dwd = nm.mu0;                      // perturbed velocity
    double* wd = nm.mu1;                       // velocity of the stored history, re-marched in a scratch vector (nm.wdot stays the march's)
    HIPCHK(c, hipMemsetAsync(dwd, 0, (size_t)n * sizeof(double), c->stream));
    HIPCHK(c, hipMemsetAsync(wd, 0, (size_t)n * sizeof(double), c->stream));
    if (dY) HIPCHK(c, hipMemcpyAsync(nm.Lam, nm.Gh, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));     // level 0: identity
    int rc = 0;
    for (int i = 1; i < levels && !rc; ++i) {
        double* out = nm.Lam + (size_t)i * n;
        const double *wi = nm.W + (size_t)i * n, *wo = nm.W + (size_t)(i - 1) * n;
        if (dY) {
            const double *dwi = nm.Gh + (size_t)i * n, *dwo = nm.Gh + (size_t)(i - 1) * n;
            rc = op_apply(c, dwi, c->r, nullptr, nullptr, nullptr, false, 0.5, nm.a);                    // (a M + K/2) dw_i
            if (!rc) rc = op_apply(c, dwo, c->Ap, nullptr, nullptr, nullptr, false, -0.5, nm.a);         // (a M - K/2) dw_{i-1}
            if (!rc) rc = op_apply(c, dwd, c->z, nullptr, nullptr, nullptr, false, 0.0, nm.b);           // b M dwdot_{i-1}
            if (rc) break;
            hipLaunchKernelGGL(k_lincomb3, dim3(vg), dim3(256), 0, c->stream, out, 1.0, (const double*)c->r, -1.0, (const double*)c->Ap, -1.0, (const double*)c->z, n);
            hipLaunchKernelGGL(k_newmark_wdot, dim3(vg), dim3(256), 0, c->stream, dwd, dwi, dwo, nm.b, n);
        }
        if (dh) {
            // (dR_i/dt) dh = M'[dh] (a (w_i - w_{i-1}) - b wdot_{i-1}) + K'[dh] (w_i + w_{i-1}) / 2
            hipLaunchKernelGGL(k_lincomb3, dim3(vg), dim3(256), 0, c->stream, c->p, 1.0, wi, 1.0, wo, 0.0, (const double*)nullptr, n);
            hipLaunchKernelGGL(k_lincomb3, dim3(vg), dim3(256), 0, c->stream, c->z, nm.a, wi, -nm.a, wo, -nm.b, (const double*)wd, n);
            ELEM_LAUNCH(c, k_apply_dh, NOEXTRA, g, EB, m, f, c->tab, (const double*)dh, 0.5, 0.0, (const double*)c->p, out);
            ELEM_LAUNCH(c, k_apply_dh, NOEXTRA, g, EB, m, f, c->tab, (const double*)dh, 0.0, 1.0, (const double*)c->z, out);
        }
        if (dFd) {
            FieldsDev fdv = f;
            fdv.f = dFd + (size_t)i * fl;
            ELEM_LAUNCH(c, k_load, NOEXTRA, g, EB, m, fdv, c->tab, out, -1.0);                               // (dR_i/df) df_i = - load(df_i)
        }
        if (mask) {
            if (dY) hipLaunchKernelGGL(k_mask_identity, dim3(vg), dim3(256), 0, c->stream, out, (const double*)(nm.Gh + (size_t)i * n), mask, n);
            else hipLaunchKernelGGL(k_mask_zero, dim3(vg), dim3(256), 0, c->stream, out, mask, n);
        }
        hipLaunchKernelGGL(k_newmark_wdot, dim3(vg), dim3(256), 0, c->stream, wd, wi, wo, nm.b, n);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (dh) hipFree(dh);
    if (dFd) hipFree(dFd);
    if (rc) return rc;
    HIPCHK(c, e);
    return 0;
}

// dY = J^-1 dR, level by level (the direct method):  dw_0 = dr_0;  A dw_i = dr_i + (a M - K/2) dw_{i-1} + b M dwdot_{i-1} on the free
// rows, dw_i = dr_i on the Dirichlet rows (their coupling into the free rows moved to the right-hand side).  The result lands in the
// adjoint-history buffer.
int femo_newmark_tangent(femo_ctx* c, const double* dR, int32_t levels) {
    HIPCHK(c, hipSetDevice(c->device));
    auto& nm = c->nm;
    if (!nm.ready) return fail(c, "call femo_newmark_setup first");
    if (levels < 1 || levels > nm.levels || !dR) return fail(c, "bad arguments");
    const int64_t n = c->ndof;
    const int vg = vec_grid(n);
    const unsigned char* mask = c->has_mask ? c->mask : nullptr;
    const size_t hb = (size_t)nm.levels * n * sizeof(double);
    if (!nm.Lam) HIPCHK(c, hipMalloc((void**)&nm.Lam, hb));
    if (!nm.Gh) HIPCHK(c, hipMalloc((void**)&nm.Gh, hb));
    HIPCHK(c, hipMemsetAsync(nm.Lam, 0, hb, c->stream));
    HIPCHK(c, hipMemcpyAsync(nm.Gh, dR, (size_t)levels * n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(nm.Lam, nm.Gh, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    double* dwd = nm.mu0;
    HIPCHK(c, hipMemsetAsync(dwd, 0, (size_t)n * sizeof(double), c->stream));
    for (int i = 1; i < levels; ++i) {
        const double *dri = nm.Gh + (size_t)i * n, *dwo = nm.Lam + (size_t)(i - 1) * n;
        double* dwi = nm.Lam + (size_t)i * n;
        if (op_apply(c, dwo, c->Ap, nullptr, nullptr, nullptr, false, -0.5, nm.a)) return 1;
        if (op_apply(c, dwd, c->z, nullptr, nullptr, nullptr, false, 0.0, nm.b)) return 1;
        hipLaunchKernelGGL(k_lincomb3, dim3(vg), dim3(256), 0, c->stream, c->b, 1.0, dri, 1.0, (const double*)c->Ap, 1.0, (const double*)c->z, n);
        if (mask) {
            hipLaunchKernelGGL(k_mask_select, dim3(vg), dim3(256), 0, c->stream, c->p, dri, mask, n);
            if (op_apply(c, c->p, c->r, nullptr, nullptr, nullptr, false, 0.5, nm.a)) return 1;          // coupling of the Dirichlet values
            hipLaunchKernelGGL(k_axpby, dim3(vg), dim3(256), 0, c->stream, c->b, -1.0, (const double*)c->r, 1.0, n);
            hipLaunchKernelGGL(k_mask_zero, dim3(vg), dim3(256), 0, c->stream, c->b, mask, n);
        }
        if (int rc = solve_dispatch(c, c->b, dwi, true, nullptr, nullptr)) return rc;
        if (mask) hipLaunchKernelGGL(k_mask_identity, dim3(vg), dim3(256), 0, c->stream, dwi, dri, mask, n);
        hipLaunchKernelGGL(k_newmark_wdot, dim3(vg), dim3(256), 0, c->stream, dwd, (const double*)dwi, dwo, nm.b, n);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// ---- CSR assembly of the elastic stiffness (what assembleMatrix(dR_du) returns in the reference,
// csdl_alpha_opt/state_operation.py:289; fea/utils_dolfinx.py:200-206) -----------------------------------------
// perm[k]: index into the element-matrix buffer (element * ld*ld + i*ld + j) of the k-th contribution in
// destination order; dest[k]: its position in the CSR value array (non-decreasing).
int femo_set_csr_map(femo_ctx* c, int32_t nnz, int64_t ncontrib, const int32_t* perm, const int32_t* dest) {
    HIPCHK(c, hipSetDevice(c->device));
    if (ncontrib != (int64_t)c->nel * c->ld * c->ld) return fail(c, "ncontrib must be nel * ldof^2");
    for (int64_t k = 0; k < ncontrib; ++k) {
        if (perm[k] < 0 || perm[k] >= ncontrib || dest[k] < 0 || dest[k] >= nnz || (k && dest[k] < dest[k - 1]))
            return fail(c, "bad CSR map (range or ordering)");
    }
    void* old[] = {c->csr_perm, c->csr_dest, c->csr_vals, c->csr_ke};
    for (void* p : old) if (p) hipFree(p);
    c->csr_perm = c->csr_dest = nullptr; c->csr_vals = c->csr_ke = nullptr;
    HIPCHK(c, hipMalloc((void**)&c->csr_perm, (size_t)ncontrib * sizeof(int)));
    HIPCHK(c, hipMalloc((void**)&c->csr_dest, (size_t)ncontrib * sizeof(int)));
    HIPCHK(c, hipMalloc((void**)&c->csr_vals, (size_t)nnz * sizeof(double)));
    HIPCHK(c, hipMalloc((void**)&c->csr_ke, (size_t)ncontrib * sizeof(double)));
    HIPCHK(c, hipMemcpy(c->csr_perm, perm, (size_t)ncontrib * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->csr_dest, dest, (size_t)ncontrib * sizeof(int), hipMemcpyHostToDevice));
    c->csr_ncontrib = ncontrib; c->csr_nnz = nnz;
    return 0;
}

// pattern + destination-sorted contribution map on the device (csr_map.h); *nnz_out = number of stored entries
int femo_build_csr_map(femo_ctx* c, int32_t* nnz_out) {
    HIPCHK(c, hipSetDevice(c->device));
    const long long nc = (long long)c->nel * c->ld * c->ld;
    if (nc >= (1ll << 31)) return fail(c, "CSR export: more than 2^31 element contributions (the matrix-free solvers have no such limit)");
    void* old[] = {c->csr_perm, c->csr_dest, c->csr_rowptr, c->csr_colidx, c->csr_vals, c->csr_ke};
    for (void* p : old) if (p) hipFree(p);
    c->csr_perm = c->csr_dest = c->csr_rowptr = c->csr_colidx = nullptr; c->csr_vals = c->csr_ke = nullptr;
    long long *k0 = nullptr, *k1 = nullptr;
    int *v0 = nullptr, *flags = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = 0, t2 = 0;
    HIPCHK(c, hipMalloc((void**)&k0, nc * sizeof(long long)));
    HIPCHK(c, hipMalloc((void**)&k1, nc * sizeof(long long)));
    HIPCHK(c, hipMalloc((void**)&v0, nc * sizeof(int)));
    HIPCHK(c, hipMalloc((void**)&c->csr_perm, nc * sizeof(int)));
    HIPCHK(c, hipMalloc((void**)&c->csr_dest, nc * sizeof(int)));
    HIPCHK(c, hipMalloc((void**)&flags, nc * sizeof(int)));
    const unsigned gb = (unsigned)((nc + 255) / 256);
    hipLaunchKernelGGL(k_csr_keys, dim3(gb), dim3(256), 0, c->stream, nc, c->nel, c->ld, c->npc, c->ndof_u, (long long)c->ndof, (const int*)c->cellp2,
                       (const int*)c->cells, k0, v0, c->cr ? c->nn : -1);
    int bits = 1;
    while (bits < 63 && (1ull << bits) <= (unsigned long long)c->ndof * (unsigned long long)c->ndof) ++bits;
    HIPCHK(c, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, k0, k1, v0, c->csr_perm, (int)nc, 0, bits, c->stream));
    HIPCHK(c, hipcub::DeviceScan::InclusiveSum(nullptr, t2, flags, c->csr_dest, (int)nc, c->stream));
    tmp_bytes = std::max(tmp_bytes, t2);
    HIPCHK(c, hipMalloc(&tmp, tmp_bytes));
    HIPCHK(c, hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, k0, k1, v0, c->csr_perm, (int)nc, 0, bits, c->stream));
    hipLaunchKernelGGL(k_csr_heads, dim3(gb), dim3(256), 0, c->stream, nc, (const long long*)k1, flags);
    HIPCHK(c, hipcub::DeviceScan::InclusiveSum(tmp, tmp_bytes, flags, c->csr_dest, (int)nc, c->stream));
    int nnz = 0;
    HIPCHK(c, hipMemcpyAsync(&nnz, c->csr_dest + (nc - 1), sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMalloc((void**)&c->csr_colidx, (size_t)nnz * sizeof(int)));
    HIPCHK(c, hipMalloc((void**)&c->csr_rowptr, ((size_t)c->ndof + 1) * sizeof(int)));
    HIPCHK(c, hipMemsetAsync(c->csr_rowptr, 0, ((size_t)c->ndof + 1) * sizeof(int), c->stream));
    hipLaunchKernelGGL(k_csr_pattern, dim3(gb), dim3(256), 0, c->stream, nc, (const long long*)k1, (const int*)flags, c->csr_dest, (long long)c->ndof,
                       c->csr_colidx, c->csr_rowptr);
    HIPCHK(c, hipcub::DeviceScan::InclusiveSum(tmp, tmp_bytes, c->csr_rowptr, c->csr_rowptr, c->ndof + 1, c->stream));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    hipFree(k0); hipFree(k1); hipFree(v0); hipFree(flags); hipFree(tmp);
    HIPCHK(c, hipMalloc((void**)&c->csr_vals, (size_t)nnz * sizeof(double)));
    HIPCHK(c, hipMalloc((void**)&c->csr_ke, (size_t)nc * sizeof(double)));
    c->csr_ncontrib = nc; c->csr_nnz = nnz;
    if (nnz_out) *nnz_out = nnz;
    return 0;
}

// the pattern to the host: rowptr (ndof + 1), colidx (nnz), int32 (sorted columns per row)
int femo_get_csr_pattern(femo_ctx* c, int32_t* rowptr, int32_t* colidx) {
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->csr_rowptr) return fail(c, "call femo_build_csr_map first");
    HIPCHK(c, hipMemcpy(rowptr, c->csr_rowptr, ((size_t)c->ndof + 1) * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(colidx, c->csr_colidx, (size_t)c->csr_nnz * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}

// vals (host, nnz) = CSR values of aK K + (no inertia) for the current fields; ms[0] = element matrices, ms[1] = scatter
int femo_assemble_csr(femo_ctx* c, double* vals, double* ms2) {
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->csr_perm) return fail(c, "call femo_set_csr_map first");
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    ELEM_LAUNCH_S(c, k_element_matrices, NOEXTRA, c->nel, 64, QPOINT_LDS(c), mesh_dev(c), fields_dev(c), c->tab, 0, c->nel, c->csr_ke);
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    HIPCHK(c, hipMemsetAsync(c->csr_vals, 0, (size_t)c->csr_nnz * sizeof(double), c->stream));
    hipLaunchKernelGGL(k_csr_segmented, dim3((unsigned)((c->csr_ncontrib + 256 * CSR_R - 1) / (256 * CSR_R))), dim3(256), 0, c->stream, c->csr_ncontrib,
                       c->csr_perm, c->csr_dest, c->csr_ke, c->csr_vals);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float a = 0, b = 0;
    hipEventElapsedTime(&a, c->ev[0], c->ev[1]);
    hipEventElapsedTime(&b, c->ev[1], c->ev[2]);
    if (ms2) { ms2[0] = a; ms2[1] = b; }
    if (vals) HIPCHK(c, hipMemcpy(vals, c->csr_vals, (size_t)c->csr_nnz * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int femo_set_stress_params(femo_ctx* c, double m, double rho) {
    if (!(m > 0) || !(rho > 0)) return fail(c, "stress aggregation parameters must be positive");
    c->stress_m = m; c->stress_rho = rho;
    return 0;
}

// alpha of pnorm_stress = 1/alpha int (m vm)^rho J dx given by the caller (RMShellPDE.pnorm_stress(alpha=...), rm_shell_pde.py:112-128)
// for the whole mesh (sel = -1) or one sub-domain; alpha <= 0 returns to "the reference area, evaluated at first use"
int femo_set_stress_alpha(femo_ctx* c, int32_t sel, double alpha) {
    if (sel < -1 || sel >= c->ntags) return fail(c, "unknown sub-domain");
    (sel < 0 ? c->stress_alpha : c->alpha_tag[sel]) = alpha > 0 ? alpha : -1.0;
    return 0;
}

int femo_set_cell_tags(femo_ctx* c, const int32_t* tags, int64_t n, int32_t ntags) {
    HIPCHK(c, hipSetDevice(c->device));
    if (n != c->nel) return fail(c, "cell tags need one entry per cell");
    if (ntags < 0) return fail(c, "negative number of sub-domains");
    for (int64_t i = 0; i < n; ++i)
        if (tags[i] < -1 || tags[i] >= ntags) return fail(c, "cell tag out of range");
    if (!c->ctag) HIPCHK(c, hipMalloc((void**)&c->ctag, (size_t)c->nel * sizeof(int)));
    HIPCHK(c, hipMemcpy(c->ctag, tags, (size_t)c->nel * sizeof(int), hipMemcpyHostToDevice));
    c->ntags = ntags; c->csel = -1;
    c->alpha_tag.assign(ntags, -1.0);
    return 0;
}

int femo_select_subdomain(femo_ctx* c, int32_t sel) {
    if (sel < -1 || sel >= c->ntags) return fail(c, "unknown sub-domain");
    c->csel = sel;
    return 0;
}

int femo_field_output(femo_ctx* c, const char* name, double* out, int64_t n) {
    HIPCHK(c, hipSetDevice(c->device));
    const std::string fname(name ? name : "");
    // von_Mises_stress(surface = 'Top' | 'Mid' | 'Bot') (rm_shell_pde.py:153-165): xi2 = h/2, 0, -h/2
    const double zf = fname == "stress" ? 0.5 : fname == "stress_mid" ? 0.0 : fname == "stress_bot" ? -0.5 : 2.0;
    if (zf > 1.0) return fail(c, "unknown field output '" + fname + "' (stress, stress_mid, stress_bot)");
    if (n != (int64_t)c->nvc * c->nel) return fail(c, "the DG1 stress field has nvc * nel entries");
    double* d = nullptr;
    HIPCHK(c, hipMalloc((void**)&d, (size_t)n * sizeof(double)));
    ELEM_LAUNCH(c, k_stress_field, NOEXTRA, nblk(c->nel, EB), EB, mesh_dev(c), fields_dev(c), c->tab, c->w, zf, d);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(out, d, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
    hipFree(d);
    HIPCHK(c, e);
    return 0;
}

int femo_last_timing(const femo_ctx* c, double* out5) {
    for (int i = 0; i < 5; ++i) out5[i] = c->timing[i];
    return 0;
}

// average duration (ms) of `reps` back-to-back launches of a named kernel, HIP events on the
// context's stream: "apply" (element operator, y += K p), "pcg_update", "pcg_direction", "diag"
int femo_bench_kernel(femo_ctx* c, const char* name, int32_t reps, double* avg_ms) {
    HIPCHK(c, hipSetDevice(c->device));
    const std::string s(name ? name : "");
    const int64_t n = c->ndof;
    const int vg = vec_grid(n);
    if (reps < 1) return fail(c, "reps must be >= 1");
    if (refresh_diag(c)) return 1;
    HIPCHK(c, hipMemsetAsync(c->scal, 0, 8 * sizeof(double), c->stream));
    hipLaunchKernelGGL(k_fill, dim3(vg), dim3(256), 0, c->stream, c->scal, 1.0, 8);
    auto launch = [&]() -> int {
        if (s == "apply") return op_apply(c, c->p, c->Ap, c->scal + 7, nullptr, nullptr, false);
        if (s == "pcg_update") { hipLaunchKernelGGL(k_pcg_update, dim3(vg), dim3(256), 0, c->stream, c->tmp, c->r, c->z, c->p, c->Ap, c->dinv, (const unsigned char*)nullptr, n, c->scal, 0); return 0; }
        if (s == "pcg_direction") { hipLaunchKernelGGL(k_pcg_direction, dim3(vg), dim3(256), 0, c->stream, c->p, c->z, c->Ap, n, c->scal, 0); return 0; }
        if (s == "diag") { ELEM_LAUNCH(c, k_diag, NOEXTRA, nblk(c->nel, EB), EB, mesh_dev(c), fields_dev(c), c->tab, c->tmp); return 0; }
        // one application of the factor to 1 / 2 / 4 vectors (the sweeps alone: no interleaving copies); needs a factorisation
        if (s == "sweeps1" || s == "sweeps2" || s == "sweeps4") {
            if (!c->fr.factored) return fail(c, "no factorisation to sweep with");
            if (mr_alloc(c)) return 1;
            if (s == "sweeps1") { hipLaunchKernelGGL(k_fill, dim3(vg), dim3(256), 0, c->stream, c->z, 1.0, n); return frontal_solve(c, c->z); }
            const int nr = s == "sweeps2" ? 2 : 4;
            hipLaunchKernelGGL(k_fill, dim3(vec_grid(n * nr)), dim3(256), 0, c->stream, c->mr_v, 1.0, n * nr);
            return nr == 2 ? frontal_solve_multi<2>(c, c->mr_v, c->mr_y) : frontal_solve_multi<4>(c, c->mr_v, c->mr_y);
        }
        return fail(c, "unknown kernel '" + s + "'");
    };
    for (int i = 0; i < 3; ++i)
        if (int rc = launch()) return rc;
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    for (int i = 0; i < reps; ++i) launch();
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0;
    hipEventElapsedTime(&ms, c->ev[0], c->ev[1]);
    *avg_ms = ms / reps;
    return 0;
}

void* femo_device_ptr(femo_ctx* c, const char* name) {
    const std::string s(name ? name : "");
    if (s == "state") return c->w;
    if (s == "adjoint") return c->lam;
    int64_t n;
    return field_ptr(c, name, &n);
}

}  // extern "C"
