/* libfemo_hip.so -- C ABI of the MI355X-native Reissner-Mindlin shell forward + adjoint path.
 *
 * The reference (LSDOlab/femo_alpha) has no FFI seam of its own: its hot path is Python that
 * calls dolfinx / PETSc through pybind11.  This ABI sits *beneath* the
 * Python operator classes the reference exposes, one entry point per dolfinx/PETSc service the
 * reference's operators consume.  Each declaration cites the reference call it replaces
 * (paths relative to /root/reference/femo_alpha).
 *
 * Conventions: every call returns 0 on success, non-zero on failure with a message available
 * from femo_last_error(); all host buffers are owned by the caller, float64 / int32, 1-D;
 * the context owns every device buffer; one context per GPU; not thread-safe; calls are
 * synchronous on return unless stated otherwise.
 *
 * State vector layout (opaque to callers of the reference as well, rm_shell_model.py:505-527):
 *   w = [ u(P2 node 0) xyz ... u(P2 node nP2-1) xyz | theta(vertex 0) xyz ... ]   ndof = 3 nP2 + 3 nn
 */
#ifndef FEMO_HIP_H
#define FEMO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct femo_ctx femo_ctx;

/* Library / device probing ------------------------------------------------------------------ */
int femo_version(void);
int femo_device_count(void);
/* Message of the last failed call on ctx (ctx == NULL: last failed femo_create). */
const char* femo_last_error(const femo_ctx* ctx);

/* Context = mesh + CG2xCG1 numbering on one GPU.
 * Replaces RMShellPDE.__init__ / ShellElement.setUpFunctionSpace (rm_shell/rm_shell_pde.py:25-47,
 * linear_shell_fenicsx/linear_shell_model.py:47-86).
 *   nvc            4 (quads, CCW) or 3 (triangles)
 *   xyz            nn*3 vertex coordinates
 *   cells          nel*nvc vertex ids
 *   cell_p2        nel*npc P2 node ids (npc = 9 quads / 6 triangles: vertices, edge midpoints, centre)
 *   elementwise_material / elementwise_pressure   DG0 instead of CG1 for VT / VF (rm_shell_pde.py:37-44)
 *   nquad          the rule of the static forms (the reference leaves it to UFL's estimate: plain dx, linear_shell_model.py:88-103).
 *                  Quadrilaterals: Gauss points per direction, 2..6.  Triangles: the DEGREE of the fully symmetric rule --
 *                  4 (6 points), 6 (12 points; also what 0 selects), 9 (19 points: UFL's estimate for these forms) or 12 (33 points);
 *                  exact literals of every rule: scripts/derive_triangle_rules.py.  The p-norm stress measure is the reference's
 *                  quadrature_degree 4 whatever nquad says (rm_shell_model.py:200-205): 3 x 3 Gauss / the 6-point rule. */
int femo_create(femo_ctx** out, int device, int32_t nn, int32_t nel, int32_t nvc, int32_t nP2,
                const double* xyz, const int32_t* cells, const int32_t* cell_p2,
                int elementwise_material, int elementwise_pressure, int nquad);
/* The same with nghost extra entries appended to every state-sized vector: DOFs that exist on other
 * element partitions only (replicated separator DOFs of the multi-GPU driver).  femo_ndof then returns
 * mesh DOFs + nghost.  No element kernel touches the extra entries. */
int femo_create_ghost(femo_ctx** out, int device, int32_t nn, int32_t nel, int32_t nvc, int32_t nP2,
                      const double* xyz, const int32_t* cells, const int32_t* cell_p2,
                      int elementwise_material, int elementwise_pressure, int nquad, int32_t nghost);
/* The same with the mixed element chosen explicitly (ShellElement.setUpFunctionSpace, linear_shell_model.py:47-86):
 *   element 0   'CG2CG1' -- or 'CG1CG1' when nP2 == nn -- exactly femo_create_ghost;
 *   element 1   'CG2CR1' (:68-73; triangles only, as in the reference): displacement on the P2 nodes, rotation on the EDGE MIDPOINTS with
 *               the Crouzeix-Raviart functions.  The rotation nodes are the P2 nodes nn .. nP2 - 1 (cell_p2 lists the edge midpoints of a
 *               cell behind its vertices): state vector [u(P2 nodes) xyz | theta(edge midpoints) xyz], femo_ndof = 3 nP2 + 3 (nP2 - nn).
 *               Provided: operator, both Dirichlet treatments, solves, scalar outputs, stress outputs, the adjoint chain for thickness / E /
 *               nu / F_solid / uhat (shape), the transient march with its adjoint, the CSR export, ghost entries (element partitions: the
 *               rotation DOFs of a replicated separator belong to its edge midpoints). */
int femo_create_element(femo_ctx** out, int device, int32_t nn, int32_t nel, int32_t nvc, int32_t nP2,
                        const double* xyz, const int32_t* cells, const int32_t* cell_p2,
                        int elementwise_material, int elementwise_pressure, int nquad, int32_t nghost, int element);
void femo_destroy(femo_ctx* ctx);

int64_t femo_ndof(const femo_ctx* ctx);
/* Length of an input field: "thickness","E","nu","density" (nn or nel), "F_solid" (3*nn or 3*nel), "uhat" (3*nn). */
int64_t femo_field_size(const femo_ctx* ctx, const char* name);

/* Dirichlet data.
 * Penalty form: beta/h_K * ||J F^-T N|| * (w - 0).v on the listed (cell, local edge) pairs,
 * replaces ElasticModelShapeOpt.penaltyResidual (linear_shell_model.py:323-333) with the
 * ds(100)/dS(100) measures of rm_shell_model.py:88-95.
 * Strong form: zero values on the listed DOFs, replaces dirichletbc/locate_dofs_geometrical
 * (rm_shell_model.py:168-180) as consumed by assembleSystem (fea/utils_dolfinx.py:208-221). */
int femo_set_penalty_facets(femo_ctx* ctx, int32_t nfacets, const int32_t* cell_and_local_edge, double beta);
int femo_set_strong_dofs(femo_ctx* ctx, int32_t n, const int32_t* dofs);

/* Copy an input field into the context (a length-1 array broadcasts) --
 * replaces update(Function, ndarray) (fea/utils_dolfinx.py:319-330).  Besides the six inputs of the model, "dirichlet"
 * (femo_ndof values) sets the prescribed state g of the penalty term beta/h_E |J F^-T N| (w - g).v
 * (linear_shell_model.py:323-333; the reference's RMShellModel always passes zeros, rm_shell_model.py:183-185). */
int femo_set_field(femo_ctx* ctx, const char* name, const double* values, int64_t n);
int femo_get_field(femo_ctx* ctx, const char* name, double* values, int64_t n);

/* State access -- replaces getFuncArray / setFuncArray on the state Function
 * (fea/utils_dolfinx.py:174-186). */
int femo_set_state(femo_ctx* ctx, const double* w);
int femo_get_state(femo_ctx* ctx, double* w);

/* y = (dR/dw) x, matrix-free, element by element (elastic energy Hessian + penalty, strong-BC rows
 * replaced by identity) -- replaces assembleMatrix(dR_du) followed by a mat-vec
 * (csdl_alpha_opt/state_operation.py:289; fea/utils_dolfinx.py:275-283). */
int femo_apply_K(femo_ctx* ctx, const double* x, double* y);
/* r = R(w) for the current fields (w == NULL: the stored state) -- replaces
 * assemble_vector(form(residual_form)) (fea/utils_dolfinx.py:194-198; linear_shell_model.py:308-321). */
int femo_residual(femo_ctx* ctx, const double* w, double* r);
/* F = int f . v J dx, the load part of the residual (linear_shell_model.py:320). */
int femo_load_vector(femo_ctx* ctx, double* F);
/* Diagonal of the operator femo_apply_K applies (the Jacobi preconditioner). */
int femo_diagonal(femo_ctx* ctx, double* d);
/* Dense element matrices K_e, nel*ldof*ldof, element-local order [u_a xyz ..., theta_b xyz ...]
 * (what FFCx's tabulate_tensor produces for derivative(residual, w)). */
int femo_element_matrices(femo_ctx* ctx, int32_t first, int32_t count, double* Ke);

/* Multifrontal Cholesky preconditioner: upload the symbolic analysis (elimination tree of a nested
 * dissection of the elements, fronts, index maps -- femo_alpha_amd/solver/symbolic.py).  Plays the part
 * of MUMPS' analysis phase behind setUpKSP_MUMPS (fea/utils_dolfinx.py:514-531).  Arrays: per tree node
 * nf, npiv, parent, left, right; front_off (ntree+1, doubles), dof_off (ntree+1); front_dofs / up_map
 * (dof_off[ntree] entries); level_off (nlevels+1) into level_nodes (ntree, children before parents);
 * elem_front (nel) and elem_map (nel*ldof) place every element matrix in its leaf front. */
int femo_set_frontal_plan(femo_ctx* ctx, int32_t ntree, int32_t nlevels, const int32_t* nf, const int32_t* npiv,
                          const int64_t* front_off, const int64_t* dof_off, const int32_t* front_dofs,
                          const int32_t* up_map, const int32_t* parent, const int32_t* left, const int32_t* right,
                          const int32_t* level_off, const int32_t* level_nodes, const int32_t* elem_front,
                          const int32_t* elem_map);
/* Numeric factorisation for the current fields (also run lazily by the solves when preconditioner == 2):
 * element matrices -> leaf fronts -> batched partial Cholesky level by level.  Replaces ksp.setUp() with
 * PC 'lu' / MUMPS (fea/utils_dolfinx.py:495-531). */
int femo_factorize(femo_ctx* ctx);
/* One factorisation with a HIP event pair around every launch; per kernel class (0 rows below the diagonal blocks,
 * 1 diagonal blocks, 2 trailing rank-k updates, 3 extend_add, 4 front_assemble, 5 memset, 6 inversion of L11, 7 unused):
 * out32[0..7] total ms, out32[8..15] launches, out32[16..23] algorithmic flops (lower triangles only) executed by the
 * launches of the class -- counted from each launch's own K and column ranges -- and out32[24..31] their compulsory HBM
 * bytes (every operand entry read once, results read and written once); classes 0, 1, 2, zero otherwise. */
int femo_factorize_profile(femo_ctx* ctx, double* out32);
/* One application of the factor (forward + backward sweep) with HIP events between the launches, per tree level L:
 * out[4 L + 0 / 1] forward sweep, first / second launch; out[4 L + 2 / 3] backward sweep (ms); n >= 4 * levels. */
int femo_sweep_profile(femo_ctx* ctx, double* out, int64_t n);
/* ... and of the sweeps with nrhs = 2 or 4 interleaved vectors (femo_solve_linear_multi): out[2 L] forward, out[2 L + 1] backward sweep of
 * level L in ms; n >= 2 nlevels. */
int femo_sweep_profile_multi(femo_ctx* ctx, int32_t nrhs, double* out, int64_t n);
/* out6: [0] front assembly ms, [1] factorisation ms, [2] front storage GB, [3] factor GFLOP,
 *       [4] non-positive pivots repaired, [5] number of fronts. */
int femo_frontal_info(const femo_ctx* ctx, double* out6);

/* Solver configuration. preconditioner: 0 = Jacobi, 2 = multifrontal Cholesky (needs a frontal plan).
 * check_every: convergence is polled on the host every this many PCG iterations. */
int femo_set_solver(femo_ctx* ctx, int preconditioner, double rtol, int32_t maxit, int32_t check_every);
/* Schedule switches and failure policy of the solvers (they are per context; nothing is read from the environment):
 *   "strict" (default 1)             a Krylov solve that reaches maxit short of rtol returns 4 with a message -- the reference
 *                                    solves with a direct LU (fea/utils_dolfinx.py:466,514-531), an unconverged state has no
 *                                    counterpart there; 0 returns 0 and leaves the judgement to iters / relres
 *   "allow_pivot_repair" (default 0) 0: a non-positive pivot makes femo_factorize (and the solves that call it) return 5;
 *                                    1: such pivots are replaced and counted (femo_frontal_info [4])
 *   "trailing" 0 auto | 1 left-looking | 2 right-looking rank-k updates; "left_min", "left_max" (auto: levels with this
 *   many fronts are left-looking; defaults 64, 2048); "super_panel" (default 512; 0 = off), "super_panel_cnt" (default 64):
 *   right-looking levels of at most that many fronts update the trailing matrix once per super-panel of that many factor
 *   columns instead of once per 128; "super_panel_ahead" (default 0): that update on a second stream beside the next
 *   super-panel's panels; "lookahead" 0/1, "lookahead_cnt": the same for the 128-column schedule;
 *   "fused_schur" (default 1): Schur complements are gathered from the children by the rank-k update that touches them
 *   first instead of by the extend-add;
 *   "fuse_rows" (default 1), "fuse_rows_cnt" (2048), "fuse_rows_np" (256): levels of at least that many fronts whose widest front has at most
 *   that many pivots form the rows under a diagonal block inside the diagonal-block kernel;
 *   "rows_fine_wg" (96), "narrow_fine_wg" (128): launches of at most that many workgroups take the kernels with 16 rows / a 16 x 16 block per wave;
 *   "stale_factor" (default 0)       n > 0: when only FIELDS (thickness, E, nu, density, uhat) changed since the last factorisation, that
 *                                    factor is kept as the PCG preconditioner and the operator is re-factorised only if a solve has not
 *                                    converged after n iterations (the iteration then restarts from the iterate it reached), and at once
 *                                    if a field has moved more than "stale_rel" (default 2e-3, relative L2 norm) from the factor's design:
 *                                    beyond that the extra iterations cost more than the factorisation (profiles/r5_stale_factor.txt).  PCG works on
 *                                    the current matrix-free operator: the solution is the same.  For optimisation loops, where successive
 *                                    designs are close; the reference itself never refreshes its derivative matrices
 *                                    (csdl_alpha_opt/state_operation.py:130-131).  femo_last_timing [3] says which factor a solve used;
 *   "strip_cnt" (0), "strip_kmax" (160), "strip_depth" (1): rank-k updates with K <= strip_kmax on levels of at least strip_cnt fronts by
 *   one workgroup per 64-row strip of a front (k_schur_strip) -- measured slower than the tile kernel, off (profiles/r5_strip_ab.txt);
 *   "sweep_fuse" (0): all consecutive wide levels of a triangular sweep as ONE launch, tiles ordered by per-front counters -- measured
 *   slower than the level-wise launches, off (profiles/r5_sweep_fuse_ab.txt); "sweep_read_mode" (its read of other workgroups' values).
 *   A tile of that launch waits for tiles with smaller workgroup ids: forward progress assumes that the workgroups of a launch are
 *   dispatched in id order (observed on gfx950, not specified); the wait is bounded and poisons its result with NaN when it runs out;
 *   "apply_lanes" (0): 5 = the matrix-free operator with five lanes per element instead of a quad of lanes -- measured slower, off
 *   (profiles/r6_apply_lanes.txt); "diag_t" (0), "sweep_ahead" (2), "multi_rhs" (1): DESIGN.md section 0;
 *   "assemble_fc" (1): front assembly with one workgroup per leaf front -- zero fill, element columns and their sums without float
 *   atomics, the front written once (k_front_assemble_fc).  1: where it was measured to pay (triangles and CG1CG1, or at least 20
 *   quadrature points per cell), 2: always, 0: never = one wave per element adding with atomics into zero-filled fronts
 *   (profiles/r5_assemble_fc_ab.txt);
 *   "sweep_w" (0): W = L21 L11^-1 stored where L21 was, a wide level of a sweep as one launch instead of two -- an application 4 % shorter,
 *   the factorisation 1.5 ms longer, off (profiles/r5_sweep_w_ab.txt); changing it discards the factor;
 *   "equilibrate" (0), "precond_nquad" (0): measured experiments, off;
 *   "grid_chunk" (fronts per launch, <= 65535);
 *   "wide_np", "wide_cnt" (which tree levels take the wide triangular-solve kernels; before femo_set_frontal_plan);
 *   "swork_slots" (default 8192; before femo_set_frontal_plan): 128 x 128 scratch blocks for the diagonal-block inverses of
 *   the levels solved with one workgroup per front -- levels with more fronts are factorised in chunks of that many;
 *   "bnd_tiled_nb" (backward sweep: levels whose largest boundary block has at least this many rows use 128 x 128 tiles
 *   with atomics for L21^T x, the others one workgroup per 16 columns); "profile_verbose" (per-launch timings of
 *   femo_factorize_profile on stderr). */
int femo_set_option(femo_ctx* ctx, const char* key, double value);
/* Krylov method for the state / adjoint / linear solves: 0 = conjugate gradients (default; the operator is SPD),
 * 1 = right-preconditioned BiCGStab with the same preconditioner (femo_set_solver).  Stands where the reference
 * chooses its PETSc KSP type (fea/utils_dolfinx.py:495-531). */
int femo_set_krylov(femo_ctx* ctx, int method);

/* Forward solve R(w) = 0 for the current fields, result kept on the device as the state.
 * Replaces FEA.solve -> solveNonlinear -> NewtonSolver + MUMPS LU (fea/fea_dolfinx.py:159-170,
 * fea/utils_dolfinx.py:338-352,438-468): the residual is linear in w, so one PCG solve of
 * K w = F reproduces what the reference's three Newton iterations converge to.
 * zero_guess != 0 starts from w = 0, otherwise from the stored state (the reference keeps the
 * previous solution as initial guess, fea_dolfinx.py:45,165). */
int femo_solve_state(femo_ctx* ctx, int zero_guess, int32_t* iters, double* relres);
/* x = K^-1 rhs with BC rows of x zeroed afterwards -- replaces FEA.solveLinearBwd / solveLinearFwd
 * (fea/fea_dolfinx.py:173-203) and the BC zeroing of state_operation.py:216-218.  K is symmetric, so
 * the same call serves both modes (reference quirk Q3). */
int femo_solve_linear(femo_ctx* ctx, const double* rhs, double* x, int32_t* iters, double* relres);
/* Several right-hand sides at once: rhs and x hold nrhs vectors of femo_ndof entries, one after the other; iters / relres: nrhs entries
 * (or NULL).  With the multifrontal preconditioner (femo_set_solver preconditioner 2) the right-hand sides go through the triangular
 * sweeps in groups of up to four with the vectors interleaved, so that the factor is read once per group instead of once per vector
 * (option "multi_rhs" 0: one at a time); every right-hand side keeps its own PCG recurrence and stopping test, so the iterates are those
 * of nrhs separate femo_solve_linear calls.  Replaces repeated StateOperation.apply_inverse_jacobian calls (state_operation.py:188-220),
 * forward or reverse (K is symmetric).  Status 4 if any right-hand side stops at maxit short of rtol (option "strict"). */
int femo_solve_linear_multi(femo_ctx* ctx, int32_t nrhs, const double* rhs, double* x, int32_t* iters, double* relres);

/* pressure = A^-1 force with A the consistent mass matrix of the pressure space [CG1]^3 (node-major xyz, 3 nn entries) -- replaces
 * csdl.solve_linear(A, force) on the matrix of RMShellPDE.construct_force_to_pressure_map (rm_shell/rm_shell_model.py:414-421,
 * rm_shell_pde.py:194-209; dynamic_rm_shell/plate_sim.py:452-468).  Jacobi-preconditioned conjugate gradients with the matrix
 * applied cell by cell on the device; A is symmetric, so the same call serves the reverse mode.  Status 4: maxit reached short of
 * rtol (option "strict"). */
int femo_force_to_pressure(femo_ctx* ctx, const double* force, double* pressure, double rtol, int32_t maxit, int32_t* iters, double* relres);

/* ---- Dynamic shell (reference femo_alpha/dynamic_rm_shell/plate_sim.py:131-140,190-215): building blocks on
 * device vectors.  The time loop and its adjoint live in femo_alpha_amd/dynamic_rm_shell (Python, like the
 * reference's PlateSim); these calls supply the operator A = aK K + aM M of one midpoint/Newmark step, its
 * factorisation and solves, the reduced strain quadrature, and the thickness-gradient pieces. */
int femo_set_operator(femo_ctx* ctx, double aK, double aM);          /* operator used by solves and femo_factorize */
int femo_set_strain_quadrature(femo_ctx* ctx, int32_t nred);         /* nred x nred Gauss for membrane/bending/shear; 0 = full rule */
/* The rule of the static forms after creation -- same meaning as femo_create's nquad (ShellElement.getQuadratureRule with given
 * degrees, linear_shell_model.py:88-103).  Factor and Jacobi diagonal of the old rule are dropped.  femo_get_quadrature: the rule in
 * use and its number of points per cell. */
int femo_set_quadrature(femo_ctx* ctx, int32_t nquad);
int femo_get_quadrature(femo_ctx* ctx, int32_t* nquad, int32_t* npoints);
/* Host only, no device needed: the quadrature and shape tables of a rule exactly as the kernels receive them (what basix tabulates for
 * the reference, linear_shell_model.py:47-103), so that a caller or a test can compare them bit for bit with its own.  Arrays hold 36
 * points: w, wS [36]; N2 [36][9]; dN2 [36][9][2]; N1, NR [36][4]; dN1, dNR [36][4][2]; any pointer may be NULL. */
int femo_quadrature_tables(int32_t nvc, int32_t nquad, int32_t nred, int32_t cg1, int32_t cr, int32_t* npoints, double* w, double* wS,
                           double* N2, double* dN2, double* N1, double* dN1, double* NR, double* dNR);
int femo_op_apply_vec2(femo_ctx* ctx, int32_t src, int32_t dst, double aK, double aM, int with_penalty);
int femo_solve_vec(femo_ctx* ctx, int32_t b, int32_t x, int zero_guess, int32_t* iters, double* relres);
int femo_vec_mask_zero(femo_ctx* ctx, int32_t id);
int femo_grad_reset(femo_ctx* ctx);                                   /* thickness-gradient accumulator := 0 */
int femo_grad_add(femo_ctx* ctx, int kind, int32_t x, int32_t y, double scale);   /* += scale y^T dK/dh x (0) or y^T dM/dh x (1) */
int femo_grad_get(femo_ctx* ctx, double* out, int64_t n);

/* ---- CSR assembly of the elastic stiffness matrix (what assembleMatrix(dR_du) hands back in the reference,
 * csdl_alpha_opt/state_operation.py:289, fea/utils_dolfinx.py:200-206).  The solver never needs it (the operator is
 * matrix-free); it exists for callers that want the matrix.  The map -- the nel*ldof^2 element contributions sorted by CSR
 * destination -- and the pattern are built on the device by femo_build_csr_map (once per mesh; what dolfinx create_matrix does
 * per form); femo_set_csr_map accepts a map built elsewhere (femo_alpha_amd/csr.py, the host cross-check).
 * ms2 = { element matrices ms, scatter ms }. */
int femo_build_csr_map(femo_ctx* ctx, int32_t* nnz);
int femo_get_csr_pattern(femo_ctx* ctx, int32_t* rowptr, int32_t* colidx);
int femo_set_csr_map(femo_ctx* ctx, int32_t nnz, int64_t ncontrib, const int32_t* perm, const int32_t* dest);
int femo_assemble_csr(femo_ctx* ctx, double* vals, double* ms2);

/* Stress aggregation parameters (m, rho) of pnorm_stress = 1/alpha int (m vm_top)^rho J dx
 * (rm_shell/rm_shell_pde.py:112-128; defaults 1e-6, 100 as rm_shell_model.py:63). */
int femo_set_stress_params(femo_ctx* ctx, double m, double rho);
/* alpha of the aggregate given by the caller (pnorm_stress(alpha=...), rm_shell_pde.py:123-127) for the whole mesh (sel = -1) or a
 * sub-domain; alpha <= 0: back to the reference area evaluated at first use. */
int femo_set_stress_alpha(femo_ctx* ctx, int32_t sel, double alpha);
/* Sub-domains for the stress aggregate -- the reference's mesh tags / dxx(i) measure
 * (rm_shell/rm_shell_model.py:101-133, 242-253).  tags[nel] holds the sub-domain index of every cell
 * (0 .. ntags-1, or -1 for none).  femo_select_subdomain(sel) restricts "pnorm_stress" and its derivatives to the
 * cells of sub-domain sel, normalised by that sub-domain's reference area; sel = -1 selects the whole mesh again. */
int femo_set_cell_tags(femo_ctx* ctx, const int32_t* tags, int64_t n, int32_t ntags);
int femo_select_subdomain(femo_ctx* ctx, int32_t sel);
/* Field outputs "stress" (top surface, xi2 = h/2), "stress_mid" (xi2 = 0), "stress_bot" (xi2 = -h/2): von Mises stress
 * (RMShellPDE.von_Mises_stress(surface=...), rm_shell/rm_shell_pde.py:153-165) L2-projected onto DG1, nvc*nel values
 * (cell-major, the cell's vertices in connectivity order) -- replaces FEA.projectFieldOutput (fea/fea_dolfinx.py:205-206,
 * csdl_alpha_opt/output_operation.py:116-123). */
int femo_field_output(femo_ctx* ctx, const char* name, double* out, int64_t n);
/* Scalar outputs for the stored state and fields: "compliance", "mass", "elastic_energy", "pnorm_stress", "volume",
 * "regularization" (the thickness term of the compliance, rm_shell_pde.py:64-83), and over the selected sub-domain
 * (femo_select_subdomain; the whole mesh if none) "tip_disp" = 0.5 int u.u J, "area" = int J (rm_shell_pde.py:95-105) and
 * "sum_stress_x|y|z|xy|xz|yz" = int sigma_ij J dx of the top-surface in-plane stress (sum_stress_subdomain, :130-150) --
 * replaces assemble_scalar(form(c)) (csdl_alpha_opt/output_operation.py:51-56; forms at
 * rm_shell/rm_shell_pde.py:64-110). */
int femo_functional(femo_ctx* ctx, const char* name, double* value);
/* Gradient vector of a scalar output with respect to "disp_solid", "thickness", "density", "E", "nu",
 * "F_solid" or "uhat" (zeros where the form does not depend on the argument; "uhat" = shape sensitivity
 * through F = I + grad(uhat), kinematics.py:12-44) -- replaces assemble(derivative(form, arg), dim=1)
 * (csdl_alpha_opt/output_operation.py:58-69). n must equal the argument's length. */
int femo_dfunctional(femo_ctx* ctx, const char* name, const char* wrt, double* out, int64_t n);
/* out = (dR/d arg)^T lambda at the stored state, arg in "thickness","E","nu","F_solid","uhat" --
 * replaces assembleMatrix(dR/d arg) + computeMatVecProductBwd
 * (csdl_alpha_opt/state_operation.py:174-184,283-286; fea/utils_dolfinx.py:294-306). */
int femo_dRdarg_T(femo_ctx* ctx, const char* arg, const double* lambda, double* out, int64_t n);

/* The whole adjoint chain on the device, no host round trips:
 *   lambda = K^-1 dJ/dw ;  out = dJ/d arg - (dR/d arg)^T lambda      (total derivative)
 * for J = "compliance" | "elastic_energy" | "mass" and arg = "thickness" | "E" | "nu" | "F_solid".
 * This is what CSDL's compute_totals assembles from OutputOperation.compute_derivatives,
 * StateOperation.apply_inverse_jacobian('rev') and compute_jacvec_product('rev')
 * (SURVEY.md section 3.3). */
int femo_total_gradient(femo_ctx* ctx, const char* functional, const char* arg, double* out, int64_t n,
                        int32_t* iters, double* relres);
/* The same for SEVERAL functionals of the state and one argument: the reference registers compliance, elastic_energy, pnorm_stress and
 * one pnorm_stress_<tag> per sub-domain on `disp_solid` (rm_shell_model.py:221-253) and the simulator solves one adjoint per output
 * (state_operation.py:188-220); here the adjoint right-hand sides dJ_i/dw are formed together and solved by femo_solve_linear_multi's
 * grouped sweeps.  subdomains[i]: the tagged sub-domain functional i is restricted to (femo_set_cell_tags; -1, or subdomains == NULL:
 * the whole mesh).  out: nfun x n, row i = d J_i / d arg; iters, relres: nfun entries or NULL. */
int femo_total_gradients(femo_ctx* ctx, int32_t nfun, const char* const* functionals, const int32_t* subdomains, const char* arg, double* out,
                         int64_t n, int32_t* iters, double* relres);

/* Timing of the most recent solve, measured with HIP events on the context's stream (ms):
 * [0] setup (diagonal / preconditioner), [1] Krylov loop (with option "stale_factor": a factorisation inside the loop included),
 * [2] total, [3] the multifrontal factor that preconditioned the solve: 0 the current operator's, 1 one kept from an earlier
 * design (option "stale_factor"), 2 a kept one that was refreshed inside the solve; [4] number of element-operator launches. */
int femo_last_timing(const femo_ctx* ctx, double* out5);

/* Average duration (ms) of `reps` back-to-back launches of one kernel, timed with HIP events on
 * the context's stream: "apply" (matrix-free element operator), "pcg_update", "pcg_direction", "diag". */
int femo_bench_kernel(femo_ctx* ctx, const char* name, int32_t reps, double* avg_ms);

/* ---- Building blocks of the element-partitioned multi-GPU driver (femo_alpha_amd/parallel.py).  The
 * reference has no multi-rank path (its meshes live on MPI.COMM_SELF, fea/utils_dolfinx.py:41); these
 * calls expose the local pieces between which the driver places its collectives.
 * Vector ids: 0 state, 1 adjoint, 2 r, 3 z, 4 p, 5 Ap, 6 b. */
void* femo_vec_ptr(femo_ctx* ctx, int32_t id);                 /* device pointer, femo_ndof doubles */
int femo_sync(femo_ctx* ctx);                                  /* wait for the context's stream */
void* femo_stream_ptr(femo_ctx* ctx);                          /* the context's hipStream_t: the multi-GPU driver makes it torch's
                                                                  current stream, so that its tensor ops and collectives are
                                                                  ordered with the library's launches without device-wide syncs */
int femo_op_apply_vec(femo_ctx* ctx, int32_t src, int32_t dst);   /* dst = K_local src (no Dirichlet mask) */
int femo_load_vec(femo_ctx* ctx, int32_t dst);                 /* dst = local load vector */
int femo_factorize_range(femo_ctx* ctx, int32_t l0, int32_t l1, int assemble);   /* tree levels [l0, l1) */
int femo_frontal_sweep(femo_ctx* ctx, int32_t vec, int32_t l0, int32_t l1, int backward);
int femo_front_schur_get(femo_ctx* ctx, int32_t front, void* dst_dev, int64_t capacity_doubles);
int femo_front_block_set(femo_ctx* ctx, int32_t front, const void* src_dev);
int femo_functionals_partial(femo_ctx* ctx, double* out3);     /* { int u.u J dx, regularisation, mass } */
int femo_dfunctional_vec(femo_ctx* ctx, const char* name, int32_t dst);
int femo_field_gradient_vec(femo_ctx* ctx, const char* functional, const char* arg, int32_t lam, double* out, int64_t n);

/* The partitioned PCG itself (the counterpart of the reference's KSP solve, fea/utils_dolfinx.py:501-531, on an element
 * partition the reference does not have).  The replicated separator entries ("top", local indices top_idx, the same global
 * order on every rank), the weights of the global dot product (1 / nranks on replicated entries) and the scatter map of the
 * gradient (sel: position of this rank's field entries in the global field) stay resident in the context.  Every call
 * below only ENQUEUES work on the context's stream; the caller issues the collective named in the comment on the same
 * stream (torch.distributed over RCCL) and is the only party that talks to other ranks.
 *   buffer 0 = ntop + 1 doubles: packed replicated entries + one scalar that rides along;   buffer 1 = the 8 device
 *   scalars, [1] = r.z. */
int femo_dist_setup(femo_ctx* ctx, int32_t ntop, const int32_t* top_idx, int32_t nranks, int32_t n_local_levels,
                    int32_t nsel, const int32_t* sel);
void* femo_dist_ptr(femo_ctx* ctx, int32_t which);
int femo_dist_pack(femo_ctx* ctx, int32_t vec);                /* buffer 0 = vec[top]            -> all-reduce(buffer 0[0..ntop)) */
int femo_dist_unpack(femo_ctx* ctx, int32_t vec);              /* vec[top] = buffer 0 */
int femo_dist_pcg_start(femo_ctx* ctx, int32_t b, int32_t x);  /* x = 0, r = b, share of b.b */
int femo_dist_precond_fwd(femo_ctx* ctx);                      /* z = r, local forward sweep      -> all-reduce(buffer 0[0..ntop]) */
int femo_dist_read(femo_ctx* ctx, double* out2);               /* host: { r.r, previous p.Ap } (the one synchronisation per iteration) */
int femo_dist_precond_rest(femo_ctx* ctx);                     /* top sweeps, local backward sweep -> all-reduce(buffer 1[1]) */
int femo_dist_direction_apply(femo_ctx* ctx, int first);       /* p, Ap = A_local p               -> all-reduce(buffer 0[0..ntop]) */
int femo_dist_update(femo_ctx* ctx, int32_t x);                /* x += alpha p, r -= alpha Ap */
/* gglob[sel] = d functional / d arg - (dR/d arg)^T lambda(vector id) over this rank's cells; gglob: device, zeroed by the
 * caller, summed over the ranks by the caller                                                      -> all-reduce(gglob) */
int femo_dist_gradient(femo_ctx* ctx, const char* functional, const char* arg, int32_t lam, void* gglob_dev, int64_t nglob);
/* packed lower triangle (column by column) of a front's Schur complement, and back into a pivot-free stand-in front: what
 * the all-gather of the subtree roots carries.  Asynchronous on the context's stream. */
int femo_front_schur_pack(femo_ctx* ctx, int32_t front, void* dst_dev, int64_t capacity_doubles);
int femo_front_block_unpack(femo_ctx* ctx, int32_t front, const void* src_dev);
/* per-class times / launches / flops / bytes of the factorisation launches since the last assembly while option
 * "profile" is on (same layout as femo_factorize_profile): the partitioned driver factorises in two ranges */
int femo_factorize_profile_get(femo_ctx* ctx, double* out32);

/* ---- Transient march (BASELINE config 5) -- replaces PlateSim.solve_dynamic_problem (dynamic_rm_shell/plate_sim.py:281-361:
 * per step update_f + solveNonlinear_mod + the velocity update of :243-244,333) and the backward sweep of the dynamic
 * StateOperation (state_operation_dynamic.py:406-427, 619-691).  Displacement history, pressure history, velocity and the
 * adjoint history stay resident in HBM; one call marches all steps.  Time discretisation of plate_sim.py:131-140:
 *   (a M + K/2) w_i = F_i + M (a w_{i-1} + b wdot_{i-1}) - K/2 w_{i-1},   a = 2/dt^2, b = 2/dt,   force at the new level.
 * Histories cross the boundary level-major: (time_levels x femo_ndof) row-major. */
int femo_newmark_setup(femo_ctx* ctx, int32_t time_levels, double dt);          /* allocates; selects the operator K/2 + a M */
int femo_newmark_set_forces(femo_ctx* ctx, const double* f_history, int32_t levels_given);   /* (levels x F_solid length) */
int femo_newmark_set_constant_load(femo_ctx* ctx, const double* F);             /* ndof load vector added to every step, or NULL */
int femo_newmark_march(femo_ctx* ctx, int32_t nsteps, int reassemble_every_step, int32_t* iters, double* relres);
int femo_newmark_get_history(femo_ctx* ctx, int32_t which, double* out);        /* which: 0 displacements, 2 adjoint */
int femo_newmark_set_history(femo_ctx* ctx, int32_t which, const double* H);   /* which: 0 displacements, 2 adjoint */
int femo_newmark_adjoint(femo_ctx* ctx, const double* G, int32_t levels);       /* (dR/dy)^T Lambda = G, O(T) recursion */
int femo_newmark_residual_T(femo_ctx* ctx, int32_t levels, double* g_thickness, double* dF);
/* Forward mode (state_operation_dynamic.py:228-329: compute_jacvec_product fwd; :534-605: apply_inverse_jacobian fwd, the
 * "tangent linear model").  Results land in the adjoint-history buffer (femo_newmark_get_history(ctx, 2, ..)). */
int femo_newmark_jvp(femo_ctx* ctx, int32_t levels, const double* dY, const double* dthickness, const double* dF);
int femo_newmark_tangent(femo_ctx* ctx, const double* dR, int32_t levels);
/* Device addresses of the resident buffers: 0 displacement history, 1 velocity of the last level the march reached, 2 adjoint /
 * tangent history.  Lifetime: until the next femo_newmark_setup on this context (which re-allocates them) or femo_destroy; a
 * caller that wraps them (the zero-copy torch views of PlateSim) must drop its views before either. */
void* femo_newmark_ptr(femo_ctx* ctx, int32_t which);

/* Raw device pointer of a named buffer ("state","thickness","E","nu","density","F_solid","uhat")
 * for zero-copy wrapping by the caller (e.g. torch.from_dlpack-free ctypes views). */
void* femo_device_ptr(femo_ctx* ctx, const char* name);

#ifdef __cplusplus
}
#endif
#endif /* FEMO_HIP_H */
