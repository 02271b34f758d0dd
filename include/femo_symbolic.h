/* femo_symbolic.h -- C ABI of libfemo_symbolic.so: the analysis phase of the multifrontal Cholesky factorisation
 * (host code only; the numeric phase is libfemo_hip.so, femo_hip.h: femo_set_frontal_plan takes the arrays built here).
 *
 * Replaces the analysis MUMPS runs inside the reference's `setUpKSP_MUMPS` / `solveNonlinear` LU solves
 * (reference femo_alpha/fea/utils_dolfinx.py:466, 495-531), driven by the mesh instead of an assembled matrix:
 * nested dissection of the elements by coordinate bisection, elimination tree, front structure and index maps.
 *
 * Arrays are read back by name with femo_plan_size / femo_plan_itemsize / femo_plan_get:
 *   tree (one entry per tree node, ids in creation order, children > parent):
 *     "lo", "hi"        element interval [lo, hi) of the node in "eorder"        int32     (ids: depth by depth, left before right)
 *     "left", "right", "parent", "depth", "height"                               int32
 *   elements:  "eorder" (bisection order -> cell), "epos" (cell -> position)     int32
 *              "elem_front" (cell -> leaf front), "elem_map" (nel x ndpc: row of every element DOF in its leaf front)
 *   nodes:     "owner" (P2 node -> tree node that eliminates it)                  int32
 *              "piv_nodes" / "piv_off", "bnd_nodes" / "bnd_off": per tree node its own nodes and the ancestor-owned
 *              nodes its subtree touches, in the order of their rows in the front (node_order; CSR; offsets int64)
 *   fronts:    "npiv", "nf" (pivot DOFs, all DOFs), "dof_off" (int64), "front_dofs" (pivots first),
 *              "up_map" (row of every boundary DOF in the parent front, -1 on pivots)
 *   schedule:  "level_nodes" / "level_off": fronts by height, largest first inside a level
 */
#ifndef FEMO_SYMBOLIC_H
#define FEMO_SYMBOLIC_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct femo_plan femo_plan;

/* Build the plan.  cell_p2: nel x npc P2 node ids (row-major); cent: nel x 3 cell centroids; cell_dofs: nel x ndpc global
 * DOF numbers in element-local order (u of the P2 nodes, then theta of the vertices); P2 nodes < nV are vertices and carry
 * 6 DOFs (u: 3 n + c, theta: 3 nP2 + 3 n + c), the others 3.  leaf_size: cells per leaf; min_depth: every branch is split
 * at least that deep (2^d subtrees for d-level element partitions).  Returns 0, or an error code with
 * femo_plan_last_error() set. */
int femo_plan_build(femo_plan** out, int32_t nel, int32_t nP2, int32_t nV, int32_t npc, int32_t ndpc, const int32_t* cell_p2,
                    const double* cent, const int32_t* cell_dofs, int32_t leaf_size, int32_t min_depth);
/* The same with the two rules of the bisection exposed (femo_plan_build = axis_rule 0, gap_coeff 0: the plain median cut along the
 * longest centroid extent).
 *   gap_coeff > 0: a piece of n >= 128 cells is cut at the largest gap between consecutive sorted centroid coordinates within
 *     n/2 +- min(1/8, gap_coeff / sqrt(n)) n (about one row of cells either way), so that the cut follows a mesh line where the mesh
 *     has any row structure; the tree then has the fixed depth max(min_depth, ceil(log2(nel / leaf_size))).  leaf_size is then the
 *     AVERAGE number of cells per leaf, not a bound: every cut may sit up to the window's width off the middle and the imbalance compounds
 *     (leaves of up to ~1.5 leaf_size cells).  A piece that runs out of cells above that depth (fewer than two) stays a leaf there: levels
 *     are assigned by HEIGHT (a leaf is level 0 wherever it hangs), so the schedule is unaffected.
 *   axis_rule 1: cut across the axis along which the piece is longest in CELLS (centroid extent / mean cell extent; cext: nel x 3
 *     extents of the cells' bounding boxes), not in length units.
 *   axis_rule 2 (the package's default): a piece of n >= 16 cells is cut along every axis it extends in (in the order of rule 1's
 *     scores), the DOFs of the nodes that cells of both halves touch are counted, and the axis with the smallest separator is taken
 *     (the first of equal ones); smaller pieces follow rule 1. */
int femo_plan_build_ex(femo_plan** out, int32_t nel, int32_t nP2, int32_t nV, int32_t npc, int32_t ndpc, const int32_t* cell_p2,
                       const double* cent, const double* cext, const int32_t* cell_dofs, int32_t leaf_size, int32_t min_depth,
                       int32_t axis_rule, double gap_coeff);
/* The same with the order of the rows inside a front exposed (femo_plan_build_ex = node_order 0):
 *   node_order 0: pivot nodes and boundary nodes of a front in ascending node id (rounds 1-5);
 *   node_order 1 (the package's default): the nodes of a separator in the order in which they lie ALONG it (coordinate, along the axis of
 *     the separator's largest extent, of the mean centroid of the cells that touch the node), and the boundary nodes of a front grouped
 *     by owner -- nearest ancestor first -- in the owner's order.  A subtree touches a connected stretch of an ancestor's separator, so a
 *     child's Schur block lands in a few long runs of consecutive parent rows ("up_map" is increasing, piecewise contiguous) and the
 *     extend-add gathers of the numeric phase read contiguous segments.  Any order of the pivots inside a front is a valid elimination
 *     order: the factor differs by rounding only. */
int femo_plan_build_ex2(femo_plan** out, int32_t nel, int32_t nP2, int32_t nV, int32_t npc, int32_t ndpc, const int32_t* cell_p2,
                        const double* cent, const double* cext, const int32_t* cell_dofs, int32_t leaf_size, int32_t min_depth,
                        int32_t axis_rule, double gap_coeff, int32_t node_order);
/* number of entries of a named array (-1: no such array); 4 or 8 bytes per entry (0: no such array) */
int64_t femo_plan_size(const femo_plan* p, const char* name);
int femo_plan_itemsize(const femo_plan* p, const char* name);
/* copy a named array out; nbytes must be size * itemsize */
int femo_plan_get(const femo_plan* p, const char* name, void* dst, int64_t nbytes);
void femo_plan_free(femo_plan* p);
const char* femo_plan_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
