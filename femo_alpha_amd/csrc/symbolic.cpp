// Symbolic analysis of the multifrontal Cholesky factorisation -- host code, no GPU involved (include/femo_symbolic.h).
//
// The reference hands its assembled Jacobian to MUMPS through PETSc (reference femo_alpha/fea/utils_dolfinx.py:466,
// 495-531); the analysis phase of that solver (ordering + elimination tree + front structure) is what this file
// replaces, driven by the mesh instead of a sparse matrix:
//   1. nested dissection of the ELEMENTS by recursive coordinate bisection (stable sort of the centroids along one axis; cut in
//      the middle, or at the largest gap near the middle: femo_plan_build_ex) down to leaves of <= leaf_size cells;
//   2. every P2 node is eliminated at the deepest tree node whose element interval holds all its elements;
//   3. front of a tree node = the DOFs of its own nodes (pivots) + the DOFs of the ancestor-owned nodes its subtree
//      touches (boundary), children's boundaries merged upwards;
//   4. index maps: boundary row of a child -> row of its parent, element DOF -> row of its leaf front; levels by height.
// The host-side Python module femo_alpha_amd/solver/symbolic.py states the same algorithm in numpy and is kept as the
// cross-check (tests/test_symbolic_native.py compares every array).  The tree grows depth by depth; the sorts of one depth are
// independent and run in parallel (OpenMP).
#include <omp.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/femo_symbolic.h"

namespace {

struct Plan {
    std::map<std::string, std::vector<int32_t>> i32;
    std::map<std::string, std::vector<int64_t>> i64;
    std::string error;
};

thread_local std::string g_error;

template <class T>
std::vector<T>& put(std::map<std::string, std::vector<T>>& m, const char* name) { return m[name]; }

}   // namespace

struct femo_plan : Plan {};

extern "C" {

const char* femo_plan_last_error(void) { return g_error.c_str(); }

void femo_plan_free(femo_plan* p) { delete p; }

int femo_plan_build_ex(femo_plan** out, int32_t nel, int32_t nP2, int32_t nV, int32_t npc, int32_t ndpc, const int32_t* cell_p2,
                       const double* cent, const double* cext, const int32_t* cell_dofs, int32_t leaf_size, int32_t min_depth,
                       int32_t axis_rule, double gap_coeff) {
    return femo_plan_build_ex2(out, nel, nP2, nV, npc, ndpc, cell_p2, cent, cext, cell_dofs, leaf_size, min_depth, axis_rule, gap_coeff, 0);
}

int femo_plan_build_ex2(femo_plan** out, int32_t nel, int32_t nP2, int32_t nV, int32_t npc, int32_t ndpc, const int32_t* cell_p2,
                        const double* cent, const double* cext, const int32_t* cell_dofs, int32_t leaf_size, int32_t min_depth,
                        int32_t axis_rule, double gap_coeff, int32_t node_order) {
    if (!out || node_order < 0 || node_order > 1 || nel < 1 || nP2 < 1 || nV < 1 || npc < 1 || ndpc < 1 || !cell_p2 || !cent || !cell_dofs || leaf_size < 1 || min_depth < 0 ||
        axis_rule < 0 || axis_rule > 2 || (axis_rule >= 1 && !cext) || !(gap_coeff >= 0.0)) {
        g_error = "femo_plan_build: bad arguments";
        return 1;
    }
    // at most 16 threads: several ranks of a multi-GPU job run this at the same time on one host
    const int nthreads = std::max(1, std::min(omp_get_max_threads(), 16));
    const int64_t ndof_u = 3 * (int64_t)nP2;
    // ---- 1. the bisection tree, depth by depth (the sorts of one depth are independent and run in parallel; ids in creation
    //         order: the nodes of a depth in the order of their parents, left before right -- children > parent)
    //   where to cut (gap_coeff > 0): not at the middle element but at the LARGEST GAP between consecutive sorted centroid
    //   coordinates within  mid +- min(1/8, gap_coeff / sqrt(n)) n  (about one row of cells either way).  On meshes with any
    //   row structure the middle element sits inside a row, and the cut through that row is a zigzag that drags both
    //   neighbouring mesh lines into the separator (BASELINE config 3: separators of 1401-1455 DOFs where a mesh line has
    //   1050); a cut at a gap follows a mesh line.  Truly unstructured meshes have no such gaps and get a cut near the middle.
    //   The tree then has a FIXED DEPTH (the smallest one whose average leaf holds <= leaf_size cells): unequal halves must
    //   not put siblings on different levels of the schedule.
    //   along which axis (axis_rule 1): the one along which the piece is longest IN CELLS (centroid extent / mean cell extent),
    //   not in metres -- a tapered wing's cells are squeezed chordwise, and the shortest separator crosses the fewest cells.
    //   axis_rule 2: measure instead of guess -- pieces of at least AXIS_NMIN cells are sorted along EVERY axis they extend in, the
    //   separator (DOFs of the nodes that cells of both halves touch) is counted for every cut position of that order, the position
    //   with the smallest one inside the window of the gap rule is taken (of equal ones the nearest to the largest gap), and the
    //   axis with the smallest separator wins (first of equal ones in the order of rule 1's scores).  Rule 1 is misled by sheared pieces: on an unstructured triangulation
    //   of the swept, tapered skin the bounding box of a quarter wing is 155 "cells" wide and 145 long, and the cut the wrong way
    //   costs a separator of 2 058 DOFs where 1 050 do (factorisation 275 -> 232 GFLOP).  Smaller pieces keep rule 1.
    int32_t fixed_depth = 0;
    while (((int64_t)leaf_size << fixed_depth) < (int64_t)nel) ++fixed_depth;
    fixed_depth = std::max(fixed_depth, min_depth);
    const bool gap_mode = gap_coeff > 0.0;
    constexpr int32_t GAP_NMIN = 128;          // smaller pieces are halved exactly (a row is a large share of them)
    constexpr int32_t AXIS_NMIN = 16;          // axis_rule 2 measures the separators of pieces of at least this many cells (config 3: Schur
                                               // blocks 7.17 -> 7.02 GB against 128; below 16 nothing changes)
    std::vector<int32_t> lo{0}, hi{nel}, left{-1}, right{-1}, parent{-1}, depth{0};
    std::vector<int32_t> eorder(nel);
    std::iota(eorder.begin(), eorder.end(), 0);
    struct Scratch { std::vector<int32_t> stamp, first, last, touched; std::vector<long long> diff; };
    std::vector<Scratch> scratch(nthreads);        // axis_rule 2: per-thread node marks (a unique value per piece and axis), first / last cell of a node
    {
        std::vector<int32_t> frontier{0};
        while (!frontier.empty()) {
            std::vector<int32_t> split(frontier.size(), -1);        // position of the cut inside [lo, hi), or -1: a leaf
            int bad = 0;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1) reduction(| : bad)
            for (int64_t k = 0; k < (int64_t)frontier.size(); ++k) {
                const int32_t t = frontier[k], a = lo[t], b = hi[t], n = b - a;
                const bool want = gap_mode ? depth[t] < fixed_depth : (n > leaf_size || depth[t] < min_depth);
                if (!want) continue;
                // a piece of one cell cannot be cut: an error below the depth the caller asked for (2^d partitions), otherwise a leaf that
                // hangs higher in the tree -- levels are assigned by height (section 4 of the header comment), so the schedule does not mind
                if (n < 2) { if (depth[t] < min_depth) bad |= 1; continue; }
                double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300}, cs[3] = {0.0, 0.0, 0.0};
                for (int32_t i = a; i < b; ++i)
                    for (int c = 0; c < 3; ++c) {
                        const double v = cent[3 * (int64_t)eorder[i] + c];
                        mn[c] = std::min(mn[c], v); mx[c] = std::max(mx[c], v);
                        if (axis_rule >= 1) cs[c] += cext[3 * (int64_t)eorder[i] + c];        // sequential sum (the numpy twin: cumsum)
                    }
                double score[3];
                for (int c = 0; c < 3; ++c) {
                    score[c] = mx[c] - mn[c];
                    if (axis_rule >= 1) { const double mean = cs[c] / n; score[c] = mean > 0.0 ? score[c] / mean : 0.0; }
                }
                int ax = 0;
                for (int c = 1; c < 3; ++c)
                    if (score[c] > score[ax]) ax = c;                     // first of equal scores, as numpy's argmax
                // sort the piece along one axis (stable, from the order the piece arrived in) and place the cut
                auto sort_and_cut = [&](int axis, std::vector<int32_t>& ord) {
                    std::stable_sort(ord.begin(), ord.end(),
                                     [&](int32_t x, int32_t y) { return cent[3 * (int64_t)x + axis] < cent[3 * (int64_t)y + axis]; });
                    int32_t m = n / 2;
                    if (gap_mode && n >= GAP_NMIN) {
                        const int32_t w = std::max<int32_t>(1, (int32_t)(std::min(0.125, gap_coeff / std::sqrt((double)n)) * n));
                        const int32_t ka = std::max<int32_t>(1, m - w), kb = std::min<int32_t>(n - 1, m + w);
                        int32_t best = m;
                        double gbest = -1.0;
                        for (int32_t kk = ka; kk <= kb; ++kk) {
                            const double g = cent[3 * (int64_t)ord[kk] + axis] - cent[3 * (int64_t)ord[kk - 1] + axis];
                            // the largest gap; of equal gaps the one nearest the middle, then the lower one
                            if (g > gbest || (g == gbest && std::abs(kk - m) < std::abs(best - m))) { gbest = g; best = kk; }
                        }
                        m = best;
                    }
                    return m;
                };
                std::vector<int32_t> ord(eorder.begin() + a, eorder.begin() + b);
                int32_t mid;
                if (axis_rule == 2 && n >= AXIS_NMIN) {
                    // candidates in the order of rule 1's scores (descending, first of equal ones first); axes without extent are out
                    int cand[3] = {0, 1, 2};
                    std::stable_sort(cand, cand + 3, [&](int x, int y) { return score[x] > score[y]; });
                    Scratch& sc = scratch[omp_get_thread_num()];
                    if (sc.stamp.empty()) { sc.stamp.assign(nP2, -1); sc.first.resize(nP2); sc.last.resize(nP2); }
                    long long sep_best = -1;
                    std::vector<int32_t> ord_best;
                    int32_t mid_best = n / 2;
                    for (int q = 0; q < 3; ++q) {
                        const int c = cand[q];
                        if (!(mx[c] > mn[c])) continue;
                        std::vector<int32_t> oc(eorder.begin() + a, eorder.begin() + b);
                        const int32_t m0 = sort_and_cut(c, oc);
                        // sep(m) for EVERY cut position m of this order in one pass: a node lies in the separator of m iff the first cell
                        // that touches it sits before m and the last one at or after m -- a difference array over the positions
                        const int32_t sid = 4 * t + q;               // (int32: trees of up to 5e8 nodes, i.e. ~3e9 cells -- beyond one device by far)
                        sc.touched.clear();
                        for (int32_t i = 0; i < n; ++i)
                            for (int al = 0; al < npc; ++al) {
                                const int32_t nd = cell_p2[(int64_t)oc[i] * npc + al];
                                if (sc.stamp[nd] != sid) { sc.stamp[nd] = sid; sc.first[nd] = i; sc.touched.push_back(nd); }
                                sc.last[nd] = i;
                            }
                        sc.diff.assign((size_t)n + 2, 0);
                        for (const int32_t nd : sc.touched)
                            if (sc.last[nd] > sc.first[nd]) {
                                const long long w = nd < nV ? 6 : 3;
                                sc.diff[sc.first[nd] + 1] += w; sc.diff[sc.last[nd] + 1] -= w;
                            }
                        for (int32_t i = 1; i <= n; ++i) sc.diff[i] += sc.diff[i - 1];             // diff[m] = sep(m), 1 <= m <= n - 1
                        // the position: inside the window of the gap rule the smallest separator; of equal ones the nearest to the
                        // largest gap, then the lower one.  (On a mesh with rows the minima ARE the mesh lines; on an unstructured one
                        // the zigzag is shortest somewhere -- 216 -> 202 GFLOP on the unstructured skin.)
                        int32_t m = m0;
                        if (gap_mode && n >= GAP_NMIN) {
                            const int32_t w = std::max<int32_t>(1, (int32_t)(std::min(0.125, gap_coeff / std::sqrt((double)n)) * n));
                            const int32_t ka = std::max<int32_t>(1, n / 2 - w), kb = std::min<int32_t>(n - 1, n / 2 + w);
                            for (int32_t kk = ka; kk <= kb; ++kk)
                                if (sc.diff[kk] < sc.diff[m] || (sc.diff[kk] == sc.diff[m] && std::abs(kk - m0) < std::abs(m - m0))) m = kk;
                        }
                        const long long sep = sc.diff[m];
                        if (sep_best < 0 || sep < sep_best) { sep_best = sep; ord_best.swap(oc); mid_best = m; ax = c; }
                    }
                    if (sep_best < 0) {                  // no axis with an extent (coincident centroids): halve the piece as it stands
                        mid = n / 2;
                    } else {
                        ord.swap(ord_best);
                        mid = mid_best;
                    }
                } else {
                    mid = sort_and_cut(ax, ord);
                }
                std::copy(ord.begin(), ord.end(), eorder.begin() + a);
                split[k] = mid;
            }
            if (bad) { g_error = "mesh too small for the requested number of partitions"; return 2; }
            std::vector<int32_t> next;
            for (size_t k = 0; k < frontier.size(); ++k) {
                if (split[k] < 0) continue;
                const int32_t t = frontier[k], a = lo[t], b = hi[t], mid = a + split[k];
                for (int side = 0; side < 2; ++side) {
                    const int32_t id = (int32_t)lo.size();
                    lo.push_back(side ? mid : a); hi.push_back(side ? b : mid); left.push_back(-1); right.push_back(-1);
                    parent.push_back(t); depth.push_back(depth[t] + 1);
                    (side ? right : left)[t] = id;
                    next.push_back(id);
                }
            }
            frontier.swap(next);
        }
    }
    const int32_t ntree = (int32_t)lo.size();
    int32_t maxdepth = 0;
    for (int32_t t = 0; t < ntree; ++t) maxdepth = std::max(maxdepth, depth[t]);
    std::vector<int32_t> epos(nel);
    for (int32_t i = 0; i < nel; ++i) epos[eorder[i]] = i;
    // ---- 2. owners
    std::vector<int32_t> amin(nP2, nel), amax(nP2, -1);
    for (int32_t e = 0; e < nel; ++e)
        for (int a = 0; a < npc; ++a) {
            const int32_t n = cell_p2[(int64_t)e * npc + a];
            if (n < 0 || n >= nP2) { g_error = "cell_p2 entry out of range"; return 3; }
            amin[n] = std::min(amin[n], epos[e]); amax[n] = std::max(amax[n], epos[e]);
        }
    std::vector<int32_t> owner(nP2, 0);
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int32_t n = 0; n < nP2; ++n) {
        int32_t t = 0;
        while (left[t] >= 0) {
            const int32_t mid = lo[right[t]];
            if (amax[n] < mid) t = left[t];
            else if (amin[n] >= mid) t = right[t];
            else break;
        }
        owner[n] = t;
    }
    std::vector<int32_t> height(ntree, 0);
    for (int32_t t = ntree - 1; t >= 0; --t)                                    // children have larger ids than their parent
        if (left[t] >= 0) height[t] = 1 + std::max(height[left[t]], height[right[t]]);
    // pivot nodes per tree node: ascending node id
    std::vector<int64_t> piv_off(ntree + 1, 0);
    for (int32_t n = 0; n < nP2; ++n) ++piv_off[owner[n] + 1];
    for (int32_t t = 0; t < ntree; ++t) piv_off[t + 1] += piv_off[t];
    std::vector<int32_t> piv_nodes(nP2);
    {
        std::vector<int64_t> fill(piv_off.begin(), piv_off.end() - 1);
        for (int32_t n = 0; n < nP2; ++n) piv_nodes[fill[owner[n]]++] = n;
    }
    // node_order 1: the nodes of a separator in the order in which they lie ALONG it, and every boundary list grouped by owner
    // (nearest ancestor first) in the owner's order.  A subtree touches a connected stretch of an ancestor's separator, so the rows a
    // child's Schur block contributes to are then a few long RUNS of consecutive rows of the parent front (one per owner group) instead
    // of node-sized snippets interleaved with the sibling's -- the gathering rank-k updates and the extend-add read their child entries
    // as contiguous segments (scripts/r6_cinv_runs.py: median run 6-12 entries on levels 1-5 with the ascending-id order).  Position
    // along a separator: the coordinate, along the axis of the largest extent of the separator's nodes, of the mean centroid of the
    // cells that touch the node (no node coordinates are handed to this library); ties by node id.
    std::vector<int32_t> rank;
    if (node_order == 1) {
        std::vector<double> nc(3 * (size_t)nP2, 0.0);
        std::vector<int32_t> cnt(nP2, 0);
        for (int32_t e = 0; e < nel; ++e)                                  // sequential sums in (cell, local node) order: the numpy twin adds in the same order
            for (int a = 0; a < npc; ++a) {
                const int32_t n = cell_p2[(int64_t)e * npc + a];
                for (int c = 0; c < 3; ++c) nc[3 * (size_t)n + c] += cent[3 * (int64_t)e + c];
                ++cnt[n];
            }
        for (int32_t n = 0; n < nP2; ++n)
            for (int c = 0; c < 3; ++c) nc[3 * (size_t)n + c] /= (double)std::max(cnt[n], 1);
        rank.assign(nP2, 0);
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 64)
        for (int32_t t = 0; t < ntree; ++t) {
            int32_t* b = piv_nodes.data() + piv_off[t];
            const int64_t n = piv_off[t + 1] - piv_off[t];
            if (n > 1) {
                double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
                for (int64_t i = 0; i < n; ++i)
                    for (int c = 0; c < 3; ++c) { const double v = nc[3 * (size_t)b[i] + c]; mn[c] = std::min(mn[c], v); mx[c] = std::max(mx[c], v); }
                int ax = 0;
                for (int c = 1; c < 3; ++c)
                    if (mx[c] - mn[c] > mx[ax] - mn[ax]) ax = c;                 // first of equal extents, as numpy's argmax
                std::stable_sort(b, b + n, [&](int32_t x, int32_t y) { return nc[3 * (size_t)x + ax] < nc[3 * (size_t)y + ax]; });
            }
            for (int64_t i = 0; i < n; ++i) rank[b[i]] = (int32_t)i;
        }
    }
    // ---- 3. boundary nodes, bottom-up (sorted ascending; union of the children's lists minus the node's own)
    std::vector<std::vector<int32_t>> bnd(ntree);
    {
        std::vector<std::vector<int32_t>> by_depth(maxdepth + 1);
        for (int32_t t = 0; t < ntree; ++t) by_depth[depth[t]].push_back(t);
        for (int32_t d = maxdepth; d >= 0; --d) {
            const auto& nodes = by_depth[d];
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 16)
            for (int64_t k = 0; k < (int64_t)nodes.size(); ++k) {
                const int32_t t = nodes[k];
                std::vector<int32_t> u;
                if (left[t] < 0) {
                    for (int32_t i = lo[t]; i < hi[t]; ++i)
                        for (int a = 0; a < npc; ++a) u.push_back(cell_p2[(int64_t)eorder[i] * npc + a]);
                    std::sort(u.begin(), u.end());
                    u.erase(std::unique(u.begin(), u.end()), u.end());
                } else {
                    const auto &x = bnd[left[t]], &y = bnd[right[t]];
                    u.resize(x.size() + y.size());
                    u.erase(std::set_union(x.begin(), x.end(), y.begin(), y.end(), u.begin()), u.end());
                }
                auto& b = bnd[t];
                b.reserve(u.size());
                for (int32_t n : u)
                    if (owner[n] != t) b.push_back(n);
            }
        }
    }
    if (node_order == 1) {
        // (the unions above need the lists in ascending id; the order of the rows is settled here, once every list is complete)
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 64)
        for (int32_t t = 0; t < ntree; ++t)
            std::sort(bnd[t].begin(), bnd[t].end(), [&](int32_t x, int32_t y) {
                const int32_t dx = depth[owner[x]], dy = depth[owner[y]];       // owners are ancestors of t: one per depth
                return dx != dy ? dx > dy : rank[x] < rank[y];
            });
    }
    std::vector<int64_t> bnd_off(ntree + 1, 0);
    for (int32_t t = 0; t < ntree; ++t) bnd_off[t + 1] = bnd_off[t] + (int64_t)bnd[t].size();
    std::vector<int32_t> bnd_nodes(bnd_off[ntree]);
    for (int32_t t = 0; t < ntree; ++t) std::copy(bnd[t].begin(), bnd[t].end(), bnd_nodes.begin() + bnd_off[t]);
    // ---- front DOF lists: pivots first, then the boundary; a vertex node carries u and theta (6 DOFs), the others u (3)
    auto ndofs_of = [&](int32_t n) { return n < nV ? 6 : 3; };
    std::vector<int32_t> npiv(ntree), nf(ntree);
    std::vector<int64_t> dof_off(ntree + 1, 0);
    for (int32_t t = 0; t < ntree; ++t) {
        int64_t a = 0, b = 0;
        for (int64_t i = piv_off[t]; i < piv_off[t + 1]; ++i) a += ndofs_of(piv_nodes[i]);
        for (int32_t n : bnd[t]) b += ndofs_of(n);
        if (a + b > INT32_MAX) { g_error = "front too large"; return 4; }
        npiv[t] = (int32_t)a; nf[t] = (int32_t)(a + b);
        dof_off[t + 1] = dof_off[t] + a + b;
    }
    if (ndof_u + 3 * (int64_t)nV > INT32_MAX) { g_error = "more than 2^31 DOFs"; return 4; }
    std::vector<int32_t> front_dofs(dof_off[ntree]);
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 64)
    for (int32_t t = 0; t < ntree; ++t) {
        int64_t w = dof_off[t];
        auto emit = [&](int32_t n) {
            for (int c = 0; c < 3; ++c) front_dofs[w++] = 3 * n + c;
            if (n < nV)
                for (int c = 0; c < 3; ++c) front_dofs[w++] = (int32_t)(ndof_u + 3 * (int64_t)n + c);
        };
        for (int64_t i = piv_off[t]; i < piv_off[t + 1]; ++i) emit(piv_nodes[i]);
        for (int32_t n : bnd[t]) emit(n);
    }
    // ---- 4. index maps.  Position of a DOF inside a front: sorted copy of the front's list + binary search
    //         (fronts are independent: parallel over parents / leaves)
    std::vector<int32_t> up_map(dof_off[ntree], -1);
    std::vector<int32_t> leaf_of_pos(nel, 0);
    for (int32_t t = 0; t < ntree; ++t)
        if (left[t] < 0)
            for (int32_t i = lo[t]; i < hi[t]; ++i) leaf_of_pos[i] = t;
    std::vector<int32_t> elem_front(nel);
    for (int32_t e = 0; e < nel; ++e) elem_front[e] = leaf_of_pos[epos[e]];
    std::vector<int32_t> elem_map((int64_t)nel * ndpc);
    int bad = 0;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 16) reduction(| : bad)
    for (int32_t t = 0; t < ntree; ++t) {
        const int32_t n = nf[t];
        const int32_t* fd = front_dofs.data() + dof_off[t];
        std::vector<std::pair<int32_t, int32_t>> srt(n);
        for (int32_t i = 0; i < n; ++i) srt[i] = {fd[i], i};
        std::sort(srt.begin(), srt.end());
        auto find = [&](int32_t dof) {
            auto it = std::lower_bound(srt.begin(), srt.end(), std::make_pair(dof, (int32_t)-1));
            if (it == srt.end() || it->first != dof) { bad |= 1; return -1; }
            return it->second;
        };
        if (left[t] >= 0) {
            for (int32_t ch : {left[t], right[t]})
                for (int64_t i = dof_off[ch] + npiv[ch]; i < dof_off[ch + 1]; ++i) up_map[i] = find(front_dofs[i]);
        } else {
            for (int32_t i = lo[t]; i < hi[t]; ++i) {
                const int32_t e = eorder[i];
                for (int k = 0; k < ndpc; ++k) elem_map[(int64_t)e * ndpc + k] = find(cell_dofs[(int64_t)e * ndpc + k]);
            }
        }
    }
    if (bad) { g_error = "a boundary or element DOF is missing from the front that should hold it"; return 5; }
    // levels by height; inside a level the largest fronts first (stable)
    int32_t nlevels = 0;
    for (int32_t t = 0; t < ntree; ++t) nlevels = std::max(nlevels, height[t] + 1);
    std::vector<int64_t> level_off(nlevels + 1, 0);
    for (int32_t t = 0; t < ntree; ++t) ++level_off[height[t] + 1];
    for (int32_t h = 0; h < nlevels; ++h) level_off[h + 1] += level_off[h];
    std::vector<int32_t> level_nodes(ntree);
    {
        std::vector<int64_t> fill(level_off.begin(), level_off.end() - 1);
        for (int32_t t = 0; t < ntree; ++t) level_nodes[fill[height[t]]++] = t;
        for (int32_t h = 0; h < nlevels; ++h)
            std::stable_sort(level_nodes.begin() + level_off[h], level_nodes.begin() + level_off[h + 1],
                             [&](int32_t a, int32_t b) { return nf[a] > nf[b]; });
    }
    femo_plan* p = new femo_plan;
    p->i32["lo"] = std::move(lo); p->i32["hi"] = std::move(hi); p->i32["left"] = std::move(left); p->i32["right"] = std::move(right);
    p->i32["parent"] = std::move(parent); p->i32["depth"] = std::move(depth); p->i32["height"] = std::move(height);
    p->i32["eorder"] = std::move(eorder); p->i32["epos"] = std::move(epos); p->i32["owner"] = std::move(owner);
    p->i32["piv_nodes"] = std::move(piv_nodes); p->i64["piv_off"] = std::move(piv_off);
    p->i32["bnd_nodes"] = std::move(bnd_nodes); p->i64["bnd_off"] = std::move(bnd_off);
    p->i32["npiv"] = std::move(npiv); p->i32["nf"] = std::move(nf); p->i64["dof_off"] = std::move(dof_off);
    p->i32["front_dofs"] = std::move(front_dofs); p->i32["up_map"] = std::move(up_map);
    p->i32["elem_front"] = std::move(elem_front); p->i32["elem_map"] = std::move(elem_map);
    p->i32["level_nodes"] = std::move(level_nodes); p->i64["level_off"] = std::move(level_off);
    *out = p;
    return 0;
}

int femo_plan_build(femo_plan** out, int32_t nel, int32_t nP2, int32_t nV, int32_t npc, int32_t ndpc, const int32_t* cell_p2,
                    const double* cent, const int32_t* cell_dofs, int32_t leaf_size, int32_t min_depth) {
    return femo_plan_build_ex(out, nel, nP2, nV, npc, ndpc, cell_p2, cent, nullptr, cell_dofs, leaf_size, min_depth, 0, 0.0);
}

int64_t femo_plan_size(const femo_plan* p, const char* name) {
    if (!p || !name) return -1;
    auto a = p->i32.find(name);
    if (a != p->i32.end()) return (int64_t)a->second.size();
    auto b = p->i64.find(name);
    if (b != p->i64.end()) return (int64_t)b->second.size();
    return -1;
}

int femo_plan_itemsize(const femo_plan* p, const char* name) {
    if (!p || !name) return 0;
    if (p->i32.count(name)) return 4;
    if (p->i64.count(name)) return 8;
    return 0;
}

int femo_plan_get(const femo_plan* p, const char* name, void* dst, int64_t nbytes) {
    if (!p || !name || !dst) { g_error = "femo_plan_get: bad arguments"; return 1; }
    auto a = p->i32.find(name);
    if (a != p->i32.end()) {
        if (nbytes != (int64_t)a->second.size() * 4) { g_error = std::string("femo_plan_get: wrong size for ") + name; return 2; }
        std::memcpy(dst, a->second.data(), nbytes);
        return 0;
    }
    auto b = p->i64.find(name);
    if (b != p->i64.end()) {
        if (nbytes != (int64_t)b->second.size() * 8) { g_error = std::string("femo_plan_get: wrong size for ") + name; return 2; }
        std::memcpy(dst, b->second.data(), nbytes);
        return 0;
    }
    g_error = std::string("femo_plan_get: no array named ") + name;
    return 3;
}

}   // extern "C"
