"""Symbolic analysis for the multifrontal Cholesky preconditioner (host side, mesh-only).

The reference factorises the assembled Jacobian with MUMPS through PETSc
(reference femo_alpha/fea/utils_dolfinx.py:466,495-531) -- a multifrontal sparse direct solver.
This module is the build's own, mesh-driven version of the analysis phase: a geometric nested
dissection of the *elements* (recursive coordinate bisection), mesh nodes assigned to the tree
node at which their elements part ways (leaf interiors / separators), and for every tree node
the dense *front* = its own (pivot) DOFs + the ancestor DOFs its subtree touches.  Element
matrices are summed straight into the leaf fronts (the original "frontal" idea), so no global
sparse matrix is ever built.

Everything here depends on the mesh only; it is computed once per context and uploaded to HBM.

The analysis runs in host C++ (``csrc/symbolic.cpp`` -> libfemo_symbolic.so, C ABI ``include/femo_symbolic.h``):
``analyse`` and ``build_plan`` call it.  The numpy statement of the same algorithm below (``impl="python"``) is the
cross-check the tests compare it with, array by array; at 1 M DOF it takes 2 s where the C++ takes 0.1 s.

Multi-GPU: the 2^d subtrees at depth d of the same tree are the element
partition; ``rank_plan`` cuts out, for one rank, its subtree, the replicated top of the tree and
pivot-free stand-ins for the other ranks' subtree roots, in the numbering of the rank's sub-mesh.
"""
from __future__ import annotations

import numpy as np

from ..mesh import ShellMesh

__all__ = ["FrontalPlan", "Tree", "analyse", "build_plan", "rank_plan"]


class FrontalPlan:
    """Flat arrays describing the elimination tree and all index maps (see ``build_plan``)."""

    def summary(self):
        nf, npv = self.nf, self.npiv
        flops = float(np.sum(npv.astype(np.float64) * nf.astype(np.float64) ** 2
                             - npv.astype(np.float64) ** 2 * nf + npv.astype(np.float64) ** 3 / 3.0))
        return dict(fronts=int(self.ntree), levels=int(self.nlevels), leaves=int(self.nleaves),
                    max_front=int(nf.max()), max_pivots=int(npv.max()),
                    front_doubles=int(self.front_off[-1]), front_GB=float(self.front_off[-1] * 8 / 1e9),
                    factor_doubles=int(np.sum(nf.astype(np.int64) * npv)), factor_gflop=flops / 1e9)


class Tree:
    """Nested-dissection tree over the elements with, per tree node, the P2 nodes it eliminates
    (``piv_nodes``) and the ancestor-owned P2 nodes its subtree touches (``bnd_nodes``)."""


def _node_dofs(nodes, nV, ndof_u, rot=None):
    """DOF ids of P2 nodes (the nodes that carry the rotation have u and theta: 6 DOFs, the others 3), concatenated in order.
    The rotation lives on the vertices (nodes < nV, theta of vertex v at ndof_u + 3 v + c) -- or, ``rot = (first, count)``: CG2CR1,
    on the edge midpoints, the P2 nodes first .. first + count - 1 (theta of edge k at ndof_u + 3 k + c)."""
    nodes = np.asarray(nodes, dtype=np.int64)
    if rot is not None:
        r0, nr = rot
        is_r = (nodes >= r0) & (nodes < r0 + nr)
        cnt = np.where(is_r, 6, 3)
        off = np.concatenate([[0], np.cumsum(cnt)])
        out = np.empty(off[-1], dtype=np.int64)
        base = off[:-1]
        for c in range(3):
            out[base + c] = 3 * nodes + c
        rb, rn = base[is_r], nodes[is_r] - r0
        for c in range(3):
            out[rb + 3 + c] = ndof_u + 3 * rn + c
        return out
    is_v = nodes < nV
    cnt = np.where(is_v, 6, 3)
    off = np.concatenate([[0], np.cumsum(cnt)])
    out = np.empty(off[-1], dtype=np.int64)
    base = off[:-1]
    for c in range(3):
        out[base + c] = 3 * nodes + c
    vb = base[is_v]
    vn = nodes[is_v]
    for c in range(3):
        out[vb + 3 + c] = ndof_u + 3 * vn + c
    return out


def _tree_from_native(A) -> Tree:
    T = Tree()
    for k in ("lo", "hi", "left", "right", "parent", "depth", "height", "eorder", "epos", "owner"):
        setattr(T, k, A[k].astype(np.int64))
    T.ntree = T.lo.size
    po, bo = A["piv_off"], A["bnd_off"]
    pn, bn = A["piv_nodes"].astype(np.int64), A["bnd_nodes"].astype(np.int64)
    T.piv_nodes = [pn[po[t]:po[t + 1]] for t in range(T.ntree)]
    T.bnd_nodes = [bn[bo[t]:bo[t + 1]] for t in range(T.ntree)]
    return T


# Rules of the bisection (csrc/symbolic.cpp, femo_plan_build_ex).  AXIS_RULE 1: cut across the axis along which a piece is longest in
# cells; GAP > 0: cut at the largest gap of the sorted centroid coordinates within about one row of cells of the middle, fixed tree
# depth.  AXIS_RULE 2: pieces of >= AXIS_NMIN cells are cut along every axis and the smallest separator wins (sheared pieces of
# unstructured meshes mislead rule 1: 275 -> 218 GFLOP on the unstructured skin; config 3: 212 -> 204).  (0, 0.0) is the plain median cut of rounds 1-3.  At BASELINE config 3 the pair below takes the factorisation from 310 to
# 210 GFLOP, the Schur traffic from 9.3 to 7.2 GB and the panel steps of levels >= 6 from 60 to 49.
AXIS_RULE = 2
GAP = 0.75
# Order of the rows inside a front (femo_plan_build_ex2): 1 = the nodes of a separator in the order in which they lie along it, boundary
# lists grouped by owner (nearest ancestor first) in the owner's order -- a child's Schur block then lands in a few long runs of consecutive
# parent rows (scripts/r6_cinv_runs.py); 0 = ascending node id (rounds 1-5).
NODE_ORDER = 1
GAP_NMIN = 128
AXIS_NMIN = 16         # rule 2 measures the separators of pieces of at least this many cells


def analyse(mesh, leaf_size=12, min_depth=0, impl="native", axis_rule=None, gap=None, node_order=None) -> Tree:
    """Bisection tree, node ownership and boundary lists.  ``min_depth`` forces every branch to be
    split at least that deep (the multi-GPU driver needs 2^d subtrees)."""
    axis_rule = AXIS_RULE if axis_rule is None else int(axis_rule)
    gap = GAP if gap is None else float(gap)
    node_order = NODE_ORDER if node_order is None else int(node_order)
    if impl == "native":
        from . import _native
        return _tree_from_native(_native.plan_arrays(mesh, leaf_size, min_depth, axis_rule, gap, node_order))
    if getattr(mesh, "element", "") == "CG2CR1":
        raise NotImplementedError("the numpy twin of the analysis knows the vertex-rotation layouts only; CG2CR1 goes through the native library")
    nel, nP2 = mesh.nel, mesh.nP2
    xc = mesh.nodes[mesh.cells]
    cent = xc.mean(axis=1)
    cext = xc.max(axis=1) - xc.min(axis=1)
    eorder = np.arange(nel)
    lo_l, hi_l, left_l, right_l, parent_l, depth_l = [0], [nel], [-1], [-1], [-1], [0]
    fixed_depth = 0
    while (leaf_size << fixed_depth) < nel:
        fixed_depth += 1
    fixed_depth = max(fixed_depth, min_depth)
    frontier = [0]
    while frontier:
        nxt = []
        for t in frontier:
            lo, hi = lo_l[t], hi_l[t]
            n = hi - lo
            want = depth_l[t] < fixed_depth if gap > 0 else (n > leaf_size or depth_l[t] < min_depth)
            if not want:
                continue
            if n < 2:
                if depth_l[t] < min_depth:
                    raise ValueError("mesh too small for the requested number of partitions")
                continue
            idx = eorder[lo:hi]
            c = cent[idx]
            ext = c.max(axis=0) - c.min(axis=0)
            score = ext
            if axis_rule >= 1:
                mean = np.cumsum(cext[idx], axis=0)[-1] / n              # sequential sums, as the C++ loop forms them
                score = np.where(mean > 0.0, score / np.where(mean > 0.0, mean, 1.0), 0.0)

            def sort_and_cut(ax):
                o = np.argsort(c[:, ax], kind="stable")
                m = n // 2
                if gap > 0 and n >= GAP_NMIN:
                    v = c[o, ax]
                    w = max(1, int(min(0.125, gap / np.sqrt(float(n))) * n))
                    ka, kb = max(1, m - w), min(n - 1, m + w)
                    g = v[ka:kb + 1] - v[ka - 1:kb]
                    cand = np.nonzero(g == g.max())[0] + ka
                    m = int(cand[np.argmin(np.abs(cand - m))])          # nearest the middle; of two equally near ones the lower
                return o, m
            if axis_rule == 2 and n >= AXIS_NMIN:
                # every axis the piece extends in, in the order of rule 1's scores; the smallest separator (in DOFs) wins, the first of equals
                best = None
                for ax in np.argsort(-score, kind="stable"):
                    if not ext[ax] > 0.0:
                        continue
                    o, m0 = sort_and_cut(int(ax))
                    # sep(m) for every cut position of this order: a node is in the separator of m iff the first cell touching it sits
                    # before m and the last one at or after m
                    nodes = mesh.cell_p2[idx[o]].ravel()
                    pos = np.repeat(np.arange(n), mesh.cell_p2.shape[1])
                    u, inv = np.unique(nodes, return_inverse=True)
                    first = np.full(u.size, n); np.minimum.at(first, inv, pos)
                    last = np.full(u.size, -1); np.maximum.at(last, inv, pos)
                    wgt = np.where(u < mesh.nV, 6, 3) * (last > first)
                    diff = np.zeros(n + 2, dtype=np.int64)
                    np.add.at(diff, first + 1, wgt); np.add.at(diff, last + 1, -wgt)
                    sepm = np.cumsum(diff)
                    m = m0
                    if gap > 0 and n >= GAP_NMIN:
                        w = max(1, int(min(0.125, gap / np.sqrt(float(n))) * n))
                        win = np.arange(max(1, n // 2 - w), min(n - 1, n // 2 + w) + 1)
                        cand = win[sepm[win] == sepm[win].min()]
                        m = int(cand[np.argmin(np.abs(cand - m0))])        # smallest separator; of equal ones nearest the largest gap, then the lower
                    sep = int(sepm[m])
                    if best is None or sep < best[0]:
                        best = (sep, o, m)
                if best is None:                      # no axis with an extent (coincident centroids): halve the piece as it stands
                    o, mid = np.arange(n), n // 2
                else:
                    _, o, mid = best
            else:
                o, mid = sort_and_cut(int(np.argmax(score)))
            eorder[lo:hi] = idx[o]
            mid += lo
            for (a, b), store in (((lo, mid), left_l), ((mid, hi), right_l)):
                lo_l.append(a); hi_l.append(b); left_l.append(-1); right_l.append(-1)
                parent_l.append(t); depth_l.append(depth_l[t] + 1)
                store[t] = len(lo_l) - 1
                nxt.append(len(lo_l) - 1)
        frontier = nxt
    T = Tree()
    T.lo, T.hi = np.array(lo_l), np.array(hi_l)
    T.left, T.right, T.parent = np.array(left_l), np.array(right_l), np.array(parent_l)
    T.depth = np.array(depth_l)
    T.ntree = T.lo.size
    T.eorder = eorder
    is_leaf = T.left < 0
    epos = np.empty(nel, dtype=np.int64)
    epos[eorder] = np.arange(nel)
    T.epos = epos
    # owner of every P2 node: the deepest tree node whose element interval holds all its elements
    amin = np.full(nP2, nel, dtype=np.int64)
    amax = np.full(nP2, -1, dtype=np.int64)
    pe = np.repeat(epos, mesh.cell_p2.shape[1])
    np.minimum.at(amin, mesh.cell_p2.ravel(), pe)
    np.maximum.at(amax, mesh.cell_p2.ravel(), pe)
    owner = np.zeros(nP2, dtype=np.int64)
    active = np.ones(nP2, dtype=bool)
    while active.any():
        ids = np.nonzero(active)[0]
        t = owner[ids]
        L, R = T.left[t], T.right[t]
        leaf = L < 0
        mid = np.where(leaf, 0, T.lo[np.where(leaf, 0, R)])
        go_l = ~leaf & (amax[ids] < mid)
        go_r = ~leaf & (amin[ids] >= mid)
        owner[ids[go_l]] = L[go_l]
        owner[ids[go_r]] = R[go_r]
        active[ids[~(go_l | go_r)]] = False
    T.owner = owner
    height = np.zeros(T.ntree, dtype=np.int64)
    order_bu = np.argsort(-T.depth, kind="stable")          # deepest first
    for t in order_bu:
        if not is_leaf[t]:
            height[t] = 1 + max(height[T.left[t]], height[T.right[t]])
    T.height = height
    ord_owner = np.argsort(owner, kind="stable")
    cnt = np.bincount(owner, minlength=T.ntree)
    st = np.concatenate([[0], np.cumsum(cnt)])
    T.piv_nodes = [ord_owner[st[t]:st[t + 1]] for t in range(T.ntree)]
    bnd = [None] * T.ntree
    for t in order_bu:
        if is_leaf[t]:
            touched = np.unique(mesh.cell_p2[eorder[T.lo[t]:T.hi[t]]])
            bnd[t] = touched[owner[touched] != t]
        else:
            u = np.union1d(bnd[T.left[t]], bnd[T.right[t]])
            bnd[t] = u[owner[u] != t]
    if node_order == 1:
        # the nodes of a separator in the order in which they lie along it (coordinate, along the axis of the separator's largest extent,
        # of the mean centroid of the touching cells; ties by node id), boundary lists grouped by owner -- nearest ancestor first -- in
        # the owner's order (csrc/symbolic.cpp, node_order 1)
        nc = np.zeros((nP2, 3))
        np.add.at(nc, mesh.cell_p2.ravel(), np.repeat(cent, mesh.cell_p2.shape[1], axis=0))      # sequential, (cell, local node) order
        nc /= np.maximum(np.bincount(mesh.cell_p2.ravel(), minlength=nP2), 1)[:, None]
        rank = np.zeros(nP2, dtype=np.int64)
        for t in range(T.ntree):
            p = T.piv_nodes[t]
            if p.size > 1:
                ax = int(np.argmax(nc[p].max(axis=0) - nc[p].min(axis=0)))
                p = p[np.argsort(nc[p, ax], kind="stable")]
                T.piv_nodes[t] = p
            rank[p] = np.arange(p.size)
        for t in range(T.ntree):
            b = bnd[t]
            bnd[t] = b[np.lexsort((rank[b], -T.depth[owner[b]]))]
    T.bnd_nodes = bnd
    return T


def _assemble_plan(dof_lists, npiv, parent, left, right, level_of, elem_front, elem_dofs):
    """FrontalPlan from per-front DOF lists (pivots first) in one consistent DOF numbering."""
    ntree = len(dof_lists)
    plan = FrontalPlan()
    plan.ntree = ntree
    plan.npiv = np.asarray(npiv, dtype=np.int32)
    plan.nf = np.array([len(d) for d in dof_lists], dtype=np.int32)
    plan.parent = np.asarray(parent, dtype=np.int32)
    plan.left, plan.right = np.asarray(left, dtype=np.int32), np.asarray(right, dtype=np.int32)
    plan.nleaves = int(np.sum(plan.left < 0))
    plan.dof_off = np.concatenate([[0], np.cumsum(plan.nf.astype(np.int64))])
    plan.front_dofs = (np.concatenate(dof_lists) if ntree else np.zeros(0)).astype(np.int32)
    plan.front_off = np.concatenate([[0], np.cumsum(plan.nf.astype(np.int64) ** 2)])
    sorters = {}

    def positions(t, dofs):
        if t not in sorters:
            fd = np.asarray(dof_lists[t])
            o = np.argsort(fd, kind="stable")
            sorters[t] = (fd[o], o)
        fs, o = sorters[t]
        k = np.searchsorted(fs, dofs)
        if np.any(k >= fs.size) or np.any(fs[np.minimum(k, fs.size - 1)] != dofs):
            raise AssertionError("a boundary DOF is missing from the parent front")
        return o[k]

    up_map = np.full(plan.front_dofs.size, -1, dtype=np.int32)
    for t in range(ntree):
        p = plan.parent[t]
        if p >= 0 and plan.nf[t] > plan.npiv[t]:
            up_map[plan.dof_off[t] + plan.npiv[t]: plan.dof_off[t + 1]] = positions(p, np.asarray(dof_lists[t])[plan.npiv[t]:])
    plan.up_map = up_map
    plan.elem_front = np.asarray(elem_front, dtype=np.int32)
    elem_map = np.empty(elem_dofs.shape, dtype=np.int32)
    order = np.argsort(plan.elem_front, kind="stable")
    bounds = np.searchsorted(plan.elem_front[order], np.arange(ntree + 1))
    for t in range(ntree):
        es = order[bounds[t]:bounds[t + 1]]
        if es.size:
            elem_map[es] = positions(t, elem_dofs[es].ravel()).reshape(es.size, -1)
    plan.elem_map = elem_map
    level_of = np.asarray(level_of)
    plan.nlevels = int(level_of.max()) + 1
    plan.height = level_of.astype(np.int32)
    lev_nodes = [np.nonzero(level_of == h)[0] for h in range(plan.nlevels)]
    plan.level_nodes = [ln[np.argsort(-plan.nf[ln], kind="stable")].astype(np.int32) for ln in lev_nodes]
    return plan


def _plan_from_native(A) -> FrontalPlan:
    plan = FrontalPlan()
    plan.ntree = int(A["nf"].size)
    for k in ("npiv", "nf", "parent", "left", "right", "front_dofs", "up_map", "elem_front", "elem_map", "height", "eorder"):
        setattr(plan, k, A[k])
    plan.nleaves = int(np.sum(plan.left < 0))
    plan.dof_off = A["dof_off"]
    plan.front_off = np.concatenate([[0], np.cumsum(plan.nf.astype(np.int64) ** 2)])
    lo = A["level_off"]
    plan.nlevels = int(lo.size - 1)
    plan.level_nodes = [A["level_nodes"][lo[h]:lo[h + 1]] for h in range(plan.nlevels)]
    return plan


def build_plan(mesh, leaf_size=12, impl="native", axis_rule=None, gap=None, node_order=None) -> FrontalPlan:
    """Single-GPU plan: every tree node is a front, levels by height."""
    axis_rule = AXIS_RULE if axis_rule is None else int(axis_rule)
    gap = GAP if gap is None else float(gap)
    node_order = NODE_ORDER if node_order is None else int(node_order)
    if impl == "native":
        from . import _native
        return _plan_from_native(_native.plan_arrays(mesh, leaf_size, 0, axis_rule, gap, node_order))
    T = analyse(mesh, leaf_size, impl="python", axis_rule=axis_rule, gap=gap, node_order=node_order)
    nV, ndof_u = mesh.nV, mesh.ndof_u
    # DOF lists of all fronts from ONE expansion of the concatenated node lists (pivots first, then the boundary)
    seq = [a for t in range(T.ntree) for a in (T.piv_nodes[t], T.bnd_nodes[t])]
    lens = np.array([len(a) for a in seq], dtype=np.int64)
    nodes_all = np.concatenate(seq).astype(np.int64) if seq else np.zeros(0, np.int64)
    csum = np.concatenate([[0], np.cumsum(np.where(nodes_all < nV, 6, 3))])
    ends = np.cumsum(lens)
    ndofs_seg = csum[ends] - csum[ends - lens]
    dofs_all = _node_dofs(nodes_all, nV, ndof_u)
    seg_off = np.concatenate([[0], np.cumsum(ndofs_seg)])
    dof_lists = [dofs_all[seg_off[2 * t]:seg_off[2 * t + 2]] for t in range(T.ntree)]
    npiv = ndofs_seg[0::2]
    leaf_of_pos = np.zeros(mesh.nel, dtype=np.int64)
    for t in np.nonzero(T.left < 0)[0]:
        leaf_of_pos[T.lo[t]:T.hi[t]] = t
    plan = _assemble_plan(dof_lists, npiv, T.parent, T.left, T.right, T.height, leaf_of_pos[T.epos], mesh.cell_dofs())
    plan.eorder = T.eorder.astype(np.int32)
    return plan


def rank_plan(mesh, T: Tree, rank, nranks):
    return _rank_plan(mesh, T, rank, nranks)


def _rank_plan(mesh, T: Tree, rank, nranks):
    """Cut the plan of one rank out of the global tree ``T`` (``analyse(mesh, leaf, min_depth=log2 nranks)``).

    Returns ``(sub, plan, info)``: the rank's sub-mesh (its elements, own numbering), the plan in the
    sub-mesh's DOF numbering extended by ghost entries for the replicated separator DOFs the rank's
    elements do not touch, and a dict with
      ``cells``       global ids of the rank's elements (sub-mesh cell order)
      ``vertices``    global vertex ids of the sub-mesh vertices
      ``nghost``      number of ghost entries
      ``top_local``   local vector indices of all replicated (top) DOFs in a canonical global order
      ``n_local_levels``  levels [0, n) are the rank's own subtree, [n, nlevels) the replicated top
      ``root_front``  local front id of the rank's subtree root
      ``stub_fronts`` local front ids standing for the subtree roots of ranks 0..nranks-1 (own entry = root_front)
      ``schur_sizes`` boundary sizes of all subtree roots (for the padded all-gather)
      ``l2g_dof``     global DOF of every local vector entry
    """
    d = int(np.log2(nranks))
    if 2 ** d != nranks:
        raise ValueError("number of ranks must be a power of two")
    roots = np.nonzero(T.depth == d)[0]
    roots = roots[np.argsort(T.lo[roots], kind="stable")]
    if roots.size != nranks or np.any(T.depth[T.left < 0] < d):
        raise ValueError("tree is not deep enough for this many ranks")
    root = int(roots[rank])
    nVg, ndof_ug = mesh.nV, mesh.ndof_u
    # CG2CR1 (linear_shell_model.py:68-73): the rotation lives on the edge midpoints, P2 nodes nV .. nV + nE - 1
    cr = getattr(mesh, "element", "") == "CG2CR1"
    rot_g = (mesh.nV, mesh.nE) if cr else None
    # ---- sub-mesh
    cells_g = np.sort(T.eorder[T.lo[root]:T.hi[root]])
    verts_g = np.unique(mesh.cells[cells_g])
    g2l_v = -np.ones(mesh.nn, dtype=np.int64)
    g2l_v[verts_g] = np.arange(verts_g.size)
    sub = ShellMesh(mesh.nodes[verts_g], g2l_v[mesh.cells[cells_g]], mesh.element)
    l2g_p2 = np.empty(sub.nP2, dtype=np.int64)
    l2g_p2[sub.cell_p2.ravel()] = mesh.cell_p2[cells_g].ravel()
    l2g_dof = np.empty(sub.ndof, dtype=np.int64)
    for c in range(3):
        l2g_dof[3 * np.arange(sub.nP2) + c] = 3 * l2g_p2 + c
        if cr:
            l2g_dof[sub.ndof_u + 3 * np.arange(sub.nE) + c] = ndof_ug + 3 * (l2g_p2[sub.nV:sub.nV + sub.nE] - nVg) + c
        else:
            l2g_dof[sub.ndof_u + 3 * np.arange(sub.nV) + c] = ndof_ug + 3 * verts_g + c
    # ---- which tree nodes become fronts here
    top = np.nonzero(T.depth < d)[0]
    in_sub = np.zeros(T.ntree, dtype=bool)
    stack = [root]
    while stack:
        t = stack.pop()
        in_sub[t] = True
        if T.left[t] >= 0:
            stack += [T.left[t], T.right[t]]
    local = np.nonzero(in_sub)[0]
    stubs = np.array([r for r in roots if r != root], dtype=np.int64)
    order = np.concatenate([local, stubs, top]).astype(np.int64)
    new_id = -np.ones(T.ntree, dtype=np.int64)
    new_id[order] = np.arange(order.size)
    # ---- replicated DOFs and ghosts
    top_dofs_g = np.sort(np.concatenate([_node_dofs(T.piv_nodes[t], nVg, ndof_ug, rot_g) for t in top])) if top.size else np.zeros(0, np.int64)
    srt = np.argsort(l2g_dof, kind="stable")
    lg_sorted = l2g_dof[srt]
    k = np.searchsorted(lg_sorted, top_dofs_g)
    k_c = np.minimum(k, max(lg_sorted.size - 1, 0))
    present = (k < lg_sorted.size) & (lg_sorted[k_c] == top_dofs_g)
    top_local = np.empty(top_dofs_g.size, dtype=np.int64)
    top_local[present] = srt[k_c[present]]
    nghost = int((~present).sum())
    top_local[~present] = sub.ndof + np.arange(nghost)
    l2g_full = np.concatenate([l2g_dof, top_dofs_g[~present]])
    srt_f = np.argsort(l2g_full, kind="stable")
    lgf_sorted = l2g_full[srt_f]

    def g2l(dofs_g):
        kk = np.searchsorted(lgf_sorted, dofs_g)
        assert np.all(lgf_sorted[kk] == dofs_g), "a front DOF is neither local nor replicated"
        return srt_f[kk]

    # ---- fronts
    dof_lists, npiv, level_of = [], [], []
    h_root = int(T.height[root])
    for t in order:
        pd = _node_dofs(T.piv_nodes[t], nVg, ndof_ug, rot_g)
        bd = _node_dofs(T.bnd_nodes[t], nVg, ndof_ug, rot_g)
        if in_sub[t]:
            dof_lists.append(g2l(np.concatenate([pd, bd]))); npiv.append(pd.size)
            level_of.append(int(T.height[t]))
        elif T.depth[t] == d:                        # another rank's subtree root: Schur complement only
            dof_lists.append(g2l(bd)); npiv.append(0)
            level_of.append(0)
        else:                                        # replicated top of the tree
            dof_lists.append(g2l(np.concatenate([pd, bd]))); npiv.append(pd.size)
            level_of.append(h_root + (d - int(T.depth[t])))
    parent = [new_id[T.parent[t]] if T.parent[t] >= 0 else -1 for t in order]
    left = [new_id[T.left[t]] if (T.left[t] >= 0 and (in_sub[t] or T.depth[t] < d)) else -1 for t in order]
    right = [new_id[T.right[t]] if (T.right[t] >= 0 and (in_sub[t] or T.depth[t] < d)) else -1 for t in order]
    # ---- elements (sub-mesh cell order) -> leaf fronts
    leaf_of_pos = np.zeros(mesh.nel, dtype=np.int64)
    for t in local[T.left[local] < 0]:
        leaf_of_pos[T.lo[t]:T.hi[t]] = t
    elem_front = new_id[leaf_of_pos[T.epos[cells_g]]]
    plan = _assemble_plan(dof_lists, npiv, parent, left, right, level_of, elem_front, sub.cell_dofs())
    info = dict(cells=cells_g, vertices=verts_g, nghost=nghost, top_local=top_local.astype(np.int64),
                n_local_levels=h_root + 1, root_front=int(new_id[root]),
                stub_fronts=[int(new_id[r]) for r in roots],
                schur_sizes=[int(_node_dofs(T.bnd_nodes[r], nVg, ndof_ug, rot_g).size) for r in roots],
                l2g_dof=l2g_full, l2g_p2=l2g_p2, n_top=int(top_dofs_g.size))
    return sub, plan, info
