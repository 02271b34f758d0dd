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
"""
from __future__ import annotations

import numpy as np

__all__ = ["FrontalPlan", "build_plan"]


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


def _node_dofs(nodes, nV, ndof_u):
    """DOF ids of P2 nodes (vertices carry u and theta: 6 DOFs, the others 3), concatenated in order."""
    nodes = np.asarray(nodes, dtype=np.int64)
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
    return out, off


def build_plan(mesh, leaf_size=16) -> FrontalPlan:
    nel, nV, nP2, ndof_u = mesh.nel, mesh.nV, mesh.nP2, mesh.ndof_u
    cent = mesh.nodes[mesh.cells].mean(axis=1)
    # ---------------------------------------------------------------- 1. bisection tree over elements
    eorder = np.arange(nel)
    lo_l, hi_l, left_l, right_l, parent_l, depth_l = [0], [nel], [-1], [-1], [-1], [0]
    stack = [0]
    while stack:
        t = stack.pop()
        lo, hi = lo_l[t], hi_l[t]
        if hi - lo <= leaf_size:
            continue
        idx = eorder[lo:hi]
        c = cent[idx]
        ax = int(np.argmax(c.max(axis=0) - c.min(axis=0)))
        eorder[lo:hi] = idx[np.argsort(c[:, ax], kind="stable")]
        mid = lo + (hi - lo) // 2
        for (a, b), store in (((lo, mid), left_l), ((mid, hi), right_l)):
            lo_l.append(a); hi_l.append(b); left_l.append(-1); right_l.append(-1)
            parent_l.append(t); depth_l.append(depth_l[t] + 1)
            store[t] = len(lo_l) - 1
            stack.append(len(lo_l) - 1)
    t_lo, t_hi = np.array(lo_l), np.array(hi_l)
    t_left, t_right, t_parent = np.array(left_l), np.array(right_l), np.array(parent_l)
    ntree = t_lo.size
    is_leaf = t_left < 0
    epos = np.empty(nel, dtype=np.int64)
    epos[eorder] = np.arange(nel)
    # ---------------------------------------------------------------- 2. owner of every P2 node
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
        L, R = t_left[t], t_right[t]
        leaf = L < 0
        mid = np.where(leaf, 0, t_lo[np.where(leaf, 0, R)])
        go_l = ~leaf & (amax[ids] < mid)
        go_r = ~leaf & (amin[ids] >= mid)
        owner[ids[go_l]] = L[go_l]
        owner[ids[go_r]] = R[go_r]
        active[ids[~(go_l | go_r)]] = False
    # ---------------------------------------------------------------- 3. heights / levels (children before parents)
    height = np.zeros(ntree, dtype=np.int64)
    order_bu = np.argsort(-np.array(depth_l), kind="stable")          # deepest first
    for t in order_bu:
        if not is_leaf[t]:
            height[t] = 1 + max(height[t_left[t]], height[t_right[t]])
    nlevels = int(height.max()) + 1
    # ---------------------------------------------------------------- 4. pivot / boundary node lists
    piv_nodes = [None] * ntree
    ord_owner = np.argsort(owner, kind="stable")
    cnt = np.bincount(owner, minlength=ntree)
    st = np.concatenate([[0], np.cumsum(cnt)])
    for t in range(ntree):
        piv_nodes[t] = ord_owner[st[t]:st[t + 1]]
    bnd_nodes = [None] * ntree
    for t in order_bu:
        if is_leaf[t]:
            touched = np.unique(mesh.cell_p2[eorder[t_lo[t]:t_hi[t]]])
            bnd_nodes[t] = touched[owner[touched] != t]
        else:
            u = np.union1d(bnd_nodes[t_left[t]], bnd_nodes[t_right[t]])
            bnd_nodes[t] = u[owner[u] != t]
    # ---------------------------------------------------------------- 5. fronts in DOF space
    plan = FrontalPlan()
    plan.ntree, plan.nlevels, plan.nleaves = ntree, nlevels, int(is_leaf.sum())
    plan.eorder = eorder.astype(np.int32)
    plan.parent = t_parent.astype(np.int32)
    plan.left, plan.right = t_left.astype(np.int32), t_right.astype(np.int32)
    plan.height = height.astype(np.int32)
    npiv = np.zeros(ntree, dtype=np.int32)
    nf = np.zeros(ntree, dtype=np.int32)
    dof_lists = [None] * ntree
    pos_in_front = [None] * ntree          # dict-free lookup: sorted dofs + positions
    for t in range(ntree):
        pd, _ = _node_dofs(piv_nodes[t], nV, ndof_u)
        bd, _ = _node_dofs(bnd_nodes[t], nV, ndof_u)
        npiv[t] = pd.size
        nf[t] = pd.size + bd.size
        dof_lists[t] = np.concatenate([pd, bd])
    plan.npiv, plan.nf = npiv, nf
    plan.dof_off = np.concatenate([[0], np.cumsum(nf.astype(np.int64))])
    plan.front_dofs = np.concatenate(dof_lists).astype(np.int32)
    plan.front_off = np.concatenate([[0], np.cumsum(nf.astype(np.int64) ** 2)])

    def positions(t, dofs):
        """positions of global DOFs inside front t (every DOF must be present)."""
        fd = dof_lists[t]
        o = np.argsort(fd, kind="stable")
        k = np.searchsorted(fd[o], dofs)
        assert np.all(fd[o][k] == dofs)
        return o[k]

    # child boundary -> parent front positions, stored at dof_off[child] + npiv[child] ...
    up_map = np.full(plan.front_dofs.size, -1, dtype=np.int32)
    for t in range(ntree):
        p = t_parent[t]
        if p >= 0 and nf[t] > npiv[t]:
            up_map[plan.dof_off[t] + npiv[t]: plan.dof_off[t + 1]] = positions(p, dof_lists[t][npiv[t]:])
    plan.up_map = up_map
    # element -> leaf front positions
    leaf_of_pos = np.zeros(nel, dtype=np.int64)
    for t in np.nonzero(is_leaf)[0]:
        leaf_of_pos[t_lo[t]:t_hi[t]] = t
    elem_front = leaf_of_pos[epos].astype(np.int32)
    cd = mesh.cell_dofs()
    elem_map = np.empty(cd.shape, dtype=np.int32)
    for t in np.nonzero(is_leaf)[0]:
        es = eorder[t_lo[t]:t_hi[t]]
        elem_map[es] = positions(t, cd[es].ravel()).reshape(es.size, -1)
    plan.elem_front, plan.elem_map = elem_front, elem_map
    # level lists (by height), sorted by front size so that a level's launches are load balanced
    lev_nodes = [np.nonzero(height == h)[0] for h in range(nlevels)]
    plan.level_nodes = [ln[np.argsort(-nf[ln], kind="stable")].astype(np.int32) for ln in lev_nodes]
    plan.owner_of_dof = None
    return plan
