"""Shell surface mesh container and DOF numbering for the CG2 x CG1 RM shell space.

The reference hands a ``dolfinx.mesh.Mesh`` (tdim 2, gdim 3) to ``RMShellModel``
(reference femo_alpha/rm_shell/rm_shell_model.py:31-81) and lets dolfinx build the
mixed space ``[Lagrange-2]^3 x [Lagrange-1]^3`` (reference
femo_alpha/rm_shell/linear_shell_fenicsx/linear_shell_model.py:60-65).  This module is
the build's substitute for both: it takes the same raw data
``reconstructFEAMesh(filename, nodes, connectivity)`` accepts (reference
femo_alpha/fea/utils_dolfinx.py:652-668), i.e. ``nodes (nn,3)`` and a
counter-clockwise ``connectivity (nel,4)`` (quads) or ``(nel,3)`` (triangles), and
derives the P2 node set, the state-vector layout and the facet sets the penalty /
strong Dirichlet conditions need.

State-vector layout (solver-internal, opaque to callers exactly as in the reference,
SURVEY.md section 8a row N)::

    w = [ u(P2 node 0) xyz, u(P2 node 1) xyz, ...,  theta(vertex 0) xyz, ... ]       (CG2CR1: theta(edge 0) xyz, ...)
    P2 node ids:  vertices 0..nV-1 | edge midpoints nV..nV+nE-1 | cell centres (quads)

so ``ndof = 3*(nV+nE+nC) + 3*nV`` on quads (``3*(nV+nE) + 3*nV`` on triangles).
Caller node order is kept: no hidden renumbering, hence the reference's
``input_global_indices`` / ``original_cell_index`` permutations
(rm_shell_model.py:398-438) are identities here.
"""
from __future__ import annotations

import numpy as np

__all__ = ["ShellMesh", "plate_mesh", "wing_skin_mesh", "unstructured_skin_mesh", "unstructured_quad_skin_mesh", "quads_to_triangles"]


class ShellMesh:
    """Linear-geometry surface mesh of quads (4 CCW vertices) or triangles in R^3."""

    def __init__(self, nodes, cells, element="CG2CG1"):
        # element: the mixed space of the state (ShellElement.setUpFunctionSpace, linear_shell_model.py:47-86).  'CG2CG1' -- the
        # one RMShellPDE selects (rm_shell_pde.py:27): displacement on the P2 nodes, rotation on the vertices; 'CG1CG1'
        # (:74-79): both on the vertices -- the "P2 node" set then IS the vertex set and every table of the displacement is the
        # bilinear / linear one; 'CG2CR1' (:68-73, triangles only as in the reference): displacement on the P2 nodes, rotation on the
        # EDGE MIDPOINTS with the Crouzeix-Raviart functions -- the rotation nodes are the P2 nodes nV .. nV + nE - 1.
        if element not in ("CG2CG1", "CG1CG1", "CG2CR1"):
            raise ValueError("Invalid element type.")
        self.element = element
        nodes = np.ascontiguousarray(np.asarray(nodes, dtype=np.float64))
        cells = np.ascontiguousarray(np.asarray(cells, dtype=np.int32))
        if nodes.ndim != 2 or nodes.shape[1] not in (2, 3):
            raise ValueError("nodes must be (nn,2) or (nn,3)")
        if nodes.shape[1] == 2:
            nodes = np.hstack([nodes, np.zeros((nodes.shape[0], 1))])
        if cells.ndim != 2 or cells.shape[1] not in (3, 4):
            raise ValueError("Invalid cell shape--should be either triangular or quadrilateral")
        if cells.size and (cells.min() < 0 or cells.max() >= nodes.shape[0]):
            raise ValueError("connectivity refers to a node that does not exist")
        if element == "CG2CR1" and cells.shape[1] != 3:
            raise ValueError("Invalid element type.")          # CR1 is a simplex element
        self.nodes = nodes
        self.cells = cells
        self.nn = nodes.shape[0]
        self.nel = cells.shape[0]
        self.nvc = cells.shape[1]            # vertices per cell (4 quad, 3 triangle)
        self.is_quad = self.nvc == 4
        self._build_edges()
        self._build_p2()

    # ------------------------------------------------------------------ topology
    def _build_edges(self):
        nvc = self.nvc
        a = self.cells
        b = np.roll(self.cells, -1, axis=1)          # edge k: vertex k -> vertex (k+1)%nvc
        lo = np.minimum(a, b).astype(np.int64)
        hi = np.maximum(a, b).astype(np.int64)
        key = (lo * self.nn + hi).ravel()
        ukey, inv = np.unique(key, return_inverse=True)
        self.nE = ukey.size
        self.edges = np.stack([ukey // self.nn, ukey % self.nn], axis=1).astype(np.int32)
        self.cell_edges = inv.reshape(self.nel, nvc).astype(np.int32)
        # edge -> incident (cell, local edge) pairs.  Surface meshes of built-up structures branch: a rib or spar
        # meets the skin along edges shared by three or more cells -- the incidence is kept as CSR lists;
        # edge_cells / edge_local hold its first two entries (all of it on a manifold mesh).
        order = np.argsort(inv, kind="stable")
        counts = np.bincount(inv, minlength=self.nE)
        start = np.concatenate([[0], np.cumsum(counts)[:-1]])
        flat_cell = (order // nvc).astype(np.int32)
        flat_loc = (order % nvc).astype(np.int32)
        self.edge_count = counts.astype(np.int32)
        self.edge_inc_off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        self.edge_inc_cell, self.edge_inc_local = flat_cell, flat_loc
        self.edge_cells = -np.ones((self.nE, 2), dtype=np.int32)
        self.edge_local = -np.ones((self.nE, 2), dtype=np.int32)
        self.edge_cells[:, 0] = flat_cell[start]
        self.edge_local[:, 0] = flat_loc[start]
        two = counts >= 2
        self.edge_cells[two, 1] = flat_cell[start[two] + 1]
        self.edge_local[two, 1] = flat_loc[start[two] + 1]
        self.boundary_edges = np.nonzero(counts == 1)[0].astype(np.int32)
        self.is_manifold = bool(counts.max(initial=0) <= 2)

    def facet_incidence(self, edges):
        """(cell, local edge) pairs of all cells incident to the given edges, (n, 2) int32."""
        edges = np.asarray(edges, dtype=np.int64)
        if edges.size == 0:
            return np.zeros((0, 2), dtype=np.int32)
        cnt = self.edge_count[edges]
        base = np.repeat(self.edge_inc_off[edges], cnt)
        within = np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt)
        idx = base + within
        return np.stack([self.edge_inc_cell[idx], self.edge_inc_local[idx]], axis=1).astype(np.int32)

    def _build_p2(self):
        nV, nE, nC = self.nn, self.nE, (self.nel if self.is_quad else 0)
        if self.element == "CG1CG1":
            nE_p2, nC = 0, 0
        else:
            nE_p2 = nE
        self.nV, self.nC = nV, nC
        self.nP2 = nV + nE_p2 + nC
        cols = [self.cells]
        if self.element != "CG1CG1":
            cols.append(nV + self.cell_edges)
        if self.is_quad and self.element != "CG1CG1":
            cols.append((nV + nE + np.arange(self.nel, dtype=np.int32))[:, None])
        # (nel, 9) quads: 4 vertices, 4 edge midpoints, centre; (nel, 6) triangles
        self.cell_p2 = np.ascontiguousarray(np.hstack(cols).astype(np.int32))
        self.npc = self.cell_p2.shape[1]
        self.ndof_u = 3 * self.nP2
        # rotation nodes: the vertices, or (CG2CR1) the edge midpoints -- node k of that set is edge k = P2 node nV + k
        self.nR = nE if self.element == "CG2CR1" else nV
        self.cell_rot = self.cell_edges if self.element == "CG2CR1" else self.cells          # (nel, nvc) rotation node of every local slot
        self.ndof_t = 3 * self.nR
        self.ndof = self.ndof_u + self.ndof_t
        self.ldof = 3 * self.npc + 3 * self.nvc      # 39 (quad) / 27 (triangle)

    @property
    def p2_coords(self):
        """Coordinates of the P2 nodes under the (bi)linear geometry map."""
        x = self.nodes
        if self.element == "CG1CG1":
            return x
        parts = [x, 0.5 * (x[self.edges[:, 0]] + x[self.edges[:, 1]])]
        if self.is_quad:
            parts.append(x[self.cells].mean(axis=1))
        return np.vstack(parts)

    def cell_dofs(self):
        """(nel, ldof) global DOF numbers, element-local order [u_a xyz ..., theta_b xyz ...]."""
        npc, nvc = self.cell_p2.shape[1], self.cells.shape[1]
        out = np.empty((self.nel, 3 * npc + 3 * nvc), dtype=np.int32)
        p3, v3 = 3 * self.cell_p2.astype(np.int32), np.int32(self.ndof_u) + 3 * self.cell_rot.astype(np.int32)
        for c in range(3):
            out[:, c:3 * npc:3] = p3 + np.int32(c)
            out[:, 3 * npc + c::3] = v3 + np.int32(c)
        return out

    def p2_integrals(self, nquad=4):
        """(nel, npc): int N2_a dS over every cell, N2 the P2 basis of the mid-surface displacement in ``cell_p2`` order --
        the consistent nodal loads of a unit pressure (nquad x nquad Gauss on quads, the rule of the device load vector; on
        triangles the vertex functions integrate to zero and every edge function to a third of the area)."""
        X = self.nodes[self.cells]
        if not self.is_quad:
            area = 0.5 * np.linalg.norm(np.cross(X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]), axis=1)
            return np.concatenate([np.zeros((self.nel, 3)), np.repeat(area[:, None] / 3.0, 3, axis=1)], axis=1)
        g, w = np.polynomial.legendre.leggauss(nquad)
        lag2 = lambda t: np.stack([0.5 * t * (t - 1.0), 1.0 - t * t, 0.5 * t * (t + 1.0)], axis=1)       # nodes -1, 0, 1
        q2 = [(0, 0), (2, 0), (2, 2), (0, 2), (1, 0), (2, 1), (1, 2), (0, 1), (1, 1)]                      # cell_p2 order
        xi, eta = np.repeat(g, nquad), np.tile(g, nquad)
        wq = np.repeat(w, nquad) * np.tile(w, nquad)
        a, b = lag2(xi), lag2(eta)
        N2 = np.stack([a[:, i] * b[:, j] for i, j in q2], axis=1)                                         # (points, 9 nodes)
        sx, sy = np.array([-1, 1, 1, -1.0]), np.array([-1, -1, 1, 1.0])
        dN = np.stack([0.25 * sx[None] * (1 + sy[None] * eta[:, None]), 0.25 * sy[None] * (1 + sx[None] * xi[:, None])], axis=-1)
        J = np.einsum("ebi,qbk->eqik", X, dN)
        det = np.linalg.norm(np.cross(J[..., 0], J[..., 1]), axis=-1)
        return np.einsum("q,eq,qa->ea", wq, det, N2)

    def cell_diameters(self):
        """UFL ``CellDiameter``: largest distance between two vertices of the cell
        (used at linear_shell_model.py:285,325,339)."""
        x = self.nodes[self.cells]
        d = np.zeros(self.nel)
        for i in range(self.nvc):
            for j in range(i + 1, self.nvc):
                d = np.maximum(d, np.linalg.norm(x[:, i] - x[:, j], axis=1))
        return d

    def recommended_leaf_size(self):
        """Cells per leaf of the nested dissection (ShellContext.enable_frontal / use_direct_solver, DistributedShell): 12 quadrilaterals;
        24 triangles -- two triangles cover one quadrilateral's area and DOFs, and leaves of 12 triangles add a tree level of small fronts
        (1 M-DOF triangle skin: forward 20.8 ms with 12, 20.3 with 24, scripts/r4_tri.py)."""
        return 12 if self.is_quad else 24

    def recommended_nquad(self, nodal_nu_varies=False):
        """The rule that reproduces the reference's integration of the static forms: Gauss points per direction on quadrilaterals, the
        DEGREE of the symmetric rule on triangles (femo_create's ``nquad``).  The reference leaves
        the degree to UFL (plain ``dx``, linear_shell_model.py:88-103), whose estimate is 43-53 on quadrilaterals
        (scripts/ufl_degree_estimate.py): (nearly) exact integration.  On affine cells (parallelograms) the integrand is a
        polynomial of degree <= 7 per direction and 4 points are exact; on any other quadrilateral the frame, the
        derivative map and the differentiated normal are rational in the reference coordinates and the answer converges in n:
          * mildly non-affine cells (BASELINE config 3: a jittered grid, the Jacobian of the bilinear map varies by a factor 1.4
            across the typical cell): 4 points are 7.5e-8 away from the limit in d compliance / d thickness, 5 points 1e-9
            (tests/test_gpu_fullsize.py::test_quadrature_rule_sensitivity_at_config3) -> 5;
          * strongly non-affine cells (an unstructured quadrilateral mesh whose typical cell is a kite: factor 3; workload
            uquad1m): 5 points are 2.1e-8 away in the gradient (9e-10 displacement, 1.1e-9 compliance), 6 points 3.6e-10
            (profiles/r5_quadrature_uquad1m.txt: exact discrete solutions at n = 5, 6, 7) -> 6.
        The measure is the MEDIAN over the cells of max / min of the Jacobian at the four corners (wing1m 1.43, uquad1m 3.00; the
        worst cells of the jittered grid reach 4.9, but what moves the solution is the typical cell); 6 from 2.0 on.
        Triangles are affine, the surface gradient, the frame and (uhat being piecewise linear) F, J are constant on a cell, so with
        nodal thickness and E the integrand is a polynomial of degree <= 6 (drilling: E h^3 omega^2) and the 12-point rule of degree 6 is
        exact -- the same number as the degree-9 rule UFL's estimate selects (scripts/ufl_degree_estimate.py), up to rounding.  Only a
        NODAL Poisson ratio that varies over a cell makes the integrand rational (E / (1 - nu^2), E / (2 (1 + nu))): then the reference's
        answer is that of its degree-9 rule, 19 points (``nodal_nu_varies``; ShellContext switches by itself when such a field arrives).
        The p-norm stress measure is degree 4 on either cell type whatever this returns (rm_shell_model.py:200-205)."""
        if not self.is_quad:
            return 9 if nodal_nu_varies else 6
        x = self.nodes[self.cells]
        defect = np.linalg.norm(x[:, 0] - x[:, 1] + x[:, 2] - x[:, 3], axis=1)       # zero for a parallelogram
        if np.all(defect <= 1e-10 * self.cell_diameters()):
            return 4
        e = [x[:, 1] - x[:, 0], x[:, 2] - x[:, 1], x[:, 2] - x[:, 3], x[:, 3] - x[:, 0]]     # edges 01, 12, 32, 03
        area = lambda u, v: np.linalg.norm(np.cross(u, v), axis=1)
        J = np.stack([area(e[0], e[3]), area(e[0], e[1]), area(e[2], e[1]), area(e[2], e[3])], axis=1)
        ratio = J.max(axis=1) / np.maximum(J.min(axis=1), 1e-300)
        return 6 if np.median(ratio) >= 2.0 else 5

    # ------------------------------------------------------------------ Dirichlet sets
    @staticmethod
    def _eval_marker(func, pts):
        """Call a dolfinx-style marker ``func(x)`` with ``x`` of shape (3, npts)."""
        out = np.asarray(func(np.ascontiguousarray(pts.T)))
        if out.shape != (pts.shape[0],):
            out = np.broadcast_to(out, (pts.shape[0],))
        return out.astype(bool)

    def locate_facets(self, func, boundary_only):
        """Facets (edges) all of whose vertices satisfy ``func`` -- the semantics of
        ``dolfinx.mesh.locate_entities_boundary`` / ``locate_entities`` used by
        ``createCustomMeasure`` (reference femo_alpha/fea/utils_dolfinx.py:555-565)."""
        mark = self._eval_marker(func, self.nodes)
        ok = mark[self.edges[:, 0]] & mark[self.edges[:, 1]]
        interior = self.edge_cells[:, 1] >= 0
        ok &= ~interior if boundary_only else np.ones_like(ok)
        return np.nonzero(ok)[0].astype(np.int32)

    def penalty_facets(self, func):
        """(cell, local_edge) pairs the penalty residual integrates over:
        tagged exterior facets once (``ds(100)``) and tagged interior facets from both
        sides (``dS(100)``, '+' and '-' restrictions) -- reference
        linear_shell_model.py:323-333 with measures from rm_shell_model.py:88-95."""
        ext = self.locate_facets(func, boundary_only=True)
        allf = self.locate_facets(func, boundary_only=False)
        inte = allf[self.edge_cells[allf, 1] >= 0]
        if not self.is_manifold:
            # a branching edge is penalised from every cell that meets it (dolfinx's dS would pick two of them; the
            # penalty only pins the DOFs of the edge, so the solution does not depend on the choice)
            return np.vstack([self.facet_incidence(ext), self.facet_incidence(inte)]).astype(np.int32)
        cells = np.concatenate([self.edge_cells[ext, 0], self.edge_cells[inte, 0], self.edge_cells[inte, 1]])
        locs = np.concatenate([self.edge_local[ext, 0], self.edge_local[inte, 0], self.edge_local[inte, 1]])
        return np.stack([cells, locs], axis=1).astype(np.int32)

    def locate_dofs_geometrical(self, func):
        """Strong-BC DOF set: ``locate_dofs_geometrical`` on both sub-spaces
        (reference rm_shell_model.py:168-180): every P2 node (u) and every vertex
        (theta) whose coordinates satisfy ``func``."""
        m2 = self._eval_marker(func, self.p2_coords)
        mv = m2[self.nV: self.nV + self.nE] if self.element == "CG2CR1" else m2[: self.nV]       # rotation nodes: edge midpoints / vertices
        un = np.nonzero(m2)[0]
        tn = np.nonzero(mv)[0]
        ud = (3 * un[:, None] + np.arange(3)[None, :]).ravel()
        td = (self.ndof_u + 3 * tn[:, None] + np.arange(3)[None, :]).ravel()
        return np.concatenate([ud, td]).astype(np.int32)

    # ------------------------------------------------------------------ partitioning
    def renumbered(self):
        """A copy of the mesh with cells along a Morton curve of their centroids and vertices in order of first
        appearance -- what dolfinx does to every mesh it reads (cells and nodes are reordered for locality; the
        reference maps back through ``mesh.topology.original_cell_index``, rm_shell_model.py:116) and what keeps the
        nodal gathers of the element kernels inside few cache lines when the caller's numbering is arbitrary.

        Returns ``(mesh, vertex_of_new, cell_of_new)``: ``mesh.nodes[i] == self.nodes[vertex_of_new[i]]`` and new cell
        ``e`` is old cell ``cell_of_new[e]``."""
        ctr = self.nodes[self.cells].mean(axis=1)
        lo, hi = ctr.min(axis=0), ctr.max(axis=0)
        q = ((ctr - lo) / np.maximum(hi - lo, 1e-300) * 1023.999).astype(np.uint64)

        def spread(v):                      # 10 bits -> every third bit
            v = (v | (v << np.uint64(16))) & np.uint64(0x030000FF)
            v = (v | (v << np.uint64(8))) & np.uint64(0x0300F00F)
            v = (v | (v << np.uint64(4))) & np.uint64(0x030C30C3)
            v = (v | (v << np.uint64(2))) & np.uint64(0x09249249)
            return v
        key = spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1)) | (spread(q[:, 2]) << np.uint64(2))
        cell_of_new = np.argsort(key, kind="stable")
        flat = self.cells[cell_of_new].ravel()
        _, first = np.unique(flat, return_index=True)
        vertex_of_new = flat[np.sort(first)]
        new_of_vertex = np.empty(self.nn, dtype=np.int64)
        new_of_vertex[vertex_of_new] = np.arange(self.nn)
        if vertex_of_new.size != self.nn:
            raise ValueError("mesh has vertices that belong to no cell")
        return ShellMesh(self.nodes[vertex_of_new], new_of_vertex[self.cells[cell_of_new]], self.element), vertex_of_new, cell_of_new

    def partition_cells(self, nparts):
        """Deterministic recursive coordinate bisection of cell centroids into
        ``nparts`` (a power of two) element sets -- SURVEY.md section 8e."""
        if nparts < 1 or nparts & (nparts - 1):
            raise ValueError("nparts must be a power of two")
        cent = self.nodes[self.cells].mean(axis=1)
        part = np.zeros(self.nel, dtype=np.int32)

        def split(idx, lo, n):
            if n == 1:
                part[idx] = lo
                return
            c = cent[idx]
            ax = int(np.argmax(c.max(axis=0) - c.min(axis=0)))
            order = idx[np.argsort(c[:, ax], kind="stable")]
            half = order.size // 2
            split(order[:half], lo, n // 2)
            split(order[half:], lo + n // 2, n // 2)

        split(np.arange(self.nel), 0, nparts)
        return part


# ---------------------------------------------------------------------- generators
def plate_mesh(width=2.0, length=10.0, nw=4, nl=20, element="CG2CG1"):
    """Flat rectangular plate, ``nl`` quads along x in [0,length], ``nw`` along y in
    [0,width]; the regenerable stand-in for the reference's LFS-only
    ``plate_2_10_quad_{nw}_{nl}`` meshes (ex_simple_shell_opt.py:27-30,42-43), clamped
    at ``x[0] <= 0`` there (:52-53)."""
    xs = np.linspace(0.0, length, nl + 1)
    ys = np.linspace(0.0, width, nw + 1)
    X, Y = np.meshgrid(xs, ys, indexing="ij")
    nodes = np.stack([X.ravel(), Y.ravel(), np.zeros(X.size)], axis=1)
    idx = np.arange((nl + 1) * (nw + 1)).reshape(nl + 1, nw + 1)
    cells = np.stack([idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(),
                      idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()], axis=1)
    return ShellMesh(nodes, cells, element)


def wing_skin_mesh(nc=116, ns=580, chord=1.2, span=6.0, jitter=0.2, shuffle=True,
                   seed_jitter=1, seed_perm=2, element="CG2CG1"):
    """Synthetic 'wing-skin' surface: an ``nc x ns``
    quad grid mapped to a cambered, tapered, twisted surface z = c(x,y), interior
    vertices jittered by ``jitter * h_cell * U(-1,1)``, then cells and nodes randomly
    renumbered so that no structured locality survives.  Root edge is y = 0."""
    s = np.linspace(0.0, 1.0, nc + 1)          # chordwise
    t = np.linspace(0.0, 1.0, ns + 1)          # spanwise
    S, T = np.meshgrid(s, t, indexing="ij")
    if jitter > 0:
        rng = np.random.default_rng(seed_jitter)
        dS = jitter / nc * rng.uniform(-1, 1, S.shape)
        dT = jitter / ns * rng.uniform(-1, 1, T.shape)
        dS[[0, -1], :] = 0; dS[:, [0, -1]] = 0
        dT[[0, -1], :] = 0; dT[:, [0, -1]] = 0
        S, T = S + dS, T + dT
    taper = 1.0 - 0.55 * T
    sweep = 0.35 * span * T * 0.25
    xloc = (S - 0.25) * chord * taper
    camber = 0.06 * chord * taper * 4.0 * S * (1.0 - S)
    twist = np.deg2rad(-4.0) * T
    x = sweep + xloc * np.cos(twist) + camber * np.sin(twist)
    z = -xloc * np.sin(twist) + camber * np.cos(twist) + 0.03 * span * T ** 2
    y = span * T
    nodes = np.stack([x.ravel(), y.ravel(), z.ravel()], axis=1)
    idx = np.arange((nc + 1) * (ns + 1)).reshape(nc + 1, ns + 1)
    cells = np.stack([idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(),
                      idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()], axis=1)
    if shuffle:
        rng = np.random.default_rng(seed_perm)
        pn = rng.permutation(nodes.shape[0])       # new id of old node i
        inv = np.empty_like(pn); inv[pn] = np.arange(pn.size)
        nodes = nodes[inv]
        cells = pn[cells]
        cells = cells[rng.permutation(cells.shape[0])]
    return ShellMesh(nodes, cells, element)


def _wing_surface(S, T, chord, span):
    """The cambered, tapered, twisted, swept surface of ``wing_skin_mesh`` at the parameter points (S chordwise, T spanwise)."""
    taper = 1.0 - 0.55 * T
    xloc = (S - 0.25) * chord * taper
    camber = 0.06 * chord * taper * 4.0 * S * (1.0 - S)
    twist = np.deg2rad(-4.0) * T
    x = 0.35 * span * T * 0.25 + xloc * np.cos(twist) + camber * np.sin(twist)
    z = -xloc * np.sin(twist) + camber * np.cos(twist) + 0.03 * span * T ** 2
    return np.stack([np.ravel(x), np.ravel(span * T), np.ravel(z)], axis=1)


def skin_triangulation(nc, ns, jitter=0.35, seed_jitter=1, tri=None):
    """Parameter points (S, T) of an (nc + 1) x (ns + 1) grid jittered by ``jitter`` cells (boundary points slide along their edge
    only) and their Delaunay triangulation in cell units, counter-clockwise, slivers between collinear boundary points dropped.
    ``tri``: a triangulation of these points computed earlier (the goldens carry theirs: qhull's output may differ between
    versions, the points do not)."""
    rng = np.random.default_rng(seed_jitter)
    S, T = np.meshgrid(np.linspace(0.0, 1.0, nc + 1), np.linspace(0.0, 1.0, ns + 1), indexing="ij")
    dS = jitter / nc * rng.uniform(-1, 1, S.shape)
    dT = jitter / ns * rng.uniform(-1, 1, T.shape)
    dS[[0, -1], :] = 0; dT[:, [0, -1]] = 0                 # boundary points slide along their edge only
    S, T = S + dS, T + dT
    if tri is None:
        from scipy.spatial import Delaunay
        # triangulate in cell units (isotropic), so that the Delaunay criterion sees the cells' real aspect
        tri = Delaunay(np.stack([S.ravel() * nc, T.ravel() * ns], axis=1)).simplices.astype(np.int64)
        st = np.stack([S.ravel(), T.ravel()], axis=1)
        a, b = st[tri[:, 1]] - st[tri[:, 0]], st[tri[:, 2]] - st[tri[:, 0]]
        area = a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]
        tri = tri[np.abs(area) > 1e-14 / (nc * ns)]             # slivers between collinear boundary points
        cw = area[np.abs(area) > 1e-14 / (nc * ns)] < 0
        tri[cw] = tri[cw][:, [0, 2, 1]]
    return S.ravel(), T.ravel(), np.asarray(tri, dtype=np.int64)


def _shuffled(nodes, cells, seed_perm, element):
    rng = np.random.default_rng(seed_perm)
    pn = rng.permutation(nodes.shape[0])                    # new id of old node i
    inv = np.empty_like(pn); inv[pn] = np.arange(pn.size)
    cells = pn[cells]
    return ShellMesh(nodes[inv], cells[rng.permutation(cells.shape[0])], element)


def unstructured_skin_mesh(nc=116, ns=580, chord=1.2, span=6.0, jitter=0.35, shuffle=True, seed_jitter=1, seed_perm=2,
                           element="CG2CG1", tri=None):
    """The wing-skin surface of ``wing_skin_mesh`` with an UNSTRUCTURED triangulation: the (nc + 1) x (ns + 1) parameter points are
    jittered by ``jitter`` cells and Delaunay-triangulated in the parameter plane (vertex valences 4..9, no mesh lines), then mapped to
    the cambered, tapered, twisted surface and renumbered at random.  Same vertex count as the quadrilateral skin (1 015 470 DOF at the
    default size, 134 560 triangles).  Needs scipy unless ``tri`` (``skin_triangulation``) is given -- a mesh generator for tests and
    benchmarks, not part of the solver path."""
    S, T, tri = skin_triangulation(nc, ns, jitter, seed_jitter, tri)
    nodes = _wing_surface(S, T, chord, span)
    return _shuffled(nodes, tri, seed_perm, element) if shuffle else ShellMesh(nodes, tri, element)


def unstructured_quad_skin_mesh(nc=47, ns=239, chord=1.2, span=6.0, jitter=0.35, shuffle=True, seed_jitter=1, seed_perm=2,
                                element="CG2CG1", tri=None):
    """An UNSTRUCTURED ALL-QUADRILATERAL wing skin -- what the reference's real shells are
    (examples/advanced_examples/lpc_gust_response_opt/ex_lpc_gust_response_opt.py:142-153: 39 488 quadrilaterals from a CAD mesher):
    the Delaunay triangulation of ``skin_triangulation`` with every triangle cut into three quadrilaterals (vertex, edge midpoint,
    centroid, edge midpoint).  Vertex valences 3 (the centroids), 4 (the edge midpoints) and 8..18 (the triangulation's vertices), no
    mesh lines, every cell a kite (strongly non-affine: the bilinear map's Jacobian varies by a factor ~2 over a cell), new points placed
    ON the cambered / twisted surface (so the cells are warped as well), numbering shuffled.  At the default size: 67 398 cells,
    1 016 124 DOF -- the size of BASELINE config 3."""
    S, T, tri = skin_triangulation(nc, ns, jitter, seed_jitter, tri)
    nV = S.size
    e = np.sort(np.stack([tri, np.roll(tri, -1, axis=1)], axis=2).reshape(-1, 2), axis=1)       # edge k of a triangle: vertex k -> k + 1
    ue, inv = np.unique(e[:, 0] * nV + e[:, 1], return_inverse=True)
    ea, eb = ue // nV, ue % nV
    mid = nV + inv.reshape(-1, 3)                                                                 # midpoint node of every triangle edge
    ctr = nV + ue.size + np.arange(tri.shape[0])
    Sx = np.concatenate([S, 0.5 * (S[ea] + S[eb]), S[tri].mean(axis=1)])
    Tx = np.concatenate([T, 0.5 * (T[ea] + T[eb]), T[tri].mean(axis=1)])
    quads = np.concatenate([np.stack([tri[:, k], mid[:, k], ctr, mid[:, (k + 2) % 3]], axis=1) for k in range(3)])
    nodes = _wing_surface(Sx, Tx, chord, span)
    return _shuffled(nodes, quads, seed_perm, element) if shuffle else ShellMesh(nodes, quads, element)


def tee_beam_mesh(width=1.0, height=0.5, length=5.0, nw=4, nh=2, nl=10):
    """A T-section: a flange plate (x along the length, y across the width, z = 0) with a web standing on its centre
    line (y = 0, 0 <= z <= height).  The edges along the junction are shared by three cells -- the smallest instance
    of the branching surfaces (skin + ribs + spars) that the reference's wing meshes are made of.  nw must be even."""
    if nw % 2:
        raise ValueError("nw must be even (the web stands on a mesh line)")
    xs = np.linspace(0.0, length, nl + 1)
    ys = np.linspace(-width / 2, width / 2, nw + 1)
    zs = np.linspace(0.0, height, nh + 1)
    fl = np.arange((nl + 1) * (nw + 1)).reshape(nl + 1, nw + 1)
    nodes = [np.stack([np.repeat(xs, nw + 1), np.tile(ys, nl + 1), np.zeros((nl + 1) * (nw + 1))], axis=1)]
    cells = [np.stack([fl[:-1, :-1].ravel(), fl[1:, :-1].ravel(), fl[1:, 1:].ravel(), fl[:-1, 1:].ravel()], axis=1)]
    # web nodes above the flange (row 0 of the web is the flange's centre line)
    wb = np.empty((nl + 1, nh + 1), dtype=np.int64)
    wb[:, 0] = fl[:, nw // 2]
    wb[:, 1:] = fl.size + np.arange((nl + 1) * nh).reshape(nl + 1, nh)
    nodes.append(np.stack([np.repeat(xs, nh), np.zeros((nl + 1) * nh), np.tile(zs[1:], nl + 1)], axis=1))
    cells.append(np.stack([wb[:-1, :-1].ravel(), wb[1:, :-1].ravel(), wb[1:, 1:].ravel(), wb[:-1, 1:].ravel()], axis=1))
    return ShellMesh(np.vstack(nodes), np.vstack(cells))


def quads_to_triangles(mesh: ShellMesh):
    """Split every quad along its 0-2 diagonal (triangle variant of a config)."""
    c = mesh.cells
    tri = np.vstack([c[:, [0, 1, 2]], c[:, [0, 2, 3]]])
    return ShellMesh(mesh.nodes, tri, mesh.element)
