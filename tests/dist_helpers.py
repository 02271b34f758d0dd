"""Helpers for the multi-process tests of femo_alpha_amd.parallel.

``NumpyEngine`` is a CPU stand-in for ``HipEngine`` with the same interface: the local operator is the
CPU oracle on the rank's sub-mesh and the multifrontal numeric phase is emulated with dense numpy on
the rank's plan.  It lets the world_size > 1 logic (partition, replicated separator DOFs, Schur
all-gather, all-reduces, distributed PCG) run under ``gloo`` without a GPU.  Test infrastructure only.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class NumpyEngine:
    def __init__(self, sub, plan, info, element_wise_material=False):
        import torch
        from oracle.rm_shell_oracle import ShellOracle
        self.torch = torch
        self.sub, self.plan, self.info = sub, plan, info
        self.ewm = element_wise_material
        self.nloc = sub.ndof
        self.nvec = sub.ndof + info["nghost"]
        self.v = {n: torch.zeros(self.nvec, dtype=torch.float64) for n in ("state", "adjoint", "r", "z", "p", "Ap", "b")}
        self.tmp = np.zeros(self.nvec)
        self.oracle = ShellOracle(sub, element_wise_material=element_wise_material)
        self.pf, self.beta = None, 1e15
        self.F, self.L = None, None

    def vec(self, name):
        return self.v[name]

    def new_tensor(self, n):
        return self.torch.zeros(n, dtype=self.torch.float64)

    def set_field(self, name, values):
        key = {"thickness": "h", "E": "E", "nu": "nu", "density": "rho", "F_solid": "f"}[name]
        self.oracle.set_fields(**{key: values})

    def set_penalty_facets(self, pairs, beta):
        from oracle.rm_shell_oracle import ShellOracle
        o = self.oracle
        self.oracle = ShellOracle(self.sub, element_wise_material=self.ewm, penalty_facets=pairs, beta=beta)
        self.oracle.set_fields(h=o.h, E=o.E, nu=o.nu, rho=o.rho, f=o.f)

    # ------------------------------------------------------------------ operator
    def apply(self, src, dst):
        y = np.zeros(self.nvec)
        y[: self.nloc] = self.oracle.apply_K(self.v[src].numpy()[: self.nloc])
        self.v[dst].copy_(self.torch.as_tensor(y))

    def load(self, dst):
        y = np.zeros(self.nvec)
        y[: self.nloc] = self.oracle.load_vector()
        self.v[dst].copy_(self.torch.as_tensor(y))

    # ------------------------------------------------------------------ dense emulation of the numeric phase
    def factor(self, l0, l1, assemble):
        p = self.plan
        if assemble:
            self.F = [np.zeros((p.nf[t], p.nf[t])) for t in range(p.ntree)]
            self.L = [None] * p.ntree
            Ke = self.oracle.element_matrices()
            for e in range(self.sub.nel):
                m = p.elem_map[e]
                self.F[p.elem_front[e]][np.ix_(m, m)] += Ke[e]
            cd = self.sub.cell_dofs()
            for dofs, blk in self.oracle._penalty_blocks():
                # the facet belongs to one cell: find it through the first DOF
                e = int(np.nonzero((cd == dofs[0]).any(axis=1) & (cd == dofs[-1]).any(axis=1))[0][0])
                t = p.elem_front[e]
                fd = p.front_dofs[p.dof_off[t]:p.dof_off[t + 1]]
                pos = np.array([int(np.nonzero(fd == d)[0][0]) for d in dofs])
                self.F[t][np.ix_(pos, pos)] += blk
        for lv in range(l0, l1):
            for t in p.level_nodes[lv]:
                for c in (p.left[t], p.right[t]):
                    if c >= 0:
                        up = p.up_map[p.dof_off[c] + p.npiv[c]:p.dof_off[c + 1]]
                        self.F[t][np.ix_(up, up)] += self.F[c][p.npiv[c]:, p.npiv[c]:]
                n = p.npiv[t]
                if n == 0:
                    self.L[t] = (np.zeros((0, 0)), np.zeros((p.nf[t], 0)))
                    continue
                L11 = np.linalg.cholesky(self.F[t][:n, :n])
                L21 = np.linalg.solve(L11, self.F[t][n:, :n].T).T
                self.F[t][n:, n:] -= L21 @ L21.T
                self.L[t] = (L11, L21)

    def schur_get(self, front, out):
        p = self.plan
        S = self.F[front][p.npiv[front]:, p.npiv[front]:]
        out[: S.size] = self.torch.as_tensor(S.ravel())

    def block_set(self, front, src):
        # the driver hands over one triangle of the (symmetric) Schur complement; this engine works on full matrices
        n = self.plan.nf[front]
        A = src.numpy()[: n * n].reshape(n, n)
        self.F[front] = np.tril(A) + np.tril(A, -1).T if np.abs(np.triu(A, 1)).max() == 0 else np.triu(A) + np.triu(A, 1).T

    def sweep(self, vec, l0, l1, backward):
        p = self.plan
        v = self.v[vec].numpy()
        if not backward:
            for lv in range(l0, l1):
                for t in p.level_nodes[lv]:
                    d = p.front_dofs[p.dof_off[t]:p.dof_off[t + 1]]; n = p.npiv[t]
                    if n == 0:
                        continue
                    y = np.linalg.solve(self.L[t][0], v[d[:n]])
                    self.tmp[d[:n]] = y
                    np.subtract.at(v, d[n:], self.L[t][1] @ y)
        else:
            for lv in range(l1 - 1, l0 - 1, -1):
                for t in p.level_nodes[lv]:
                    d = p.front_dofs[p.dof_off[t]:p.dof_off[t + 1]]; n = p.npiv[t]
                    if n == 0:
                        continue
                    v[d[:n]] = np.linalg.solve(self.L[t][0].T, self.tmp[d[:n]] - self.L[t][1].T @ v[d[n:]])

    # ------------------------------------------------------------------ outputs
    def functionals_partial(self):
        w = self.v["state"].numpy()[: self.nloc]
        reg = self.oracle.regularization()
        return np.array([self.oracle.compliance(w) - reg, reg, self.oracle.mass()])

    def dfunctional_vec(self, name, dst):
        assert name == "compliance"
        y = np.zeros(self.nvec)
        y[: self.nloc] = self.oracle.dcompliance_du(self.v["state"].numpy()[: self.nloc])
        self.v[dst].copy_(self.torch.as_tensor(y))

    def field_gradient_vec(self, functional, arg, lam):
        assert functional == "compliance" and arg == "thickness"
        w = self.v["state"].numpy()[: self.nloc]
        l = self.v[lam].numpy()[: self.nloc]
        return self.oracle.dcompliance_dh(w) - self.oracle.dRdfield_T("h", w, l)


def make_case(kind="wing"):
    from femo_alpha_amd.mesh import plate_mesh, wing_skin_mesh
    if kind == "wing":
        m = wing_skin_mesh(8, 24, shuffle=True)
        marker = lambda x: np.less(x[1], 1e-12)
    else:
        m = plate_mesh(2.0, 10.0, 6, 24)
        marker = lambda x: np.less(x[0], 3e-16)
    rng = np.random.default_rng(7)
    fields = dict(thickness=0.02 * (1 + 0.2 * rng.uniform(-1, 1, m.nn)), E=np.array([7e9]), nu=np.array([0.3]),
                  density=np.array([2700.0]), F_solid=rng.uniform(-1, 1, (m.nn, 3)) * 50)
    return m, marker, fields


def reference_solution(m, marker, fields):
    from oracle.rm_shell_oracle import ShellOracle
    o = ShellOracle(m, penalty_facets=m.penalty_facets(marker))
    o.set_fields(h=fields["thickness"], E=fields["E"], nu=fields["nu"], rho=fields["density"], f=fields["F_solid"])
    w, J, dJ = o.forward_adjoint()
    return w, J, dJ, o.mass()


def worker(rank, world, port, kind, engine, result_path):
    """Entry point of one spawned rank: run the distributed driver, rank 0 stores the results."""
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from femo_alpha_amd.parallel import Comm, DistributedShell
        m, marker, fields = make_case(kind)
        factory = (lambda sub, plan, info: NumpyEngine(sub, plan, info)) if engine == "numpy" else None
        ds = DistributedShell(m, Comm(dist), bc_marker=marker, leaf_size=4, engine_factory=factory, device=0)
        ds.rtol = 1e-12
        ds.set_fields(**fields)
        it, rel = ds.solve_state()
        w = ds.gather_state()
        J, M = ds.functional("compliance"), ds.functional("mass")
        g, it2, rel2 = ds.total_gradient("compliance", "thickness")
        if rank == 0:
            np.savez(result_path, w=w, J=J, M=M, g=g, it=it, rel=rel, it2=it2, rel2=rel2,
                     nghost=ds.info["nghost"], ntop=ds.info["n_top"])
    finally:
        dist.destroy_process_group()
