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
    def __init__(self, sub, plan, info, element_wise_material=False, nquad=None):
        import torch
        from oracle.rm_shell_oracle import ShellOracle
        self.torch = torch
        self.sub, self.plan, self.info = sub, plan, info
        self.ewm = element_wise_material
        self.nloc = sub.ndof
        self.nvec = sub.ndof + info["nghost"]
        self.v = {n: torch.zeros(self.nvec, dtype=torch.float64) for n in ("state", "adjoint", "r", "z", "p", "Ap", "b")}
        self.tmp = np.zeros(self.nvec)
        self.nquad = nquad                 # the rule of the whole mesh, the same on every rank
        self.oracle = ShellOracle(sub, element_wise_material=element_wise_material, nquad=nquad)
        self.pf, self.beta = None, 1e15
        self.F, self.L = None, None
        self.topbuf = self.scal = None

    def on_stream(self):
        import contextlib
        return contextlib.nullcontext()

    def vec(self, name):
        return self.v[name]

    def new_tensor(self, n):
        return self.torch.zeros(n, dtype=self.torch.float64)

    def dist_setup(self, top_local, nranks, n_local_levels, sel):
        self.top = np.asarray(top_local, dtype=np.int64)
        self.nl, self.nlev = int(n_local_levels), self.plan.nlevels
        self.sel = np.asarray(sel, dtype=np.int64)
        self.w = np.ones(self.nvec)
        self.w[self.top] = 1.0 / nranks
        self.topbuf = self.torch.zeros(self.top.size + 1, dtype=self.torch.float64)
        self.scal = self.torch.zeros(8, dtype=self.torch.float64)

    # ------------------------------------------------------------------ the phases of the partitioned PCG (femo_dist_*)
    def pack(self, vec):
        self.topbuf[:-1] = self.v[vec][self.top]

    def unpack(self, vec):
        self.v[vec][self.top] = self.topbuf[:-1]

    def pcg_start(self, b, x):
        self.v[x].zero_()
        self.v["r"].copy_(self.v[b])
        bn = self.v[b].numpy()
        self.scal.zero_()
        self.scal[3] = float(np.sum(self.w * bn * bn))

    def precond_fwd(self):
        self.v["z"].copy_(self.v["r"])
        self._save = self.v["z"][self.top].clone()
        self.sweep("z", 0, self.nl, False)
        self.topbuf[:-1] = self.v["z"][self.top] - self._save
        self.topbuf[-1] = self.scal[3]

    def read(self):
        return float(self.topbuf[-1]), float(self.scal[2])

    def precond_rest(self):
        self.v["z"][self.top] = self._save + self.topbuf[:-1]
        self.sweep("z", self.nl, self.nlev, False)
        self.sweep("z", self.nl, self.nlev, True)
        self.sweep("z", 0, self.nl, True)
        self.scal[1:4] = 0.0
        self.scal[1] = float(np.sum(self.w * self.v["r"].numpy() * self.v["z"].numpy()))

    def direction_apply(self, first):
        if first:
            self.v["p"].copy_(self.v["z"])
        else:
            self.v["p"].mul_(float(self.scal[1] / self.scal[0])).add_(self.v["z"])
        self.apply("p", "Ap")
        self.scal[2] = float(self.v["p"].numpy() @ self.v["Ap"].numpy())
        self.topbuf[:-1] = self.v["Ap"][self.top]
        self.topbuf[-1] = self.scal[2]

    def update(self, x):
        self.v["Ap"][self.top] = self.topbuf[:-1]
        self.scal[2] = self.topbuf[-1]
        alpha = float(self.scal[1] / self.scal[2])
        self.v[x].add_(self.v["p"], alpha=alpha)
        self.v["r"].add_(self.v["Ap"], alpha=-alpha)
        rn = self.v["r"].numpy()
        self.scal[3] = float(np.sum(self.w * rn * rn))
        self.scal[0] = self.scal[1]

    def gradient(self, functional, arg, lam, gglob):
        gglob[self.sel] = self.torch.as_tensor(self.field_gradient_vec(functional, arg, lam))

    def set_field(self, name, values):
        key = {"thickness": "h", "E": "E", "nu": "nu", "density": "rho", "F_solid": "f"}[name]
        self.oracle.set_fields(**{key: values})

    def set_penalty_facets(self, pairs, beta):
        from oracle.rm_shell_oracle import ShellOracle
        o = self.oracle
        self.oracle = ShellOracle(self.sub, element_wise_material=self.ewm, penalty_facets=pairs, beta=beta, nquad=self.nquad)
        self.oracle.set_fields(h=o.h, E=o.E, nu=o.nu, rho=o.rho, f=o.f)

    def set_quadrature(self, nquad):
        from oracle.rm_shell_oracle import ShellOracle
        o = self.oracle
        self.nquad = int(nquad)
        self.oracle = ShellOracle(self.sub, element_wise_material=self.ewm, penalty_facets=o.penalty_facets, beta=o.beta, nquad=self.nquad)
        self.oracle.set_fields(h=o.h, E=o.E, nu=o.nu, rho=o.rho, f=o.f)
        self.F = self.L = None

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

    def schur_pack(self, front, out):
        """Lower triangle, column by column (the layout of femo_front_schur_pack)."""
        p = self.plan
        S = self.F[front][p.npiv[front]:, p.npiv[front]:]
        n = S.shape[0]
        r, c = np.tril_indices(n)
        o = np.lexsort((r, c))                      # column-major order of the lower triangle
        out[: r.size] = self.torch.as_tensor(S[r[o], c[o]])

    def block_unpack(self, front, src):
        n = self.plan.nf[front]
        r, c = np.tril_indices(n)
        o = np.lexsort((r, c))
        A = np.zeros((n, n))
        A[r[o], c[o]] = src.numpy()[: r.size]
        self.F[front] = A + np.tril(A, -1).T

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
    if kind == "wing1m":           # BASELINE config 3 / 4: the bench workload itself (1 015 470 DOF)
        from bench import make_workload
        m, fields, marker, _ = make_workload("wing1m")
        return m, marker, {k: np.asarray(v, dtype=np.float64) for k, v in fields.items()}
    if kind == "wing":
        m = wing_skin_mesh(8, 24, shuffle=True)
        marker = lambda x: np.less(x[1], 1e-12)
    elif kind == "wing_tri_nu":    # triangles with a nodal Poisson ratio: the rule of the whole mesh is decided on the global field
        from femo_alpha_amd.mesh import quads_to_triangles
        m = quads_to_triangles(wing_skin_mesh(8, 24, shuffle=True))
        marker = lambda x: np.less(x[1], 1e-12)
    elif kind == "wing_cr":        # ShellElement 'CG2CR1' (linear_shell_model.py:68-73): triangles, rotation on the edge midpoints
        from femo_alpha_amd.mesh import ShellMesh, quads_to_triangles
        t = quads_to_triangles(wing_skin_mesh(8, 24, shuffle=True))
        m = ShellMesh(t.nodes, t.cells, "CG2CR1")
        marker = lambda x: np.less(x[1], 1e-12)
    else:
        m = plate_mesh(2.0, 10.0, 6, 24)
        marker = lambda x: np.less(x[0], 3e-16)
    rng = np.random.default_rng(7)
    fields = dict(thickness=0.02 * (1 + 0.2 * rng.uniform(-1, 1, m.nn)), E=np.array([7e9]), nu=np.array([0.3]),
                  density=np.array([2700.0]), F_solid=rng.uniform(-1, 1, (m.nn, 3)) * 50)
    if kind == "wing_tri_nu":
        fields["nu"] = 0.3 * (1 + 0.3 * rng.uniform(-1, 1, m.nn))       # a nodal Poisson ratio that varies: every rank must take degree 9
    return m, marker, fields


def reference_solution(m, marker, fields):
    from oracle.rm_shell_oracle import ShellOracle
    o = ShellOracle(m, penalty_facets=m.penalty_facets(marker))
    o.set_fields(h=fields["thickness"], E=fields["E"], nu=fields["nu"], rho=fields["density"], f=fields["F_solid"])
    w, J, dJ = o.forward_adjoint()
    return w, J, dJ, o.mass()


def worker(rank, world, port, kind, engine, result_path, backend="gloo"):
    """Entry point of one spawned rank: run the distributed driver, rank 0 stores the results.  ``backend`` "nccl" (= RCCL) wants one
    device per rank: on a one-GPU box only world 1 -- which still sends every collective of the driver through RCCL on device
    tensors, in stream order with the library's launches (the gloo path copies through the host and would hide an ordering bug)."""
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    if backend == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from femo_alpha_amd.parallel import Comm, DistributedShell
        m, marker, fields = make_case(kind)
        factory = (lambda sub, plan, info: NumpyEngine(sub, plan, info, nquad=m.recommended_nquad())) if engine == "numpy" else None
        ds = DistributedShell(m, Comm(dist), bc_marker=marker, leaf_size=12 if kind == "wing1m" else 4, engine_factory=factory, device=0)
        res = run_driver(ds, fields)
        if rank == 0:
            np.savez(result_path, **res)
    finally:
        dist.destroy_process_group()


def run_driver(ds, fields, rtol=1e-12):
    """Forward solve, outputs and adjoint gradient of one rank's driver; everything a test compares."""
    ds.rtol = rtol
    ds.set_fields(**fields)
    it, rel = ds.solve_state()
    w = ds.gather_state()
    J, M = ds.functional("compliance"), ds.functional("mass")
    g, it2, rel2 = ds.total_gradient("compliance", "thickness")
    return dict(w=w, J=J, M=M, g=g, it=it, rel=rel, it2=it2, rel2=rel2, nghost=ds.info["nghost"], ntop=ds.info["n_top"], nquad=ds.nquad)


# ---- several ranks inside ONE process (threads): a GPU box admits at most 6 processes on its card, so the 8-partition case of
# BASELINE config 4 runs its eight drivers as threads, each with its own context and HIP stream on the one GPU.  The
# collectives are sums over the threads' host copies in rank order (the same order on every rank, so all ranks see
# bit-identical results, as with a real all-reduce).
class ThreadGroup:
    def __init__(self, n):
        import threading
        self.n = n
        self.barrier = threading.Barrier(n)
        self.slots = [None] * n


class ThreadComm:
    def __init__(self, group, rank):
        self.g, self.rank, self.size = group, rank, group.n

    def _exchange(self, h):
        self.g.slots[self.rank] = h
        self.g.barrier.wait()
        parts = list(self.g.slots)
        self.g.barrier.wait()               # nobody overwrites its slot before everyone has read all of them
        return parts

    def allreduce_(self, t, op="sum"):
        import torch
        parts = self._exchange(t.detach().cpu().clone())
        tot = parts[0].clone()
        for q in parts[1:]:
            tot = torch.maximum(tot, q) if op == "max" else tot + q
        t.copy_(tot)
        return t

    def allgather(self, t):
        import torch
        return torch.stack(self._exchange(t.detach().cpu().clone())).to(t.device)


def run_threads(world, m, marker, fields, leaf_size=12, rtol=1e-12, device=0):
    """``world`` drivers with the HIP engine as threads of this process; returns rank 0's results."""
    import threading
    from femo_alpha_amd.parallel import DistributedShell
    from femo_alpha_amd.solver.symbolic import analyse
    tree = analyse(m, leaf_size, min_depth=int(np.log2(world)))
    group = ThreadGroup(world)
    out, err = [None] * world, [None] * world

    def body(rank):
        try:
            ds = DistributedShell(m, ThreadComm(group, rank), bc_marker=marker, leaf_size=leaf_size, device=device, tree=tree)
            out[rank] = run_driver(ds, fields, rtol)
            ds.eng.ctx.close()
        except BaseException as e:          # noqa: BLE001 -- a dead rank must not leave the others waiting at the barrier
            err[rank] = e
            group.barrier.abort()

    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for e in err:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    for e in err:
        if e is not None:
            raise e
    return out[0]
