"""Element-partitioned (multi-GPU) forward + adjoint driver: one process per GPU, collectives through
``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in the CPU tests).

The reference has no multi-rank path (meshes are read on ``MPI.COMM_SELF``, reference
femo_alpha/fea/utils_dolfinx.py:41,45); this is the build's own design:

* **partition** -- the 2^d subtrees at depth d of the nested-dissection tree of the elements
  (``solver.symbolic.rank_plan``); a rank owns its elements and every DOF only they touch; the
  separator DOFs above depth d are *replicated* on all ranks (a few thousand entries);
* **operator** -- local element-by-element apply, then ONE all-reduce (sum) over the replicated
  entries: the "RCCL all-reduce of halo DOFs" of the north star;
* **preconditioner** -- each rank factorises its own subtree; the Schur complements of the 2^d
  subtree roots are all-gathered (dense, (2 x separator)^2 doubles each) and the small top of the tree
  is factorised redundantly by every rank, so a solve needs one more all-reduce (of the replicated
  right-hand-side entries between the local and the top part of the forward sweep) and nothing else;
* **dots** -- interior entries summed over ranks (one packed scalar all-reduce), replicated entries
  counted once.

PCG then converges in the same 2-3 iterations as on one GPU.  The Krylov loop lives here, in Python,
on zero-copy torch views of the context's device vectors: a solve is a handful of collectives and
kernel launches, so host orchestration costs microseconds against milliseconds of device work.
"""
from __future__ import annotations

import numpy as np

from .solver.symbolic import analyse, rank_plan

__all__ = ["Comm", "HipEngine", "DistributedShell"]


class Comm:
    """Rank/size + the three collectives the driver needs; degenerates gracefully to one rank."""

    def __init__(self, dist=None):
        self.dist = dist if (dist is not None and dist.is_initialized()) else None
        self.rank = self.dist.get_rank() if self.dist else 0
        self.size = self.dist.get_world_size() if self.dist else 1
        self._stage = self.dist is not None and self.dist.get_backend() == "gloo"

    def allreduce_(self, t, op="sum"):
        """In-place sum (or max) over ranks; device tensors are staged through the host under gloo."""
        if self.dist is None:
            return t
        rop = self.dist.ReduceOp.MAX if op == "max" else self.dist.ReduceOp.SUM
        if self._stage and t.is_cuda:
            h = t.cpu()
            self.dist.all_reduce(h, op=rop)
            t.copy_(h)
        else:
            self.dist.all_reduce(t, op=rop)
        return t

    def allgather(self, t):
        """(size, len(t)) tensor of every rank's ``t`` (equal lengths): one ``all_gather_into_tensor`` over RCCL; gloo (the
        CPU tests and rehearsals, which stage device tensors through the host) has no such call and gathers a list."""
        if self.dist is None:
            return t.reshape(1, -1)
        import torch
        if self._stage:
            h = t.cpu() if t.is_cuda else t
            parts = [torch.empty_like(h) for _ in range(self.size)]
            self.dist.all_gather(parts, h)
            return torch.stack(parts).to(t.device)
        out = torch.empty((self.size, t.numel()), dtype=t.dtype, device=t.device)
        self.dist.all_gather_into_tensor(out, t)
        return out


class HipEngine:
    """The local pieces on one MI355X (a ``ShellContext`` on the rank's sub-mesh)."""

    def __init__(self, sub, plan, info, device=0, element_wise_material=False, elementwise_pressure=False):
        import torch
        from .backend import ShellContext
        self.torch = torch
        self.ctx = ShellContext(sub, element_wise_material, elementwise_pressure, device=device, nghost=info["nghost"])
        self.ctx.enable_frontal(plan=plan)
        self.ctx.set_solver(preconditioner=2, rtol=1e-10, maxit=50, check_every=1)
        self.device = torch.device("cuda", device)
        self.nvec = self.ctx.ndof
        self._views = {}
        # torch's tensor ops and collectives run on the context's own stream: every library call is synchronous on return and
        # everything torch enqueues afterwards is ordered behind it, so no device-wide synchronisation is needed in between
        self.stream = self.ctx.torch_stream()
        torch.cuda.set_stream(self.stream)

    def vec(self, name):
        if name not in self._views:
            self._views[name] = self.ctx.vec_tensor(name)
        return self._views[name]

    def new_tensor(self, n):
        return self.torch.zeros(n, dtype=self.torch.float64, device=self.device)

    def _sync(self):
        pass            # same stream on both sides (see __init__); kept as the seam where a two-stream engine would wait

    def set_field(self, name, values):
        self.ctx.set_field(name, values)

    def set_penalty_facets(self, pairs, beta):
        self.ctx.set_penalty_facets(pairs, beta)

    def apply(self, src, dst):
        self._sync(); self.ctx.op_apply_vec(src, dst)

    def load(self, dst):
        self._sync(); self.ctx.load_vec(dst)

    def factor(self, l0, l1, assemble):
        self._sync(); self.ctx.factorize_range(l0, l1, assemble); self.ctx.sync()

    def schur_get(self, front, out):
        self._sync(); self.ctx.front_schur_get(front, out)

    def block_set(self, front, src):
        self._sync(); self.ctx.front_block_set(front, src)

    def sweep(self, vec, l0, l1, backward):
        self._sync(); self.ctx.frontal_sweep(vec, l0, l1, backward)

    def functionals_partial(self):
        self._sync(); return self.ctx.functionals_partial()

    def dfunctional_vec(self, name, dst):
        self._sync(); self.ctx.dfunctional_vec(name, dst)

    def field_gradient_vec(self, functional, arg, lam):
        self._sync(); return self.ctx.field_gradient_vec(functional, arg, lam)


class DistributedShell:
    """Forward solve, scalar outputs and the adjoint gradient on an element partition."""

    def __init__(self, mesh, comm: Comm, bc_marker=None, beta=1.0e15, leaf_size=12, engine_factory=None,
                 element_wise_material=False, device=0, tree=None):
        import torch
        self.torch = torch
        self.mesh, self.comm = mesh, comm
        self.ewm = bool(element_wise_material)
        d = int(np.log2(comm.size))
        self.tree = analyse(mesh, leaf_size, min_depth=d) if tree is None else tree
        self.sub, self.plan, self.info = rank_plan(mesh, self.tree, comm.rank, comm.size)
        factory = engine_factory or (lambda sub, plan, info: HipEngine(sub, plan, info, device=device,
                                                                       element_wise_material=element_wise_material))
        self.eng = factory(self.sub, self.plan, self.info)
        if bc_marker is not None:
            self.eng.set_penalty_facets(self.sub.penalty_facets(bc_marker), beta)
        nvec = self.sub.ndof + self.info["nghost"]
        dev = self.eng.vec("r").device
        is_top = np.zeros(nvec, dtype=bool)
        is_top[self.info["top_local"]] = True
        self.top_idx = torch.as_tensor(self.info["top_local"], dtype=torch.int64, device=dev)
        self.int_idx = torch.as_tensor(np.nonzero(~is_top)[0], dtype=torch.int64, device=dev)
        # weights of the global dot product: 1 on the entries only this rank holds, 1 / size on the replicated ones (the number
        # of ranks is a power of two, so the weight is exact and the replicated entries add up to one copy in the all-reduce)
        wd = np.ones(is_top.size)
        wd[is_top] = 1.0 / comm.size
        self.wdot = torch.as_tensor(wd, dtype=torch.float64, device=dev) if comm.size > 1 else None
        self.nl, self.nlev = self.info["n_local_levels"], self.plan.nlevels
        self.rtol, self.maxit = 1e-10, 50
        self.factored = False
        self.last = {}

    # ------------------------------------------------------------------ inputs
    def set_fields(self, thickness=None, E=None, nu=None, density=None, F_solid=None):
        """Global arrays (caller numbering); each rank keeps the entries of its vertices / cells."""
        sel = self.info["cells"] if self.ewm else self.info["vertices"]
        for name, v in (("thickness", thickness), ("E", E), ("nu", nu), ("density", density)):
            if v is not None:
                v = np.asarray(v, dtype=np.float64).ravel()
                self.eng.set_field(name, v if v.size == 1 else v[sel])
        if F_solid is not None:
            f = np.asarray(F_solid, dtype=np.float64).reshape(-1, 3)
            self.eng.set_field("F_solid", f[self.info["vertices"]])
        self.factored = False

    # ------------------------------------------------------------------ collectives on replicated entries
    def _sum_top(self, name):
        v = self.eng.vec(name)
        buf = v.index_select(0, self.top_idx)
        self.comm.allreduce_(buf)
        v.index_copy_(0, self.top_idx, buf)

    def dot_t(self, a, b):
        """a . b over the global vector as a 0-dim device tensor (no host synchronisation): one weighted local dot and one
        scalar all-reduce; replicated entries carry the weight 1 / size, so they are counted once."""
        va, vb = self.eng.vec(a), self.eng.vec(b)
        if self.wdot is None:
            return self.torch.dot(va, vb)
        loc = self.torch.dot(va * self.wdot, vb).reshape(1)
        self.comm.allreduce_(loc)
        return loc[0]

    def dot(self, a, b):
        return float(self.dot_t(a, b))

    # ------------------------------------------------------------------ operator, preconditioner
    def apply(self, src, dst):
        self.eng.apply(src, dst)
        self._sum_top(dst)

    def factorize(self):
        eng, info = self.eng, self.info
        eng.factor(0, self.nl, True)
        if self.comm.size > 1:
            # the Schur complements of the subtree roots travel as packed lower triangles (they are symmetric and only the
            # lower triangle of a front is ever read), padded to the largest one: ONE all_gather_into_tensor
            sizes = info["schur_sizes"]
            n_me = sizes[self.comm.rank]
            full = eng.new_tensor(max(n_me * n_me, 1))
            eng.schur_get(info["root_front"], full)
            cap = max(n * (n + 1) // 2 for n in sizes)
            mine = eng.new_tensor(cap)
            il = self._tril(n_me)
            mine[: il.numel()] = full[il]
            gathered = self.comm.allgather(mine)
            for q in range(self.comm.size):
                if q != self.comm.rank:
                    n = sizes[q]
                    iq = self._tril(n)
                    blk = eng.new_tensor(n * n)
                    blk[iq] = gathered[q, : iq.numel()]
                    eng.block_set(info["stub_fronts"][q], blk)
        eng.factor(self.nl, self.nlev, False)
        self.factored = True

    def _tril(self, n):
        """Flat (column-major) indices of the lower triangle of an n x n block, cached per size."""
        cache = self.__dict__.setdefault("_tril_cache", {})
        if n not in cache:
            r, c = np.tril_indices(n)
            cache[n] = self.torch.as_tensor(np.sort(r + n * c), dtype=self.torch.int64, device=self.eng.vec("r").device)
        return cache[n]

    def precondition(self, name):
        """name <- (L L^T)^-1 name; replicated entries must agree on all ranks on entry."""
        eng = self.eng
        v = eng.vec(name)
        before = v.index_select(0, self.top_idx)
        eng.sweep(name, 0, self.nl, False)
        if self.comm.size > 1:
            delta = v.index_select(0, self.top_idx) - before
            self.comm.allreduce_(delta)
            v.index_copy_(0, self.top_idx, before + delta)
        eng.sweep(name, self.nl, self.nlev, False)
        eng.sweep(name, self.nl, self.nlev, True)
        eng.sweep(name, 0, self.nl, True)

    # ------------------------------------------------------------------ PCG
    def _pcg(self, b, x):
        eng = self.eng
        if not self.factored:
            self.factorize()
        vb, vx, vr, vz, vp, vAp = (eng.vec(n) for n in (b, x, "r", "z", "p", "Ap"))
        vx.zero_()
        vr.copy_(vb)
        bb = self.dot(b, b)
        rr, k, rz = bb, 0, None
        # the scalars of an iteration (r.z, p.Ap, alpha, beta) stay on the device; one host read per iteration: (r.r, p.Ap)
        while bb > 0 and rr > self.rtol ** 2 * bb and k < self.maxit:
            vz.copy_(vr)
            self.precondition("z")
            rz_new = self.dot_t("r", "z")
            if k == 0:
                vp.copy_(vz)
            else:
                vp.mul_(rz_new / rz).add_(vz)
            rz = rz_new
            self.apply("p", "Ap")
            pAp = self.dot_t("p", "Ap")
            alpha = rz / pAp
            vx.addcmul_(vp, alpha)
            vr.addcmul_(vAp, -alpha)
            rr_t = self.dot_t("r", "r")
            rr, pAp_h = self.torch.stack([rr_t, pAp]).tolist()
            if not pAp_h > 0:
                raise RuntimeError("PCG broke down: p.Ap <= 0")
            k += 1
        return k, (rr / bb) ** 0.5 if bb > 0 else 0.0

    def solve_state(self):
        """Forward solve K w = F (cold start); the state stays on the devices."""
        self.eng.load("b")
        self._sum_top("b")
        it, rel = self._pcg("b", "state")
        self.last["forward"] = (it, rel)
        return it, rel

    def functional(self, name):
        uu, reg, mass = self.eng.functionals_partial()
        t = self.eng.new_tensor(2)
        t.copy_(self.torch.tensor([uu + reg, mass], dtype=self.torch.float64))
        self.comm.allreduce_(t)
        if name == "compliance":
            return float(t[0])
        if name == "mass":
            return float(t[1])
        raise ValueError(name)

    def total_gradient(self, functional="compliance", arg="thickness"):
        """d functional / d arg on the caller's global numbering (summed over ranks)."""
        self.eng.dfunctional_vec(functional, "b")
        self._sum_top("b")
        it, rel = self._pcg("b", "adjoint")
        self.last["adjoint"] = (it, rel)
        g_loc = self.eng.field_gradient_vec(functional, arg, "adjoint")
        # the global gradient buffer and the index of this rank's entries in it live on the device for the life of the driver
        if getattr(self, "_grad_buf", None) is None:
            n_glob = self.mesh.nel if self.ewm else self.mesh.nn
            sel = self.info["cells"] if self.ewm else self.info["vertices"]
            self._grad_buf = self.eng.new_tensor(n_glob)
            self._grad_sel = self.torch.as_tensor(np.asarray(sel), dtype=self.torch.int64, device=self._grad_buf.device)
        g = self._grad_buf
        g.zero_()
        g[self._grad_sel] = self.torch.as_tensor(g_loc).to(g.device)
        self.comm.allreduce_(g)
        return g.cpu().numpy(), it, rel

    def gather_state(self):
        """Global state vector on every rank (tests / post-processing)."""
        w = self.eng.vec("state")
        out = self.eng.new_tensor(self.mesh.ndof)
        cnt = self.eng.new_tensor(self.mesh.ndof)
        l2g = self.torch.as_tensor(self.info["l2g_dof"], dtype=self.torch.int64, device=out.device)
        # interior entries are unique to this rank; replicated entries are taken from every rank and averaged
        out[l2g] = w
        cnt[l2g] = 1.0
        self.comm.allreduce_(out)
        self.comm.allreduce_(cnt)
        return (out / cnt.clamp(min=1.0)).cpu().numpy()
