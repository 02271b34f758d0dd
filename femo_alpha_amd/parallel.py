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

PCG then converges in the same 2-3 iterations as on one GPU.  The arithmetic of the Krylov loop is the
library's (``femo_dist_*``, include/femo_hip.h): fused vector kernels with device-resident scalars, the
replicated-index list and the dot weights resident in the context; every library call between two
collectives only enqueues work on the context's HIP stream.  Python keeps what only it can do -- the
collectives (``torch.distributed``; issued on that same stream, on zero-copy views of two small device
buffers) and the one convergence test per iteration.  Three collectives per iteration:

    precond_fwd     -> all-reduce(replicated entries of z after the local forward sweep  +  r.r)
    precond_rest    -> all-reduce(r.z)                                      [one scalar]
    direction_apply -> all-reduce(replicated entries of A p  +  p.A p)      [p^T A p = sum over ranks of p^T A_local p]
    update
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

    def __init__(self, sub, plan, info, device=0, element_wise_material=False, elementwise_pressure=False, nquad=None):
        import torch
        from .backend import ShellContext
        self.torch = torch
        # nquad: the rule of the WHOLE mesh (DistributedShell passes it) -- a partition made of affine cells only must not
        # integrate with another rule than its neighbours
        self.ctx = ShellContext(sub, element_wise_material, elementwise_pressure, device=device, nghost=info["nghost"], nquad=nquad)
        self.ctx.enable_frontal(plan=plan)
        self.ctx.set_solver(preconditioner=2, rtol=1e-10, maxit=50, check_every=1)
        self.device = torch.device("cuda", device)
        self.nvec = self.ctx.ndof
        self._views = {}
        # torch's collectives and the few tensor ops of the driver run on the context's own stream (``on_stream``): ordered
        # with the library's launches by the stream itself, no device-wide synchronisation in between.  The stream is made
        # current only inside ``on_stream`` -- a second engine or a caller that uses torch's default stream is not affected.
        self.stream = self.ctx.torch_stream()
        self.topbuf = self.scal = None

    def on_stream(self):
        return self.torch.cuda.stream(self.stream)

    def vec(self, name):
        if name not in self._views:
            self._views[name] = self.ctx.vec_tensor(name)
        return self._views[name]

    def new_tensor(self, n):
        return self.torch.zeros(n, dtype=self.torch.float64, device=self.device)

    def dist_setup(self, top_local, nranks, n_local_levels, sel):
        self.ctx.dist_setup(top_local, nranks, n_local_levels, sel)
        self.topbuf, self.scal = self.ctx.dist_tensors()

    def set_field(self, name, values):
        self.ctx.set_field(name, values)

    def set_quadrature(self, nquad):
        self.ctx.set_quadrature(nquad)

    def set_penalty_facets(self, pairs, beta):
        self.ctx.set_penalty_facets(pairs, beta)

    def load(self, dst):
        self.ctx.load_vec(dst)

    def pack(self, vec):
        self.ctx.dist_pack(vec)

    def unpack(self, vec):
        self.ctx.dist_unpack(vec)

    def factor(self, l0, l1, assemble):
        self.ctx.factorize_range(l0, l1, assemble)

    def schur_pack(self, front, out):
        self.ctx.front_schur_pack(front, out)

    def block_unpack(self, front, src):
        self.ctx.front_block_unpack(front, src)

    def pcg_start(self, b, x):
        self.ctx.dist_pcg_start(b, x)

    def precond_fwd(self):
        self.ctx.dist_precond_fwd()

    def read(self):
        return self.ctx.dist_read()

    def precond_rest(self):
        self.ctx.dist_precond_rest()

    def direction_apply(self, first):
        self.ctx.dist_direction_apply(first)

    def update(self, x):
        self.ctx.dist_update(x)

    def functionals_partial(self):
        return self.ctx.functionals_partial()

    def dfunctional_vec(self, name, dst):
        self.ctx.dfunctional_vec(name, dst)

    def gradient(self, functional, arg, lam, gglob):
        self.ctx.dist_gradient(functional, arg, lam, gglob)


class DistributedShell:
    """Forward solve, scalar outputs and the adjoint gradient on an element partition."""

    def __init__(self, mesh, comm: Comm, bc_marker=None, beta=1.0e15, leaf_size=None, engine_factory=None,
                 element_wise_material=False, device=0, tree=None, nquad=None, strict=True):
        import torch
        self.torch = torch
        self.strict = bool(strict)       # a PCG that stops at maxit short of rtol raises (option "strict" of the single-GPU path)
        self.mesh, self.comm = mesh, comm
        self.ewm = bool(element_wise_material)
        d = int(np.log2(comm.size))
        leaf_size = mesh.recommended_leaf_size() if leaf_size is None else int(leaf_size)
        self.tree = analyse(mesh, leaf_size, min_depth=d) if tree is None else tree
        self.sub, self.plan, self.info = rank_plan(mesh, self.tree, comm.rank, comm.size)
        self._rule_auto = nquad is None
        self.nquad = mesh.recommended_nquad() if nquad is None else int(nquad)
        factory = engine_factory or (lambda sub, plan, info: HipEngine(sub, plan, info, device=device,
                                                                       element_wise_material=element_wise_material, nquad=self.nquad))
        self.eng = factory(self.sub, self.plan, self.info)
        if bc_marker is not None:
            self.eng.set_penalty_facets(self.sub.penalty_facets(bc_marker), beta)
        self.nl, self.nlev = self.info["n_local_levels"], self.plan.nlevels
        self.ntop = int(self.info["n_top"])
        sel = self.info["cells"] if self.ewm else self.info["vertices"]
        # resident in the engine: the replicated entries (same global order on every rank), the dot weights (1 on the entries
        # only this rank holds, 1 / size on the replicated ones -- size is a power of two, so the weight is exact and the
        # replicated entries add up to one copy in the all-reduce), the place of this rank's field entries in the global field
        self.eng.dist_setup(self.info["top_local"], comm.size, self.nl, sel)
        self.rtol, self.maxit = 1e-10, 50
        self.factored = False
        self.last = {}
        self._grad_buf = None

    # ------------------------------------------------------------------ inputs
    def set_fields(self, thickness=None, E=None, nu=None, density=None, F_solid=None):
        """Global arrays (caller numbering); each rank keeps the entries of its vertices / cells."""
        sel = self.info["cells"] if self.ewm else self.info["vertices"]
        for name, v in (("thickness", thickness), ("E", E), ("nu", nu), ("density", density)):
            if v is not None:
                v = np.asarray(v, dtype=np.float64).ravel()
                if name == "nu" and self._rule_auto and not self.mesh.is_quad:
                    # the rule of the WHOLE mesh, decided on the global field so that every rank takes the same one (ShellContext.set_field)
                    varies = (not self.ewm) and v.size > 1 and bool(np.any(v != v[0]))
                    nq = self.mesh.recommended_nquad(nodal_nu_varies=varies)
                    if nq != self.nquad:
                        self.eng.set_quadrature(nq)
                        self.nquad = nq
                self.eng.set_field(name, v if v.size == 1 else v[sel])
        if F_solid is not None:
            f = np.asarray(F_solid, dtype=np.float64).reshape(-1, 3)
            self.eng.set_field("F_solid", f[self.info["vertices"]])
        self.factored = False

    # ------------------------------------------------------------------ collectives on replicated entries
    def _sum_top(self, name):
        """Sum a vector's replicated entries over the ranks (after a local operator application)."""
        if self.comm.size == 1:
            return
        self.eng.pack(name)
        self.comm.allreduce_(self.eng.topbuf[: self.ntop])
        self.eng.unpack(name)

    # ------------------------------------------------------------------ preconditioner set-up
    def factorize(self):
        # everything below -- the scratch tensor, the Schur all-gather and its host staging under gloo -- must run on the
        # context's stream, where the pack / unpack kernels run; the scope is re-entrant, callers need not wrap the call
        with self.eng.on_stream():
            self._factorize()

    def _factorize(self):
        eng, info = self.eng, self.info
        eng.factor(0, self.nl, True)
        if self.comm.size > 1:
            # the Schur complements of the subtree roots travel as packed lower triangles (they are symmetric and only the
            # lower triangle of a front is ever read), padded to the largest one: ONE all_gather_into_tensor
            sizes = info["schur_sizes"]
            cap = max(n * (n + 1) // 2 for n in sizes)
            if getattr(self, "_schur_mine", None) is None:
                self._schur_mine = eng.new_tensor(max(cap, 1))
            eng.schur_pack(info["root_front"], self._schur_mine)
            gathered = self.comm.allgather(self._schur_mine)
            for q in range(self.comm.size):
                if q != self.comm.rank:
                    eng.block_unpack(info["stub_fronts"][q], gathered[q])
            self._schur_keep = gathered            # alive until the unpack kernels have run (the next factorisation at the latest)
        eng.factor(self.nl, self.nlev, False)
        self.factored = True

    # ------------------------------------------------------------------ PCG
    def _pcg(self, b, x):
        eng, comm, ntop = self.eng, self.comm, self.ntop
        with eng.on_stream():
            if not self.factored:
                self.factorize()
            eng.pcg_start(b, x)
            self.last["converged"] = True          # of THIS solve; the maxit branch below flips it
            k, bb, rr = 0, None, None
            while True:
                # z = r and the forward sweep over the rank's subtree; its contributions to the replicated entries and the
                # rank's share of r.r travel together
                eng.precond_fwd()
                if comm.size > 1:
                    comm.allreduce_(eng.topbuf)
                rr, pAp = eng.read()                      # the one host synchronisation of the iteration
                if bb is None:
                    bb = rr                               # r = b before the first iteration
                if not rr == rr:
                    raise RuntimeError("PCG broke down (NaN residual)")
                if k > 0 and not pAp > 0:
                    raise RuntimeError("PCG broke down: p.Ap <= 0 (preconditioner or operator not positive definite)")
                if bb == 0 or rr <= self.rtol ** 2 * bb:
                    break
                if k >= self.maxit:
                    # as the single-GPU path does under option "strict" (status 4): an unconverged state must not flow on into
                    # the bench line or the adjoint gradient
                    self.last["converged"] = False
                    if self.strict:
                        from .backend import FemoConvergenceError
                        raise FemoConvergenceError(f"partitioned PCG stopped at maxit = {self.maxit} with relative residual "
                                                   f"{(rr / bb) ** 0.5:.3e} > rtol = {self.rtol:.1e}")
                    break
                eng.precond_rest()                        # top of the tree, backward sweeps, share of r.z
                if comm.size > 1:
                    comm.allreduce_(eng.scal[1:2])
                eng.direction_apply(k == 0)               # p, A_local p; replicated entries of A p + p.A_local p
                if comm.size > 1:
                    comm.allreduce_(eng.topbuf)
                eng.update(x)
                k += 1
        return k, (rr / bb) ** 0.5 if bb > 0 else 0.0

    def solve_state(self):
        """Forward solve K w = F (cold start); the state stays on the devices."""
        with self.eng.on_stream():
            self.eng.load("b")
            self._sum_top("b")
        it, rel = self._pcg("b", "state")
        self.last["forward"] = (it, rel)
        return it, rel

    def functional(self, name):
        uu, reg, mass = self.eng.functionals_partial()
        with self.eng.on_stream():
            t = self.eng.new_tensor(2)
            t.copy_(self.torch.tensor([uu + reg, mass], dtype=self.torch.float64))
            self.comm.allreduce_(t)
            vals = t.tolist()
        if name == "compliance":
            return vals[0]
        if name == "mass":
            return vals[1]
        raise ValueError(name)

    def total_gradient(self, functional="compliance", arg="thickness"):
        """d functional / d arg on the caller's global numbering (summed over ranks)."""
        with self.eng.on_stream():
            self.eng.dfunctional_vec(functional, "b")
            self._sum_top("b")
        it, rel = self._pcg("b", "adjoint")
        self.last["adjoint"] = (it, rel)
        with self.eng.on_stream():
            # the global gradient lives on the device: the rank's quadrature sweep scatters into its own entries, one all-reduce
            if self._grad_buf is None:
                self._grad_buf = self.eng.new_tensor(self.mesh.nel if self.ewm else self.mesh.nn)
            g = self._grad_buf
            g.zero_()
            self.eng.gradient(functional, arg, "adjoint", g)
            self.comm.allreduce_(g)
            out = g.cpu().numpy()
        return out, it, rel

    def gather_state(self):
        """Global state vector on every rank (tests / post-processing)."""
        with self.eng.on_stream():
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
