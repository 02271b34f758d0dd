"""``ShellContext``: one mesh resident on one MI355X, a thin object wrapper over the C ABI
(include/femo_hip.h).  All arithmetic happens in libfemo_hip.so; this class only converts
numpy arrays to pointers and status codes to exceptions."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import FemoHipError, dptr, iptr
from .mesh import ShellMesh

PENALTY_BETA = 1.0e15      # reference linear_shell_model.py:324


class FemoConvergenceError(FemoHipError):
    """A Krylov solve stopped at maxit short of rtol (status 4; option 'strict')."""


class FemoNotPositiveDefiniteError(FemoHipError):
    """The multifrontal Cholesky met a non-positive pivot (status 5; option 'allow_pivot_repair')."""


class ShellContext:
    VEC_IDS = {"state": 0, "adjoint": 1, "r": 2, "z": 3, "p": 4, "Ap": 5, "b": 6}

    def __init__(self, mesh: ShellMesh, element_wise_material=False, elementwise_pressure=False,
                 nquad=None, device=0, nghost=0):
        self.lib = _lib.load()
        self.mesh = mesh
        # The rule of the static forms: by default what the mesh asks for (ShellMesh.recommended_nquad) -- quadrilaterals: Gauss points
        # per direction, 4 on affine cells, where that is exact, 5 / 6 on warped ones, where the reference's near-exact integration is
        # only met to 1e-9 by more points; triangles: the DEGREE of the symmetric rule, 6 where that is exact, and 9 (UFL's estimate for
        # these forms) from the moment a nodal Poisson ratio that varies over the cells arrives (set_field) -- unless the caller named
        # the rule.
        self._rule_auto = nquad is None
        nquad = mesh.recommended_nquad() if nquad is None else int(nquad)
        self.nquad = nquad
        self.element_wise_material = bool(element_wise_material)
        self.elementwise_pressure = bool(elementwise_pressure)
        h = C.c_void_p()
        xyz = np.ascontiguousarray(mesh.nodes, dtype=np.float64)
        cells = np.ascontiguousarray(mesh.cells, dtype=np.int32)
        cp2 = np.ascontiguousarray(mesh.cell_p2, dtype=np.int32)
        if getattr(mesh, "element", "CG2CG1") == "CG2CR1":
            # rotation on the edge midpoints (Crouzeix-Raviart, linear_shell_model.py:68-73): the element is named explicitly
            rc = self.lib.femo_create_element(C.byref(h), int(device), mesh.nn, mesh.nel, mesh.nvc, mesh.nP2,
                                              dptr(xyz), iptr(cells), iptr(cp2),
                                              int(self.element_wise_material), int(self.elementwise_pressure), int(nquad),
                                              int(nghost), 1)
        else:
            rc = self.lib.femo_create_ghost(C.byref(h), int(device), mesh.nn, mesh.nel, mesh.nvc, mesh.nP2,
                                            dptr(xyz), iptr(cells), iptr(cp2),
                                            int(self.element_wise_material), int(self.elementwise_pressure), int(nquad),
                                            int(nghost))
        if rc:
            raise FemoHipError(f"femo_create failed ({rc}): {self.lib.femo_last_error(None).decode()}")
        self._h = h
        self.device = int(device)
        self.nghost = int(nghost)
        self.ndof = int(self.lib.femo_ndof(h))            # vector length: mesh DOFs + ghost entries
        assert self.ndof == mesh.ndof + self.nghost
        # schedule experiments without touching the caller: FEMO_OPTIONS="sweep_w=0,strip_cnt=256" sets those options on every context
        import os
        self.env_options = {}
        for kv in filter(None, (t.strip() for t in os.environ.get("FEMO_OPTIONS", "").split(","))):
            k, sep, v = kv.partition("=")
            try:
                if not sep:
                    raise ValueError
                self.env_options[k.strip()] = float(v)
            except ValueError:
                raise FemoHipError(f"FEMO_OPTIONS: '{kv}' is not of the form option=number (e.g. FEMO_OPTIONS=\"sweep_w=0,strip_cnt=256\")") from None
            self.set_option(k.strip(), self.env_options[k.strip()])
        if self.env_options:
            import sys
            print(f"femo_alpha_amd: FEMO_OPTIONS sets {self.env_options} on this context", file=sys.stderr)

    # ------------------------------------------------------------------ plumbing
    def _chk(self, rc):
        if rc:
            msg = self.lib.femo_last_error(self._h).decode()
            if rc == 4:
                raise FemoConvergenceError(msg)
            if rc == 5:
                raise FemoNotPositiveDefiniteError(msg)
            raise FemoHipError(msg)

    def set_option(self, key, value):
        """Schedule switches and failure policy (femo_set_option): 'strict', 'allow_pivot_repair', 'trailing',
        'left_min', 'left_max', 'super_panel', 'super_panel_cnt', 'super_panel_ahead', 'lookahead', 'lookahead_cnt', 'grid_chunk',
        'wide_np', 'wide_cnt', 'profile_verbose'."""
        self._chk(self.lib.femo_set_option(self._h, key.encode(), float(value)))

    def close(self):
        if getattr(self, "_h", None):
            self.lib.femo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _vec(a):
        return np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel())

    # ------------------------------------------------------------------ data in
    def field_size(self, name):
        n = int(self.lib.femo_field_size(self._h, name.encode()))
        if n < 0:
            raise FemoHipError(f"unknown field '{name}'")
        return n

    def set_field(self, name, values):
        v = self._vec(values)
        self._chk(self.lib.femo_set_field(self._h, name.encode(), dptr(v), v.size))
        if name == "nu" and self._rule_auto and not self.mesh.is_quad:
            # a nodal Poisson ratio that varies over a cell makes the integrand rational: the reference's answer is then that of the
            # degree-9 rule UFL selects; everywhere else degree 6 integrates the same polynomial exactly (ShellMesh.recommended_nquad)
            varies = (not self.element_wise_material) and v.size > 1 and bool(np.any(v != v[0]))
            self.set_quadrature(self.mesh.recommended_nquad(nodal_nu_varies=varies))

    def set_quadrature(self, nquad):
        """The rule of the static forms (femo_set_quadrature): Gauss points per direction on quadrilaterals, the degree of the symmetric
        rule (4, 6, 9, 12) on triangles."""
        if int(nquad) != self.nquad:
            self._chk(self.lib.femo_set_quadrature(self._h, int(nquad)))
            self.nquad = int(nquad)

    def quadrature(self):
        """(rule, points per cell) in use."""
        n, q = C.c_int32(), C.c_int32()
        self._chk(self.lib.femo_get_quadrature(self._h, C.byref(n), C.byref(q)))
        return int(n.value), int(q.value)

    def get_field(self, name):
        out = np.empty(self.field_size(name))
        self._chk(self.lib.femo_get_field(self._h, name.encode(), dptr(out), out.size))
        return out

    def set_penalty_facets(self, facets, beta=PENALTY_BETA):
        f = np.ascontiguousarray(np.asarray(facets, dtype=np.int32).reshape(-1, 2))
        self._chk(self.lib.femo_set_penalty_facets(self._h, f.shape[0], iptr(f), float(beta)))

    def set_strong_dofs(self, dofs):
        d = np.ascontiguousarray(np.asarray(dofs, dtype=np.int32).ravel())
        self._chk(self.lib.femo_set_strong_dofs(self._h, d.size, iptr(d)))

    def set_state(self, w):
        w = self._vec(w)
        if w.size != self.ndof:
            raise ValueError("state has the wrong length")
        self._chk(self.lib.femo_set_state(self._h, dptr(w)))

    def get_state(self):
        w = np.empty(self.ndof)
        self._chk(self.lib.femo_get_state(self._h, dptr(w)))
        return w

    # ------------------------------------------------------------------ operators
    def apply_K(self, x):
        x = self._vec(x); y = np.empty(self.ndof)
        self._chk(self.lib.femo_apply_K(self._h, dptr(x), dptr(y)))
        return y

    def residual(self, w=None):
        r = np.empty(self.ndof)
        wp = None if w is None else dptr(self._vec(w))
        self._chk(self.lib.femo_residual(self._h, wp, dptr(r)))
        return r

    def load_vector(self):
        F = np.empty(self.ndof)
        self._chk(self.lib.femo_load_vector(self._h, dptr(F)))
        return F

    def diagonal(self):
        d = np.empty(self.ndof)
        self._chk(self.lib.femo_diagonal(self._h, dptr(d)))
        return d

    def element_matrices(self, first=0, count=None):
        count = self.mesh.nel - first if count is None else count
        ld = self.mesh.ldof
        Ke = np.empty((count, ld, ld))
        self._chk(self.lib.femo_element_matrices(self._h, first, count, dptr(Ke)))
        return Ke

    # ------------------------------------------------------------------ multifrontal preconditioner
    def enable_frontal(self, leaf_size=None, plan=None, **options):
        """Run the symbolic analysis on the host (mesh only) and upload it; afterwards
        ``set_solver(preconditioner=2)`` selects the multifrontal Cholesky preconditioner.
        ``plan`` may carry a ready-made plan (the multi-GPU driver passes rank-local plans);
        ``options`` are plan-shaping switches set before the upload (``wide_np``, ``wide_cnt``)."""
        import time
        from .solver.symbolic import build_plan
        leaf_size = self.mesh.recommended_leaf_size() if leaf_size is None else int(leaf_size)
        for k, v in options.items():
            self.set_option(k, v)
        t0 = time.perf_counter()
        plan = self.plan = build_plan(self.mesh, leaf_size) if plan is None else plan
        self.symbolic_s = time.perf_counter() - t0
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        i64 = lambda a: np.ascontiguousarray(a, dtype=np.int64)
        level_off = i32(np.concatenate([[0], np.cumsum([len(l) for l in plan.level_nodes])]))
        level_nodes = i32(np.concatenate(plan.level_nodes))
        arrs = dict(nf=i32(plan.nf), npiv=i32(plan.npiv), front_off=i64(plan.front_off), dof_off=i64(plan.dof_off),
                    front_dofs=i32(plan.front_dofs), up_map=i32(plan.up_map), parent=i32(plan.parent), left=i32(plan.left),
                    right=i32(plan.right), level_off=level_off, level_nodes=level_nodes,
                    elem_front=i32(plan.elem_front), elem_map=i32(plan.elem_map))
        p64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
        self._chk(self.lib.femo_set_frontal_plan(
            self._h, plan.ntree, plan.nlevels, iptr(arrs["nf"]), iptr(arrs["npiv"]), p64(arrs["front_off"]),
            p64(arrs["dof_off"]), iptr(arrs["front_dofs"]), iptr(arrs["up_map"]), iptr(arrs["parent"]), iptr(arrs["left"]),
            iptr(arrs["right"]), iptr(arrs["level_off"]), iptr(arrs["level_nodes"]), iptr(arrs["elem_front"]),
            iptr(arrs["elem_map"])))
        return plan

    def factorize(self):
        self._chk(self.lib.femo_factorize(self._h))
        return self.frontal_info()

    def factorize_profile(self, run=True):
        """One factorisation timed per kernel class with HIP events on the context's stream.  ``run=False``: only read what
        the factorisations since the last assembly recorded while option "profile" was on (the partitioned driver)."""
        t = np.zeros(32)
        self._chk(self.lib.femo_factorize_profile(self._h, dptr(t)) if run else self.lib.femo_factorize_profile_get(self._h, dptr(t)))
        names = ["panel_rows", "panel_diag", "trailing", "extend_add", "front_assemble", "memset", "l11_inverse"]
        out = {n: dict(ms=t[i], launches=int(t[8 + i])) for i, n in enumerate(names)}
        out["trailing_flops"], out["panel_rows_flops"], out["panel_diag_flops"] = t[18], t[16], t[17]
        out["trailing_bytes"], out["panel_rows_bytes"], out["panel_diag_bytes"] = t[26], t[24], t[25]
        # the rank-k launches above / below the ridge of the chip (9.8 flop per compulsory byte): bounded by the matrix cores / by HBM
        out["trailing_mfma_bound"] = dict(ms=t[7], launches=int(t[15]), flops=t[23], bytes=t[31])
        out["trailing_hbm_bound"] = dict(ms=t[2] - t[7], launches=int(t[10]) - int(t[15]), flops=t[18] - t[23], bytes=t[26] - t[31])
        return out

    def sweep_profile(self, detail=False):
        """(nlevels, 2) array: ms of the forward / backward triangular sweep per tree level (``detail``: (nlevels, 4),
        the two launches of each sweep separately)."""
        t = np.zeros(4 * self.plan.nlevels)
        self._chk(self.lib.femo_sweep_profile(self._h, dptr(t), t.size))
        t = t.reshape(-1, 4)
        return t if detail else np.stack([t[:, 0] + t[:, 1], t[:, 2] + t[:, 3]], axis=1)

    def sweep_profile_multi(self, nrhs):
        """(nlevels, 2): ms of the forward / backward sweep per level with ``nrhs`` (2 or 4) interleaved vectors."""
        t = np.zeros(2 * self.plan.nlevels)
        self._chk(self.lib.femo_sweep_profile_multi(self._h, int(nrhs), dptr(t), t.size))
        return t.reshape(-1, 2)

    def frontal_info(self):
        t = np.zeros(6)
        self._chk(self.lib.femo_frontal_info(self._h, dptr(t)))
        return dict(assemble_ms=t[0], factor_ms=t[1], front_GB=t[2], factor_gflop=t[3], pivots_repaired=int(t[4]),
                    fronts=int(t[5]))

    # ------------------------------------------------------------------ solves
    def set_solver(self, preconditioner=0, rtol=1e-10, maxit=200000, check_every=50):
        self._chk(self.lib.femo_set_solver(self._h, preconditioner, rtol, maxit, check_every))

    def use_direct_solver(self, leaf_size=None, rtol=1e-12, maxit=40):
        """What the reference's LU stands for (fea/utils_dolfinx.py:466,514-531): symbolic analysis once, then every
        solve = multifrontal Cholesky of the current operator + a few refinement steps of PCG on the true residual."""
        if getattr(self, "plan", None) is None:
            self.enable_frontal(leaf_size)
        self.set_solver(preconditioner=2, rtol=rtol, maxit=maxit, check_every=1)

    def set_krylov(self, method="cg"):
        """'cg' (default) or 'bicgstab' for the state, adjoint and linear solves."""
        self._chk(self.lib.femo_set_krylov(self._h, {"cg": 0, "bicgstab": 1}[method]))

    def solve_state(self, zero_guess=True):
        it = C.c_int32(); rr = C.c_double()
        self._chk(self.lib.femo_solve_state(self._h, int(zero_guess), C.byref(it), C.byref(rr)))
        return it.value, rr.value

    def solve_linear(self, rhs):
        rhs = self._vec(rhs); x = np.empty(self.ndof)
        it = C.c_int32(); rr = C.c_double()
        self._chk(self.lib.femo_solve_linear(self._h, dptr(rhs), dptr(x), C.byref(it), C.byref(rr)))
        return x, it.value, rr.value

    def solve_linear_multi(self, rhs):
        """x_i = K^-1 rhs_i for the rows of ``rhs`` (nrhs, ndof): with the multifrontal preconditioner the right-hand sides share the
        triangular sweeps in groups of up to four (femo_solve_linear_multi).  Returns (x (nrhs, ndof), iterations, relative residuals)."""
        rhs = np.ascontiguousarray(np.asarray(rhs, dtype=np.float64).reshape(-1, self.ndof))
        nr = rhs.shape[0]
        x = np.empty_like(rhs)
        it = np.zeros(nr, dtype=np.int32); rr = np.zeros(nr)
        self._chk(self.lib.femo_solve_linear_multi(self._h, nr, dptr(rhs), dptr(x), iptr(it), dptr(rr)))
        return x, it, rr

    def force_to_pressure(self, force, rtol=1e-13, maxit=500):
        """pressure = A^-1 force, A the consistent mass matrix of [CG1]^3 (rm_shell_model.py:414-421): Jacobi-PCG on the device."""
        f = self._vec(force)
        if f.size != 3 * self.mesh.nn:
            raise ValueError("force vector has the wrong length")
        out = np.empty_like(f)
        it = C.c_int32(); rr = C.c_double()
        self._chk(self.lib.femo_force_to_pressure(self._h, dptr(f), dptr(out), float(rtol), int(maxit), C.byref(it), C.byref(rr)))
        self.last_force_to_pressure = (it.value, rr.value)
        return out

    # ------------------------------------------------------------------ outputs
    def functional(self, name):
        v = C.c_double()
        self._chk(self.lib.femo_functional(self._h, name.encode(), C.byref(v)))
        return v.value

    # ------------------------------------------------------------------ dynamic-shell building blocks
    def set_operator(self, aK=1.0, aM=0.0):
        self._chk(self.lib.femo_set_operator(self._h, float(aK), float(aM)))

    def set_strain_quadrature(self, nred):
        self._chk(self.lib.femo_set_strain_quadrature(self._h, int(nred)))

    def op_apply_vec2(self, src, dst, aK, aM, with_penalty=True):
        self._chk(self.lib.femo_op_apply_vec2(self._h, self.VEC_IDS[src], self.VEC_IDS[dst], float(aK), float(aM), int(with_penalty)))

    def solve_vec(self, b, x, zero_guess=True):
        it = C.c_int32(); rr = C.c_double()
        self._chk(self.lib.femo_solve_vec(self._h, self.VEC_IDS[b], self.VEC_IDS[x], int(zero_guess), C.byref(it), C.byref(rr)))
        return it.value, rr.value

    def vec_mask_zero(self, name):
        self._chk(self.lib.femo_vec_mask_zero(self._h, self.VEC_IDS[name]))

    def grad_reset(self):
        self._chk(self.lib.femo_grad_reset(self._h))

    def grad_add(self, kind, x, y, scale):
        self._chk(self.lib.femo_grad_add(self._h, {"K": 0, "M": 1}[kind], self.VEC_IDS[x], self.VEC_IDS[y], float(scale)))

    def grad_get(self):
        out = np.empty(self.field_size("thickness"))
        self._chk(self.lib.femo_grad_get(self._h, dptr(out), out.size))
        return out

    # ------------------------------------------------------------------ transient march in the library (femo_newmark_*)
    def newmark_setup(self, time_levels, dt):
        self._chk(self.lib.femo_newmark_setup(self._h, int(time_levels), float(dt)))
        self._nm_levels = int(time_levels)

    def newmark_set_forces(self, f_history):
        f = np.ascontiguousarray(np.asarray(f_history, dtype=np.float64).reshape(-1, self.field_size("F_solid")))
        self._chk(self.lib.femo_newmark_set_forces(self._h, dptr(f.ravel()), f.shape[0]))

    def newmark_set_constant_load(self, F=None):
        self._chk(self.lib.femo_newmark_set_constant_load(self._h, None if F is None else dptr(self._vec(F))))

    def newmark_march(self, nsteps, reassemble_every_step=False):
        it = np.zeros(nsteps, dtype=np.int32); rr = np.zeros(nsteps)
        self._chk(self.lib.femo_newmark_march(self._h, int(nsteps), int(reassemble_every_step), iptr(it), dptr(rr)))
        return list(zip(it.tolist(), rr.tolist()))

    def newmark_history(self, which=0):
        """(time_levels, ndof) host copy of the displacement (0) or adjoint (2) history."""
        out = np.empty((self._nm_levels, self.ndof))
        self._chk(self.lib.femo_newmark_get_history(self._h, int(which), dptr(out.ravel())))
        return out

    def newmark_set_history(self, H, which=0):
        H = np.ascontiguousarray(np.asarray(H, dtype=np.float64).reshape(self._nm_levels, self.ndof))
        self._chk(self.lib.femo_newmark_set_history(self._h, int(which), dptr(H.ravel())))

    def newmark_adjoint(self, G):
        G = np.ascontiguousarray(np.asarray(G, dtype=np.float64).reshape(-1, self.ndof))
        self._chk(self.lib.femo_newmark_adjoint(self._h, dptr(G.ravel()), G.shape[0]))

    def newmark_residual_T(self, levels):
        g = np.empty(self.field_size("thickness")); dF = np.empty((levels, self.field_size("F_solid")))
        self._chk(self.lib.femo_newmark_residual_T(self._h, int(levels), dptr(g), dptr(dF.ravel())))
        return g, dF

    def newmark_jvp(self, levels, dY=None, dthickness=None, dF=None):
        """[J dY + (dR/dt) dthickness + (dR/df) dF] as a (levels, ndof) array (forward mode of the whole-history residual)."""
        a = None if dY is None else np.ascontiguousarray(np.asarray(dY, dtype=np.float64).reshape(levels, self.ndof))
        b = None if dthickness is None else self._vec(dthickness)
        cf = None if dF is None else np.ascontiguousarray(np.asarray(dF, dtype=np.float64).reshape(levels, self.field_size("F_solid")))
        self._chk(self.lib.femo_newmark_jvp(self._h, int(levels), None if a is None else dptr(a.ravel()), None if b is None else dptr(b),
                                            None if cf is None else dptr(cf.ravel())))
        return self.newmark_history(2)[:levels]

    def newmark_tangent(self, dR):
        """dY = J^-1 dR (the tangent linear march), (levels, ndof)."""
        dR = np.ascontiguousarray(np.asarray(dR, dtype=np.float64).reshape(-1, self.ndof))
        self._chk(self.lib.femo_newmark_tangent(self._h, dptr(dR.ravel()), dR.shape[0]))
        return self.newmark_history(2)[: dR.shape[0]]

    def newmark_tensor(self, which=0):
        """Zero-copy torch view of the resident displacement history (0: (levels, ndof)), velocity (1: (ndof,)) or adjoint history (2)."""
        ptr = self.lib.femo_newmark_ptr(self._h, int(which))
        n = self.ndof if which == 1 else self._nm_levels * self.ndof
        t = self._dev_tensor(ptr, n)
        return t if which == 1 else t.view(self._nm_levels, self.ndof)

    # ------------------------------------------------------------------ CSR export
    def enable_csr(self, host_map=False):
        """Pattern + destination-sorted contribution map, once per mesh: built on the device (radix sort of the nel * ld^2
        (row, column) keys); ``host_map=True`` builds it with numpy instead (femo_alpha_amd/csr.py, the cross-check)."""
        if host_map:
            from .csr import build_csr_map
            self.csr = build_csr_map(self.mesh)
            self._chk(self.lib.femo_set_csr_map(self._h, self.csr["nnz"], self.csr["perm"].size, iptr(self.csr["perm"]),
                                                iptr(self.csr["dest"])))
            return self.csr
        nnz = C.c_int32()
        self._chk(self.lib.femo_build_csr_map(self._h, C.byref(nnz)))
        rowptr, colidx = np.empty(self.mesh.ndof + 1, dtype=np.int32), np.empty(nnz.value, dtype=np.int32)
        self._chk(self.lib.femo_get_csr_pattern(self._h, iptr(rowptr), iptr(colidx)))
        self.csr = dict(nnz=int(nnz.value), rowptr=rowptr, colidx=colidx)
        return self.csr

    def assemble_csr(self):
        """scipy CSR matrix of the elastic stiffness for the current fields (no Dirichlet treatment)."""
        import scipy.sparse as sp
        vals = np.empty(self.csr["nnz"]); ms = np.zeros(2)
        self._chk(self.lib.femo_assemble_csr(self._h, dptr(vals), dptr(ms)))
        self.csr_timing = dict(element_matrices_ms=ms[0], scatter_ms=ms[1])
        n = self.mesh.ndof
        return sp.csr_matrix((vals, self.csr["colidx"], self.csr["rowptr"]), shape=(n, n))

    def set_stress_params(self, m=1e-6, rho=100.0):
        self._stress_params = None             # whatever a Form cached as "the parameters the context holds" is void now
        self._chk(self.lib.femo_set_stress_params(self._h, float(m), float(rho)))

    def set_stress_alpha(self, alpha=None, sel=-1):
        """The aggregate's normalisation given by the caller (None: the reference area, evaluated at first use)."""
        self._chk(self.lib.femo_set_stress_alpha(self._h, int(sel), -1.0 if alpha is None else float(alpha)))

    def set_cell_tags(self, tags, ntags):
        """Sub-domain index of every cell (-1: none) for the per-tag stress aggregates."""
        t = np.ascontiguousarray(tags, dtype=np.int32)
        self._chk(self.lib.femo_set_cell_tags(self._h, iptr(t), t.size, int(ntags)))

    def select_subdomain(self, sel=-1):
        self._chk(self.lib.femo_select_subdomain(self._h, int(sel)))

    def field_output(self, name="stress"):
        """DG1 projection of the von Mises stress: 'stress' (top surface), 'stress_mid', 'stress_bot'."""
        out = np.empty(self.mesh.nvc * self.mesh.nel)
        self._chk(self.lib.femo_field_output(self._h, name.encode(), dptr(out), out.size))
        return out

    def arg_size(self, wrt):
        return self.ndof if wrt == "disp_solid" else self.field_size(wrt)

    def dfunctional(self, name, wrt):
        out = np.empty(self.arg_size(wrt))
        self._chk(self.lib.femo_dfunctional(self._h, name.encode(), wrt.encode(), dptr(out), out.size))
        return out

    def dRdarg_T(self, arg, lam):
        lam = self._vec(lam); out = np.empty(self.field_size(arg))
        self._chk(self.lib.femo_dRdarg_T(self._h, arg.encode(), dptr(lam), dptr(out), out.size))
        return out

    def total_gradient(self, functional, arg):
        out = np.empty(self.field_size(arg))
        it = C.c_int32(); rr = C.c_double()
        self._chk(self.lib.femo_total_gradient(self._h, functional.encode(), arg.encode(), dptr(out), out.size,
                                               C.byref(it), C.byref(rr)))
        return out, it.value, rr.value

    def total_gradients(self, functionals, arg, subdomains=None):
        """d J_i / d arg for several functionals of the state with ONE grouped adjoint solve (femo_total_gradients); ``subdomains``:
        per functional the tagged sub-domain it is restricted to (-1: the whole mesh).  Returns (gradients (nfun, n), iterations,
        relative residuals)."""
        names = [f.encode() for f in functionals]
        nf = len(names)
        arr = (C.c_char_p * nf)(*names)
        out = np.empty((nf, self.field_size(arg)))
        it = np.zeros(nf, dtype=np.int32); rr = np.zeros(nf)
        sub = None if subdomains is None else np.ascontiguousarray(np.asarray(subdomains, dtype=np.int32))
        self._chk(self.lib.femo_total_gradients(self._h, nf, arr, None if sub is None else iptr(sub), arg.encode(), dptr(out), out.shape[1],
                                                iptr(it), dptr(rr)))
        return out, it, rr

    # ------------------------------------------------------------------ building blocks of the multi-GPU driver
    def vec_tensor(self, name):
        """Zero-copy torch view (float64, cuda) of one of the context's state-sized device vectors."""
        import torch

        class _Arr:      # __cuda_array_interface__ carrier
            pass
        a = _Arr()
        ptr = self.lib.femo_vec_ptr(self._h, self.VEC_IDS[name])
        a.__cuda_array_interface__ = dict(shape=(self.ndof,), typestr="<f8", data=(int(ptr), False), version=2)
        t = torch.as_tensor(a, device=f"cuda:{self.device}")
        t._femo_owner = self          # keep the context alive while the view exists
        return t

    def sync(self):
        self._chk(self.lib.femo_sync(self._h))

    def torch_stream(self):
        """The context's HIP stream as a ``torch.cuda.ExternalStream``: with it as torch's current stream, tensor ops on the
        zero-copy views and the collectives issued from Python are ordered with the library's kernels by the stream itself."""
        import torch
        return torch.cuda.ExternalStream(int(self.lib.femo_stream_ptr(self._h)), device=torch.device("cuda", self.device))

    def op_apply_vec(self, src, dst):
        self._chk(self.lib.femo_op_apply_vec(self._h, self.VEC_IDS[src], self.VEC_IDS[dst]))

    def load_vec(self, dst):
        self._chk(self.lib.femo_load_vec(self._h, self.VEC_IDS[dst]))

    def factorize_range(self, l0, l1, assemble):
        self._chk(self.lib.femo_factorize_range(self._h, l0, l1, int(assemble)))

    def frontal_sweep(self, vec, l0, l1, backward):
        self._chk(self.lib.femo_frontal_sweep(self._h, self.VEC_IDS[vec], l0, l1, int(backward)))

    def front_schur_get(self, front, dst_tensor):
        self._chk(self.lib.femo_front_schur_get(self._h, int(front), C.c_void_p(dst_tensor.data_ptr()), dst_tensor.numel()))

    def front_block_set(self, front, src_tensor):
        self._chk(self.lib.femo_front_block_set(self._h, int(front), C.c_void_p(src_tensor.data_ptr())))

    def functionals_partial(self):
        t = np.zeros(3)
        self._chk(self.lib.femo_functionals_partial(self._h, dptr(t)))
        return t

    def dfunctional_vec(self, name, dst):
        self._chk(self.lib.femo_dfunctional_vec(self._h, name.encode(), self.VEC_IDS[dst]))

    def field_gradient_vec(self, functional, arg, lam):
        out = np.empty(self.field_size(arg))
        self._chk(self.lib.femo_field_gradient_vec(self._h, functional.encode(), arg.encode(), self.VEC_IDS[lam],
                                                   dptr(out), out.size))
        return out

    # ---- the partitioned PCG (femo_dist_*): every call enqueues on the context's stream; collectives are the caller's
    def dist_setup(self, top_idx, nranks, n_local_levels, sel):
        ti = np.ascontiguousarray(top_idx, dtype=np.int32)
        se = np.ascontiguousarray(sel, dtype=np.int32)
        self._chk(self.lib.femo_dist_setup(self._h, ti.size, iptr(ti), int(nranks), int(n_local_levels), se.size, iptr(se)))
        self._ntop = ti.size

    def _dev_tensor(self, ptr, n):
        import torch

        class _Arr:
            pass
        a = _Arr()
        a.__cuda_array_interface__ = dict(shape=(n,), typestr="<f8", data=(int(ptr), False), version=2)
        t = torch.as_tensor(a, device=f"cuda:{self.device}")
        t._femo_owner = self
        return t

    def dist_tensors(self):
        """Zero-copy torch views of the two buffers the collectives run on: (ntop + 1 packed entries, 8 device scalars)."""
        return (self._dev_tensor(self.lib.femo_dist_ptr(self._h, 0), self._ntop + 1),
                self._dev_tensor(self.lib.femo_dist_ptr(self._h, 1), 8))

    def dist_pack(self, vec):
        self._chk(self.lib.femo_dist_pack(self._h, self.VEC_IDS[vec]))

    def dist_unpack(self, vec):
        self._chk(self.lib.femo_dist_unpack(self._h, self.VEC_IDS[vec]))

    def dist_pcg_start(self, b, x):
        self._chk(self.lib.femo_dist_pcg_start(self._h, self.VEC_IDS[b], self.VEC_IDS[x]))

    def dist_precond_fwd(self):
        self._chk(self.lib.femo_dist_precond_fwd(self._h))

    def dist_read(self):
        t = np.zeros(2)
        self._chk(self.lib.femo_dist_read(self._h, dptr(t)))
        return float(t[0]), float(t[1])

    def dist_precond_rest(self):
        self._chk(self.lib.femo_dist_precond_rest(self._h))

    def dist_direction_apply(self, first):
        self._chk(self.lib.femo_dist_direction_apply(self._h, int(first)))

    def dist_update(self, x):
        self._chk(self.lib.femo_dist_update(self._h, self.VEC_IDS[x]))

    def dist_gradient(self, functional, arg, lam, gglob_tensor):
        self._chk(self.lib.femo_dist_gradient(self._h, functional.encode(), arg.encode(), self.VEC_IDS[lam],
                                              C.c_void_p(gglob_tensor.data_ptr()), gglob_tensor.numel()))

    def front_schur_pack(self, front, dst_tensor):
        self._chk(self.lib.femo_front_schur_pack(self._h, int(front), C.c_void_p(dst_tensor.data_ptr()), dst_tensor.numel()))

    def front_block_unpack(self, front, src_tensor):
        self._chk(self.lib.femo_front_block_unpack(self._h, int(front), C.c_void_p(src_tensor.data_ptr())))

    def last_timing(self):
        t = np.zeros(5)
        self._chk(self.lib.femo_last_timing(self._h, dptr(t)))
        return dict(setup_ms=t[0], krylov_ms=t[1], total_ms=t[2], factor_state=int(t[3]), operator_launches=int(t[4]))

    def bench_kernel(self, name, reps=50):
        v = C.c_double()
        self._chk(self.lib.femo_bench_kernel(self._h, name.encode(), reps, C.byref(v)))
        return v.value
