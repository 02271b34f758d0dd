"""CPU ORACLE -- baseline infrastructure only, never the product path.

The "reference CPU path" column of the benchmark: the reference itself
(FEniCSx + PETSc/MUMPS) cannot run here, so what is timed on the host cores is this repository's float64 restatement
of its algorithm, kind "port":

  assembly          oracle/cpu_kernels.cpp, C++/OpenMP: element matrices (the arithmetic of FFCx's tabulate_tensor for
                    derivative(residual, w)) scattered into a CSR pattern (dolfinx assemble_matrix, fea/utils_dolfinx.py:200-206)
  direct solve      (a) scipy SuperLU on that CSR matrix -- a serial sparse LU as the nearest installed stand-in for
                        PETSc 'preonly' + 'lu' + MUMPS (fea/utils_dolfinx.py:466,514-531);
                    (b) a multifrontal Cholesky on dense fronts with LAPACK/BLAS (what MUMPS is algorithmically), all
                        host cores -- the "best CPU effort" a maintainer could reach without leaving the CPU.
  adjoint set-up    the sparse dR/dh, dR/dE, dR/dnu at the state (C++), dR/df, dR/du and A again (state_operation.py:283-296)

Two protocols are composed from the measured phases:
  "as the reference runs it"   3 Newton iterations = 3 assemblies + 3 factorisations + 3 solves + 4 residuals
                               (fea/utils_dolfinx.py:438-468), adjoint set-up = 7 matrix assemblies + 1 factorisation
  "best effort"                1 assembly + 1 factorisation + 1 solve, the factor reused by the adjoint
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import subprocess
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "_build")
SRC = os.path.join(HERE, "cpu_kernels.cpp")
LIB = os.path.join(BUILD, "libcpu_oracle.so")
FLAGS = ["-O3", "-march=x86-64-v3", "-fopenmp", "-shared", "-fPIC", "-std=c++17"]

_lib = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)


def build(force=False):
    """g++ the C++ restatement into oracle/_build/ (content-hashed like the HIP library)."""
    os.makedirs(BUILD, exist_ok=True)
    # every source the translation unit includes (the element core is included three times; a change there must rebuild as well)
    srcs = [SRC] + [os.path.join(HERE, f) for f in ("cpu_core.inc", "cpu_ext.inc", "cpu_dd.h")]
    digest = hashlib.sha256(b"".join(open(f, "rb").read() for f in srcs) + " ".join(FLAGS).encode()).hexdigest()
    stamp = LIB + ".srchash"
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == digest:
        return LIB
    tmp = f"{LIB}.{os.getpid()}.tmp"             # several ranks may build at once: link privately, publish atomically
    res = subprocess.run(["g++", *FLAGS, "-o", tmp, SRC], capture_output=True, text=True)
    if res.returncode:
        raise RuntimeError("g++ failed on oracle/cpu_kernels.cpp:\n" + res.stderr[-3000:])
    os.replace(tmp, LIB)
    with open(f"{stamp}.{os.getpid()}.tmp", "w") as fh:
        fh.write(digest + "\n")
    os.replace(f"{stamp}.{os.getpid()}.tmp", stamp)
    return LIB


def load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.cpu_max_threads.restype = C.c_int
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def _l(a):
    return a.ctypes.data_as(_lp)


def _blas_ptr(module, name):
    """Raw function pointer of a BLAS/LAPACK routine from scipy's Cython capsules (Fortran calling convention)."""
    cap = module.__pyx_capi__[name]
    C.pythonapi.PyCapsule_GetName.restype = C.c_char_p
    C.pythonapi.PyCapsule_GetName.argtypes = [C.py_object]
    C.pythonapi.PyCapsule_GetPointer.restype = C.c_void_p
    C.pythonapi.PyCapsule_GetPointer.argtypes = [C.py_object, C.c_char_p]
    return C.c_void_p(C.pythonapi.PyCapsule_GetPointer(cap, C.pythonapi.PyCapsule_GetName(cap)))


class CpuShell:
    """Host-side data of one workload for the C++ kernels (shares tables and fields with a ShellOracle)."""

    def __init__(self, oracle):
        o = self.o = oracle
        m = self.mesh = o.mesh
        self.lib = load()
        c = np.ascontiguousarray
        self.nodes = c(m.nodes, dtype=np.float64)
        self.cells = c(m.cells, dtype=np.int32)
        self.cell_p2 = c(m.cell_p2, dtype=np.int32)
        self.N1, self.dN1, self.dN2 = c(o.N1), c(o.dN1), c(o.dN2)
        self.w, self.wS = c(o.wts, dtype=np.float64), c(o.wts_strain, dtype=np.float64)
        self.hK = c(o.hK, dtype=np.float64)
        self.uhat = c(o.uhat) if np.any(o.uhat) else None
        self.N2 = c(o.N2)
        self.ld = o.ldof
        self._pattern = None

    def _common(self):
        o, m = self.o, self.mesh
        return (m.nel, o.nvc, o.npc, o.nq, _d(self.nodes), _i(self.cells))

    def _fields(self):
        o = self.o
        self._h, self._E, self._nu = (np.ascontiguousarray(a, dtype=np.float64) for a in (o.h, o.E, o.nu))
        return (_d(self._h), _d(self._E), _d(self._nu), int(o.ewm), _d(self.hK), int(self.mesh.is_quad))

    def _tables(self):
        return (_d(self.N1), _d(self.dN1), _d(self.dN2), _d(self.w), _d(self.wS))

    def _u(self):
        return _d(self.uhat) if self.uhat is not None else None

    def pattern(self):
        """CSR sparsity of K from the connectivity (what dolfinx's create_matrix builds once per form)."""
        if self._pattern is None:
            dofs = self.o.dofs
            n = self.mesh.ndof
            rows = np.repeat(dofs, self.ld, axis=1).ravel()
            cols = np.tile(dofs, (1, self.ld)).ravel()
            P = sp.csr_matrix((np.ones(rows.size, dtype=np.int8), (rows, cols)), shape=(n, n))
            P.sum_duplicates()
            P.sort_indices()
            self._pattern = (P.indptr.astype(np.int32), P.indices.astype(np.int32))
        return self._pattern

    def element_matrices(self, deriv=0, nthreads=1):
        Ke = np.empty((self.mesh.nel, self.ld, self.ld))
        rc = self.lib.cpu_element_matrices(*self._common(), self._u(), *self._tables(), *self._fields(), int(deriv), _d(Ke), int(nthreads))
        assert rc == 0
        return Ke

    def assemble_K(self, nthreads=1, with_bc=True):
        """Elastic stiffness in CSR through the C++ assembly; penalty blocks / strong rows added like the numpy oracle does."""
        rowptr, colidx = self.pattern()
        vals = np.zeros(colidx.size)
        m, o = self.mesh, self.o
        rc = self.lib.cpu_assemble_csr(*self._common(), _i(self.cell_p2), m.ndof_u, self._u(), *self._tables(), *self._fields(),
                                       _i(rowptr), _i(colidx), _d(vals), int(nthreads))
        assert rc == 0, "CSR pattern does not hold an element entry"
        K = sp.csr_matrix((vals, colidx, rowptr), shape=(m.ndof, m.ndof))
        if with_bc:
            if o.penalty_facets.shape[0]:
                r, c_, v = [], [], []
                for d, blk in o._penalty_blocks():
                    r.append(np.repeat(d, d.size)); c_.append(np.tile(d, d.size)); v.append(blk.ravel())
                K = K + sp.csr_matrix((np.concatenate(v), (np.concatenate(r), np.concatenate(c_))), shape=K.shape)
            if o.strong_dofs.size:
                K = o._apply_strong(K)
        return K

    def assemble_K_extended(self, nthreads=1):
        """(rowptr, colidx, values) of K = elastic stiffness + penalty blocks with the values in numpy.longdouble: element mathematics
        and sums in x87 extended precision (cpu_assemble_csr_ld).  The operator of the goldens -- a float64-assembled matrix carries
        rounding of its own that moves the 1 M-DOF solution by ~1e-7.  Penalty clamp only (what the golden workloads use)."""
        if np.finfo(np.longdouble).nmant < 63:
            raise RuntimeError("numpy longdouble is not the x87 80-bit type on this machine")
        m, o = self.mesh, self.o
        if o.strong_dofs.size:
            raise NotImplementedError("extended-precision assembly handles the penalty clamp")
        base = self.pattern()
        # the penalty blocks couple DOFs of one cell edge: inside the pattern already
        rowptr, colidx = base
        vals = np.zeros(colidx.size, dtype=np.longdouble)
        rc = self.lib.cpu_assemble_csr_ld(*self._common(), _i(self.cell_p2), m.ndof_u, self._u(), *self._tables(), *self._fields(),
                                          _i(rowptr), _i(colidx), vals.ctypes.data_as(C.c_void_p), int(nthreads))
        assert rc == 0, "CSR pattern does not hold an element entry"
        for d, blk in o._penalty_blocks():
            for a in range(d.size):
                row = colidx[rowptr[d[a]]:rowptr[d[a] + 1]]
                pos = rowptr[d[a]] + np.searchsorted(row, d)
                assert np.array_equal(colidx[pos], d)
                vals[pos] += blk[a].astype(np.longdouble)
        return rowptr, colidx, vals

    def assemble_K_dd(self, nthreads=1):
        """(rowptr, colidx, values (nnz, 2)) of the same operator in DOUBLE-DOUBLE arithmetic, values as (hi, lo) pairs
        (cpu_assemble_csr_dd; cpu_dd.h): the portable twin of assemble_K_extended -- no x87 needed, ~104 bits instead of 64."""
        m, o = self.mesh, self.o
        if o.strong_dofs.size:
            raise NotImplementedError("extended-precision assembly handles the penalty clamp")
        rowptr, colidx = self.pattern()
        vals = np.zeros((colidx.size, 2))
        rc = self.lib.cpu_assemble_csr_dd(*self._common(), _i(self.cell_p2), m.ndof_u, self._u(), *self._tables(), *self._fields(),
                                          _i(rowptr), _i(colidx), vals.ctypes.data_as(C.c_void_p), int(nthreads))
        assert rc == 0, "CSR pattern does not hold an element entry"
        for d, blk in o._penalty_blocks():                    # (float64 blocks, as in the x87 operator: added exactly)
            for a in range(d.size):
                row = colidx[rowptr[d[a]]:rowptr[d[a] + 1]]
                pos = rowptr[d[a]] + np.searchsorted(row, d)
                assert np.array_equal(colidx[pos], d)
                hi = vals[pos, 0] + blk[a]                    # two_sum: nothing of the 1e15-sized block is lost beside the elastic entry
                v = hi - vals[pos, 0]
                lo = (vals[pos, 0] - (hi - v)) + (blk[a] - v) + vals[pos, 1]
                vals[pos, 0], vals[pos, 1] = hi + lo, lo - ((hi + lo) - hi)
        return rowptr, colidx, vals

    def load_vector_dd(self):
        """F as (ndof, 2) (hi, lo) pairs (cpu_load_vector_dd)."""
        m, o = self.mesh, self.o
        F = np.zeros((m.ndof, 2))
        f = np.ascontiguousarray(o.f, dtype=np.float64)
        N2 = np.ascontiguousarray(o.N2)
        rc = self.lib.cpu_load_vector_dd(*self._common(), _i(self.cell_p2), self._u(), _d(self.N1), _d(self.dN1), _d(self.dN2), _d(N2),
                                         _d(self.w), _d(f), int(o.ewp), int(m.is_quad), F.ctypes.data_as(C.c_void_p))
        assert rc == 0
        return F

    def load_vector_extended(self):
        """F in numpy.longdouble (cpu_load_vector_ld)."""
        m, o = self.mesh, self.o
        F = np.zeros(m.ndof, dtype=np.longdouble)
        f = np.ascontiguousarray(o.f, dtype=np.float64)
        N2 = np.ascontiguousarray(o.N2)
        rc = self.lib.cpu_load_vector_ld(*self._common(), _i(self.cell_p2), self._u(), _d(self.N1), _d(self.dN1), _d(self.dN2), _d(N2),
                                         _d(self.w), _d(f), int(o.ewp), int(m.is_quad), F.ctypes.data_as(C.c_void_p))
        assert rc == 0
        return F

    def load_vector(self, nthreads=1):
        """F = int f . v J dx through the C++ vector assembly (strong rows zeroed like the numpy oracle)."""
        m, o = self.mesh, self.o
        F = np.zeros(m.ndof)
        f = np.ascontiguousarray(o.f, dtype=np.float64)
        N2 = np.ascontiguousarray(o.N2)
        rc = self.lib.cpu_load_vector(*self._common(), _i(self.cell_p2), self._u(), _d(self.N1), _d(self.dN1), _d(self.dN2), _d(N2),
                                      _d(self.w), _d(f), int(o.ewp), int(m.is_quad), _d(F), int(nthreads))
        assert rc == 0
        if o.strong_dofs.size:
            F[o.strong_dofs] = 0.0
        return F

    def apply_K(self, x, nthreads=1):
        """y = (K_elastic + penalty) x, matrix-free (C++/OpenMP element sweep; the penalty blocks as one small sparse matrix)."""
        m, o = self.mesh, self.o
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros(m.ndof)
        rc = self.lib.cpu_apply_K(*self._common(), _i(self.cell_p2), m.ndof_u, self._u(), *self._tables(), *self._fields(),
                                  _d(x), _d(y), int(nthreads))
        assert rc == 0
        if o.penalty_facets.shape[0]:
            if getattr(self, "_P", None) is None:
                r, c_, v = [], [], []
                for d, blk in o._penalty_blocks():
                    r.append(np.repeat(d, d.size)); c_.append(np.tile(d, d.size)); v.append(blk.ravel())
                self._P = sp.csr_matrix((np.concatenate(v), (np.concatenate(r), np.concatenate(c_))), shape=(m.ndof, m.ndof))
            y += self._P @ x
        return y

    def apply_op(self, x, aK, aM, nthreads=1, out=None):
        """out += (aK K_elastic + aM M) x, matrix-free (the step operator of the transient path and its right-hand side
        products; M = rho h (u.v + h_K^2 theta.eta) J dx, oracle assemble_M)."""
        m, o = self.mesh, self.o
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros(m.ndof) if out is None else out
        rho = np.ascontiguousarray(o.rho, dtype=np.float64)
        rc = self.lib.cpu_apply_op(*self._common(), _i(self.cell_p2), m.ndof_u, self._u(), *self._tables(), *self._fields(),
                                   _d(self.N2), _d(rho), C.c_double(aK), C.c_double(aM), _d(x), _d(y), int(nthreads))
        assert rc == 0
        return y

    def drdfield_T(self, name, state, lam, nthreads=1, scale=1.0, out=None):
        """out += scale * (dR/d field)^T lam at ``state`` by quadrature -- no matrix is assembled."""
        m, o = self.mesh, self.o
        out = np.zeros(o.h.size) if out is None else out
        st, lm = (np.ascontiguousarray(a, dtype=np.float64) for a in (state, lam))
        rc = self.lib.cpu_drdfield_T(*self._common(), _i(self.cell_p2), m.ndof_u, self._u(), *self._tables(), *self._fields(),
                                     {"h": 1, "E": 2, "nu": 3}[name], _d(st), _d(lm), C.c_double(scale), _d(out), int(nthreads))
        assert rc == 0
        return out

    def dcompliance(self, state, nthreads=1):
        """(d compliance / d u, d compliance / d h) in one sweep: 2 int N2 u J dx and the regularisation's thickness gradient."""
        from oracle.rm_shell_oracle import REG_ALPHA1
        m, o = self.mesh, self.o
        st = np.ascontiguousarray(state, dtype=np.float64)
        du, dh = np.zeros(m.ndof), np.zeros(o.h.size)
        N2 = np.ascontiguousarray(o.N2)
        hh = np.ascontiguousarray(o.h, dtype=np.float64)
        rc = self.lib.cpu_dcompliance(*self._common(), _i(self.cell_p2), self._u(), _d(self.N1), _d(self.dN1), _d(self.dN2), _d(N2),
                                      _d(self.w), _d(hh), int(o.ewm), C.c_double(REG_ALPHA1), int(m.is_quad), _d(st), _d(du), _d(dh),
                                      int(nthreads))
        assert rc == 0
        return du, dh

    def assemble_drdfield(self, name, state, nthreads=1):
        """Sparse ndof x n_field matrix dR/d(field) at ``state`` (field in h, E, nu)."""
        m, o = self.mesh, self.o
        nfe = 1 if o.ewm else o.nvc
        out = np.empty((m.nel, self.ld, nfe))
        st = np.ascontiguousarray(state, dtype=np.float64)
        rc = self.lib.cpu_assemble_drdfield(*self._common(), _i(self.cell_p2), m.ndof_u, self._u(), *self._tables(), *self._fields(),
                                            {"h": 1, "E": 2, "nu": 3}[name], _d(st), _d(out), int(nthreads))
        assert rc == 0
        cols = (np.arange(m.nel)[:, None] if o.ewm else m.cells)
        rows = np.repeat(o.dofs, nfe, axis=1).ravel()
        cc = np.tile(cols, (1, self.ld)).ravel()
        return sp.csr_matrix((out.ravel(), (rows, cc)), shape=(m.ndof, m.nel if o.ewm else m.nn))


class CpuMultifrontal:
    """Numeric multifrontal Cholesky on the CPU over a FrontalPlan (nested dissection of the elements): leaf fronts
    from the C++ element matrices, then level by level extend-add + dpotrf / dtrsm / dsyrk per front.  Levels with many
    fronts run one front per OpenMP thread with single-threaded BLAS; the few large fronts at the top run one after the
    other with multi-threaded BLAS."""

    def __init__(self, shell: CpuShell, plan, nthreads):
        import scipy.linalg.cython_blas as cb
        import scipy.linalg.cython_lapack as cl
        self.s, self.p, self.nthreads = shell, plan, int(nthreads)
        self.ptr = dict(potrf=_blas_ptr(cl, "dpotrf"), trsm=_blas_ptr(cb, "dtrsm"), syrk=_blas_ptr(cb, "dsyrk"),
                        trsv=_blas_ptr(cb, "dtrsv"), gemv=_blas_ptr(cb, "dgemv"))
        c = np.ascontiguousarray
        p = plan
        self.nf, self.npiv = c(p.nf, dtype=np.int32), c(p.npiv, dtype=np.int32)
        self.front_off, self.dof_off = c(p.front_off, dtype=np.int64), c(p.dof_off, dtype=np.int64)
        self.left, self.right = c(p.left, dtype=np.int32), c(p.right, dtype=np.int32)
        self.up_map, self.front_dofs = c(p.up_map, dtype=np.int32), c(p.front_dofs, dtype=np.int32)
        self.elem_map = c(p.elem_map, dtype=np.int32)
        order = np.argsort(p.elem_front, kind="stable").astype(np.int32)
        self.elem_order = order
        self.elem_start = np.searchsorted(p.elem_front[order], np.arange(p.ntree + 1)).astype(np.int32)
        self.levels = [c(l, dtype=np.int32) for l in p.level_nodes]
        self.order = c(np.concatenate(self.levels), dtype=np.int32)
        self.level_off = c(np.concatenate([[0], np.cumsum([l.size for l in self.levels])]), dtype=np.int32)
        self.F = np.empty(int(p.front_off[-1]))
        self.leaves = c(np.nonzero(p.left < 0)[0], dtype=np.int32)
        self.operator = (1.0, 0.0)          # (aK, aM): the matrix factorised is aK K + aM M (+ Dirichlet treatment)

    def factorize(self):
        """Returns (assembly seconds, factorisation seconds)."""
        from threadpoolctl import threadpool_limits
        s, lib = self.s, self.s.lib
        t0 = time.perf_counter()
        if self.operator == (1.0, 0.0):
            rc = lib.cpu_fronts_assemble(self.leaves.size, _i(self.leaves), _i(self.nf), _l(self.front_off), _i(self.elem_start),
                                         _i(self.elem_order), _i(self.elem_map), s.o.nvc, s.o.npc, s.o.nq, _d(s.nodes), _i(s.cells),
                                         s._u(), *s._tables(), *s._fields(), _d(self.F), self.nthreads)
        else:
            rho = np.ascontiguousarray(s.o.rho, dtype=np.float64)
            rc = lib.cpu_fronts_assemble_op(self.leaves.size, _i(self.leaves), _i(self.nf), _l(self.front_off), _i(self.elem_start),
                                            _i(self.elem_order), _i(self.elem_map), s.o.nvc, s.o.npc, s.o.nq, _d(s.nodes), _i(s.cells),
                                            s._u(), *s._tables(), *s._fields(), _d(s.N2), _d(rho), C.c_double(self.operator[0]),
                                            C.c_double(self.operator[1]), _d(self.F), self.nthreads)
        assert rc == 0
        self._dirichlet()
        t1 = time.perf_counter()
        for lev in self.levels:
            # one front per OpenMP thread (BLAS single-threaded) from half as many fronts as threads on; the few fronts above run one after
            # the other with threaded BLAS.  Measured on the GPU box's host at 16 threads (profiles/r5_cpu_levels.txt): the level of 16 fronts
            # 26 against 126 ms, of 8 fronts 55 against 84, of 4 fronts 82 against 56 -- round 4's rule (2 x threads) sent the levels of 16
            # and 8 fronts through the slow branch: factorisation 0.53 -> 0.39 s
            many = lev.size >= max(2, self.nthreads // 2) or self.nthreads == 1
            with threadpool_limits(limits=1 if many else self.nthreads):
                rc = lib.cpu_fronts_factor_level(lev.size, _i(lev), _i(self.nf), _i(self.npiv), _l(self.front_off), _l(self.dof_off),
                                                 _i(self.left), _i(self.right), _i(self.up_map), _d(self.F), self.ptr["potrf"],
                                                 self.ptr["trsm"], self.ptr["syrk"], self.nthreads if many else 1)
            if rc:
                raise np.linalg.LinAlgError("front not positive definite")
        return t1 - t0, time.perf_counter() - t1

    def _dirichlet(self):
        """Penalty blocks into the leaf front of the facet's cell; strong rows/columns replaced by identity."""
        o, p = self.s.o, self.p
        if o.penalty_facets.shape[0]:
            for k, (d, blk) in enumerate(o._penalty_blocks()):          # six blocks per facet (3 components x {u, theta})
                cell = int(o.penalty_facets[k // 6, 0])
                t = int(p.elem_front[cell])
                fd = p.front_dofs[p.dof_off[t]:p.dof_off[t + 1]]
                srt = np.argsort(fd)
                pos = srt[np.searchsorted(fd[srt], d)]
                n = int(p.nf[t])
                Ft = self.F[p.front_off[t]:p.front_off[t + 1]].reshape(n, n, order="F")
                ii, jj = np.meshgrid(pos, pos, indexing="ij")
                lower = ii >= jj
                np.add.at(Ft, (ii[lower], jj[lower]), blk[lower])
        if o.strong_dofs.size:
            # strong rows / columns out of the leaf fronts, unit diagonal in their place (dolfinx assemble_matrix(bcs) semantics,
            # oracle _apply_strong).  A constrained DOF shared by k leaves ends up with the pivot k: its row of the system reads
            # k x = 0 either way, the right-hand side is zeroed there.
            if getattr(self, "_strong_leaves", None) is None:
                is_strong = np.zeros(o.mesh.ndof, dtype=bool)
                is_strong[o.strong_dofs] = True
                self._strong_leaves = []
                for t in self.leaves:
                    fd = p.front_dofs[p.dof_off[t]:p.dof_off[t + 1]]
                    pos = np.nonzero(is_strong[fd])[0]
                    if pos.size:
                        self._strong_leaves.append((int(t), pos))
            for t, pos in self._strong_leaves:
                n = int(p.nf[t])
                Ft = self.F[p.front_off[t]:p.front_off[t + 1]].reshape(n, n, order="F")
                Ft[pos, :] = 0.0
                Ft[:, pos] = 0.0
                Ft[pos, pos] = 1.0

    def solve(self, b, by_level=True):
        """Triangular sweeps with the factor.  ``by_level``: the fronts of a tree level in parallel (one per OpenMP thread,
        single-threaded BLAS) -- the round-2 sweep visited the fronts one after the other and did not scale with the cores."""
        from threadpoolctl import threadpool_limits
        x = np.array(b, dtype=np.float64)
        if by_level and self.nthreads > 1:
            with threadpool_limits(limits=1):
                self.s.lib.cpu_fronts_solve_levels(len(self.levels), _i(self.level_off), _i(self.order), _i(self.nf), _i(self.npiv),
                                                   _l(self.front_off), _l(self.dof_off), _i(self.front_dofs), _d(self.F), _d(x),
                                                   self.ptr["trsv"], self.ptr["gemv"], self.nthreads)
            return x
        with threadpool_limits(limits=self.nthreads):
            self.s.lib.cpu_fronts_solve(self.order.size, _i(self.order), _i(self.nf), _i(self.npiv), _l(self.front_off), _l(self.dof_off),
                                        _i(self.front_dofs), _d(self.F), _d(x), self.ptr["trsv"], self.ptr["gemv"])
        return x


def dynamic_march(cs, mf, f_history, dt, nsteps, cores, reassemble_every_step=False):
    """Midpoint / Newmark march of the reference's PlateSim on the host cores (oracle dynamic_history with the C++/OpenMP kernels
    and the multifrontal Cholesky in place of numpy + SuperLU): per step
        (2/dt^2 M + K/2) w_i = F_i + M (2/dt^2 w_{i-1} + 2/dt wdot_{i-1}) - K/2 w_{i-1},   wdot_i = 2/dt (w_i - w_{i-1}) - wdot_{i-1},
    strong Dirichlet rows, zero initial state, ONE direct solve per step (solveNonlinear_mod runs a single Newton iteration,
    nonlinear_utils.py:220-229).  ``reassemble_every_step``: the step operator is assembled and factorised before every solve, as
    the reference does (plate_sim.py:319 -> nonlinear_utils.py:210-233) and BASELINE config 5 words it; otherwise once.
    Returns (last state, seconds per phase: assemble, factor, rhs, solve)."""
    o = cs.o
    a, b = 2.0 / dt ** 2, 2.0 / dt
    mf.operator = (0.5, a)
    mf.nthreads = cores
    keep = np.ones(o.mesh.ndof); keep[o.strong_dofs] = 0.0
    w = np.zeros(o.mesh.ndof); wd = np.zeros(o.mesh.ndof)
    f0 = o.f.copy()
    t = dict(assemble=0.0, factor=0.0, rhs=0.0, solve=0.0)
    for i in range(1, nsteps + 1):
        if reassemble_every_step or i == 1:
            ta, tf = mf.factorize()
            t["assemble"] += ta; t["factor"] += tf
        t0 = time.perf_counter()
        o.f = np.asarray(f_history[min(i, len(f_history) - 1)], dtype=np.float64).reshape(-1, 3)
        rhs = cs.load_vector(cores)
        y = cs.apply_op(w, -0.5, a, cores)                 # (a M - K/2) w_{i-1}
        cs.apply_op(wd, 0.0, b, cores, out=y)              # + b M wdot_{i-1}
        rhs += keep * y
        rhs[o.strong_dofs] = 0.0
        t1 = time.perf_counter()
        wn = mf.solve(rhs)
        t2 = time.perf_counter()
        wd = b * (wn - w) - wd
        w = wn
        t["rhs"] += t1 - t0; t["solve"] += t2 - t1
    o.f = f0
    return w, t


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_quota_cores():
    """CPUs' worth of time the control group of this process may use per scheduling period (cgroup v2 ``cpu.max``, v1
    ``cpu.cfs_quota_us`` / ``cpu.cfs_period_us``), or None without a quota.  A one-GPU box of this pool lists 256 hardware threads in
    ``sched_getaffinity`` and grants 16: more runnable threads than that are throttled by the scheduler for the rest of every 100 ms
    period -- which is why round 4's "all cores" leg (64 threads) took 3.7 s for a factorisation that 16 threads do in 0.5 s
    (profiles/r5_cpu_levels.txt: stalls of 70-100 ms per parallel region, at random)."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        if quota != "max":
            return max(1, int(quota) // int(period))
        return None
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
            quota, period = int(fq.read()), int(fp.read())
        return max(1, quota // period) if quota > 0 else None
    except (OSError, ValueError):
        return None


def cpu_throttled_periods():
    """Scheduling periods in which this control group was throttled so far (cgroup v2 ``cpu.stat``), or None."""
    try:
        with open("/sys/fs/cgroup/cpu.stat") as fh:
            for line in fh:
                if line.startswith("nr_throttled"):
                    return int(line.split()[1])
    except (OSError, ValueError):
        pass
    return None


def host_cores(cap=None):
    """Threads the baseline may use: the cores this process can run on AND is granted time on (affinity mask, then the control
    group's CPU quota), capped at the CPU share a one-GPU box gives a job (16; ``FEMO_CPU_CORES`` overrides; ``cap`` = 64 asks for
    "every core").  scipy's OpenBLAS is built for at most 64 threads and crashes beyond."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = cpu_quota_cores()
    if quota is not None:
        n = min(n, quota)
    cap = int(os.environ.get("FEMO_CPU_CORES", 16)) if cap is None else cap
    return max(1, min(n, cap, 64))


def _median_time(fn, repeats, warm=1):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), ts


def measure(oracle, plan, cores, repeats=5, superlu=True, log=None):
    """Phase timings (seconds, medians of ``repeats`` after one warm-up) with ``cores`` threads.  Returns a dict with
    the phases and the two composed protocols of BASELINE.md section 3."""
    from threadpoolctl import threadpool_limits
    say = log or (lambda *a: None)
    cs = CpuShell(oracle)
    m = oracle.mesh
    cs.pattern()
    out = dict(cores=int(cores), repeats=int(repeats), ndof=int(m.ndof), cells=int(m.nel))
    K_holder = {}
    out["assemble_csr_s"], _ = _median_time(lambda: K_holder.__setitem__("K", cs.assemble_K(cores)), repeats)
    say(f"  [{cores} cores] CSR assembly {out['assemble_csr_s']:.2f} s")
    K = K_holder["K"]
    b = cs.load_vector(cores)
    out["load_vector_s"], _ = _median_time(lambda: cs.load_vector(cores), repeats, warm=0)
    t0 = time.perf_counter(); r = K @ b; out["residual_s"] = time.perf_counter() - t0 + out["load_vector_s"]
    # (b) multifrontal Cholesky, LAPACK/BLAS
    mf = CpuMultifrontal(cs, plan, cores)
    asm, fac = [], []
    for k in range(repeats + 1):
        a, f = mf.factorize()
        if k:
            asm.append(a); fac.append(f)
    out["mf_assemble_s"], out["mf_factor_s"] = float(np.median(asm)), float(np.median(fac))
    w = mf.solve(b)
    for _ in range(2):                                     # refinement on the true residual, as Newton iterations 2-3 do
        w += mf.solve(b - K @ w)
    out["mf_solve_s"], _ = _median_time(lambda: mf.solve(b), min(repeats, 3), warm=0)
    out["mf_relres"] = float(np.linalg.norm(b - K @ w) / np.linalg.norm(b))
    say(f"  [{cores} cores] multifrontal: assemble {out['mf_assemble_s']:.2f} s, factor {out['mf_factor_s']:.2f} s, "
        f"solve {out['mf_solve_s']:.2f} s, relres {out['mf_relres']:.1e}")
    # adjoint pieces at the state w
    t0 = time.perf_counter()
    for name in ("h", "E", "nu"):
        cs.assemble_drdfield(name, w, cores)
    out["assemble_drdfield_x3_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    rhs = oracle.dcompliance_du(w)
    lam = mf.solve(rhs)
    lam += mf.solve(rhs - K @ lam)
    g = oracle.dcompliance_dh(w) - cs.assemble_drdfield("h", w, cores).T @ lam
    out["adjoint_gradient_best_s"] = time.perf_counter() - t0
    # (a) SuperLU on the assembled matrix (serial)
    if superlu:
        try:
            with threadpool_limits(limits=cores):
                t0 = time.perf_counter()
                lu = spla.splu(K.tocsc(), permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
                out["superlu_factor_s"] = time.perf_counter() - t0
                t0 = time.perf_counter(); ws = lu.solve(b); out["superlu_solve_s"] = time.perf_counter() - t0
            out["superlu_vs_mf"] = float(np.abs(ws - w).max() / np.abs(w).max())
            say(f"  [{cores} cores] SuperLU factor {out['superlu_factor_s']:.1f} s, solve {out['superlu_solve_s']:.2f} s")
        except MemoryError:
            # scipy's SuperLU indexes its fill with 32-bit integers: "Not enough memory to perform factorization" at 1 M DOF
            superlu = False
            out["superlu"] = "failed: MemoryError (32-bit fill indices; the 1M-DOF factor does not fit)"
            say(f"  [{cores} cores] SuperLU cannot factorise this matrix (MemoryError): the multifrontal Cholesky stands in")
    # composed protocols
    nd = m.ndof
    best = out["mf_assemble_s"] + out["mf_factor_s"] + 3 * out["mf_solve_s"]
    out["best_effort"] = dict(forward_s=best, dof_per_s=nd / best, adjoint_gradient_s=out["adjoint_gradient_best_s"],
                              what="1 front assembly + 1 multifrontal Cholesky + 3 solves (refinement), factor reused by the adjoint")
    if not superlu:
        # the reference's protocol with the multifrontal Cholesky as the direct solver (an LU -- what PC 'lu' runs -- would do
        # twice its flops: this favours the CPU)
        fwd = 3 * (out["assemble_csr_s"] + out["mf_assemble_s"] + out["mf_factor_s"] + out["mf_solve_s"]) + 4 * out["residual_s"]
        adj = 3 * out["assemble_csr_s"] + out["assemble_drdfield_x3_s"] + out["mf_assemble_s"] + out["mf_factor_s"]
        out["as_reference"] = dict(forward_s=fwd, dof_per_s=nd / fwd, adjoint_setup_s=adj,
                                   what="3 x (CSR assembly + multifrontal Cholesky + solve) + 4 residuals (utils_dolfinx.py:438-468); "
                                        "adjoint set-up 7 matrix assemblies + 1 factorisation (state_operation.py:260-296)")
    if superlu:
        fwd = 3 * (out["assemble_csr_s"] + out["superlu_factor_s"] + out["superlu_solve_s"]) + 4 * out["residual_s"]
        # 7 matrices: dR/du, A (two K assemblies), dR/dh, dR/dE, dR/dnu (measured), dR/df and dR/duhat (counted as one K
        # assembly together: a mass-like matrix and the shape derivative of the residual form), + 1 factorisation
        adj = 3 * out["assemble_csr_s"] + out["assemble_drdfield_x3_s"] + out["superlu_factor_s"]
        out["as_reference"] = dict(forward_s=fwd, dof_per_s=nd / fwd, adjoint_setup_s=adj,
                                   what="3 x (CSR assembly + SuperLU factorisation + solve) + 4 residuals (utils_dolfinx.py:438-468); "
                                        "adjoint set-up 7 matrix assemblies + 1 factorisation (state_operation.py:260-296)")
        b1 = out["assemble_csr_s"] + out["superlu_factor_s"] + 2 * out["superlu_solve_s"]
        out["best_effort_superlu"] = dict(forward_s=b1, dof_per_s=nd / b1, what="1 CSR assembly + 1 SuperLU factorisation + 2 solves")
    return out
