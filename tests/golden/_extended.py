"""Shared by the golden generators: the discrete operator in extended precision and the refinement against it.

Why: the goldens are meant to be THE solution of the discrete problem (mesh, fields, quadrature tables as float64 data), to which
both the HIP path and the float64 CPU restatement are approximations.  A matrix assembled in float64 is not that: its entries
carry rounding of their own, and on these thin shells the solution is sensitive to it far beyond 1e-8 -- at BASELINE config 3
(1 M DOF, 1.27 mm skin) a change of the 5-point Gauss weights in the last place (<= 4e-16) moves displacement / compliance /
gradient by 3.5e-7 / 2.8e-7 / 3.5e-7, at config 2 (255 k DOF) the same change of the 4-point table moves them by 4e-9 / 8e-9 / 1e-8.
So the element matrices are formed and summed in x87 extended precision (oracle/cpu_kernels.cpp, cpu_assemble_csr_ld: 64-bit
mantissa, unit round-off 5e-20), the load vector likewise, and a float64 factorisation is only the preconditioner of an iterative
refinement whose residual F - K x is accumulated in extended precision on that operator."""
import numpy as np


def extended_system(cs, cores):
    """(rowptr, colidx, values as longdouble), load vector as longdouble -- of the CpuShell ``cs``."""
    rowptr, colidx, vals = cs.assemble_K_extended(cores)
    return (rowptr, colidx, vals), cs.load_vector_extended()


def refine(Kx, solve, b, x, steps=12, tol=1e-17, log=None):
    """x <- x + solve(b - K x) with the residual in extended precision; returns (x as float64, size of the last correction
    relative to the solution).  ``b`` may be float64 or longdouble."""
    if np.finfo(np.longdouble).nmant < 63:
        raise RuntimeError("numpy longdouble is not the x87 80-bit type on this machine")
    rowptr, colidx, data = Kx
    xl = np.asarray(x).astype(np.longdouble)
    bl = np.asarray(b).astype(np.longdouble)
    rel = np.inf
    for k in range(steps):
        r = bl - np.add.reduceat(data * xl[colidx], rowptr[:-1])
        dx = solve(np.asarray(r, dtype=np.float64))
        xl += dx
        rel = float(np.abs(dx).max() / np.abs(xl).max())
        if log:
            log(f"    refinement {k}: correction {rel:.1e}")
        if rel < tol:
            break
    return np.asarray(xl, dtype=np.float64), rel
