"""ctypes binding of libfemo_symbolic.so (include/femo_symbolic.h): the analysis phase in host C++."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _build

_lib = None
_i32p = C.POINTER(C.c_int32)

SIGNATURES = {
    "femo_plan_build": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _i32p,
                                  C.POINTER(C.c_double), _i32p, C.c_int32, C.c_int32]),
    "femo_plan_build_ex": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _i32p,
                                     C.POINTER(C.c_double), C.POINTER(C.c_double), _i32p, C.c_int32, C.c_int32, C.c_int32, C.c_double]),
    "femo_plan_build_ex2": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _i32p,
                                      C.POINTER(C.c_double), C.POINTER(C.c_double), _i32p, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_int32]),
    "femo_plan_size": (C.c_int64, [C.c_void_p, C.c_char_p]),
    "femo_plan_itemsize": (C.c_int, [C.c_void_p, C.c_char_p]),
    "femo_plan_get": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]),
    "femo_plan_free": (None, [C.c_void_p]),
    "femo_plan_last_error": (C.c_char_p, []),
}

ARRAYS = ("lo", "hi", "left", "right", "parent", "depth", "height", "eorder", "epos", "owner", "piv_nodes", "piv_off",
          "bnd_nodes", "bnd_off", "npiv", "nf", "dof_off", "front_dofs", "up_map", "elem_front", "elem_map", "level_nodes",
          "level_off")


def load():
    global _lib
    if _lib is None:
        path = _build.build_symbolic() if _build.symbolic_needs_build() else _build.SYM_LIB
        lib = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def plan_arrays(mesh, leaf_size, min_depth=0, axis_rule=0, gap=0.0, node_order=0):
    """All arrays of include/femo_symbolic.h for ``mesh`` as a dict of numpy arrays.

    The library assumes that the P2 nodes below ``nV`` carry six DOFs (u and theta) and the others three.  For CG2CR1 the rotation
    lives on the EDGE nodes, so the mesh is handed over with its P2 nodes relabelled -- edge midpoints first (they are the "vertices"
    of that layout), vertices behind them -- and the node and DOF numbers of the result are mapped back."""
    lib = load()
    cr = getattr(mesh, "element", "") == "CG2CR1"
    nV_lib = mesh.nE if cr else mesh.nV
    cell_p2 = np.ascontiguousarray(mesh.cell_p2, dtype=np.int32)
    cell_dofs = np.ascontiguousarray(mesh.cell_dofs(), dtype=np.int32)
    if cr:
        nV, nE = mesh.nV, mesh.nE
        to_lib = np.concatenate([nE + np.arange(nV), np.arange(nE)]).astype(np.int32)       # actual P2 node -> the library's number
        from_lib = np.argsort(to_lib).astype(np.int64)
        cell_p2 = np.ascontiguousarray(to_lib[cell_p2])
        cell_dofs = cell_dofs.copy()
        nu = 3 * cell_p2.shape[1]
        cell_dofs[:, :nu] = (3 * np.repeat(cell_p2, 3, axis=1) + np.tile(np.arange(3, dtype=np.int32), cell_p2.shape[1])[None, :])
        cell_dofs = np.ascontiguousarray(cell_dofs, dtype=np.int32)          # the theta entries (ndof_u + 3 edge + c) are the same in both layouts
    xc = mesh.nodes[mesh.cells]
    cent = np.ascontiguousarray(xc.mean(axis=1), dtype=np.float64)
    cext = np.ascontiguousarray(xc.max(axis=1) - xc.min(axis=1), dtype=np.float64)
    h = C.c_void_p()
    dp = C.POINTER(C.c_double)
    rc = lib.femo_plan_build_ex2(C.byref(h), mesh.nel, mesh.nP2, nV_lib, cell_p2.shape[1], cell_dofs.shape[1],
                                 cell_p2.ctypes.data_as(_i32p), cent.ctypes.data_as(dp), cext.ctypes.data_as(dp),
                                 cell_dofs.ctypes.data_as(_i32p), int(leaf_size), int(min_depth), int(axis_rule), float(gap), int(node_order))
    if rc:
        msg = lib.femo_plan_last_error().decode()
        raise ValueError(msg) if rc == 2 else RuntimeError(msg)
    try:
        out = {}
        for name in ARRAYS:
            n, isz = lib.femo_plan_size(h, name.encode()), lib.femo_plan_itemsize(h, name.encode())
            a = np.empty(n, dtype=np.int32 if isz == 4 else np.int64)
            if lib.femo_plan_get(h, name.encode(), a.ctypes.data_as(C.c_void_p), a.nbytes):
                raise RuntimeError(lib.femo_plan_last_error().decode())
            out[name] = a
    finally:
        lib.femo_plan_free(h)
    out["elem_map"] = out["elem_map"].reshape(mesh.nel, -1)
    if cr:
        fd = out["front_dofs"].astype(np.int64)
        isu = fd < mesh.ndof_u
        fd[isu] = 3 * from_lib[fd[isu] // 3] + fd[isu] % 3
        out["front_dofs"] = fd.astype(out["front_dofs"].dtype)
        for k in ("piv_nodes", "bnd_nodes"):
            out[k] = from_lib[out[k]].astype(out[k].dtype)
        out["owner"] = out["owner"][to_lib]
    return out
