"""ctypes binding of libfemo_hip.so (include/femo_hip.h).  Fails loudly when the HIP
library is missing -- there is deliberately no CPU fallback behind this module."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from ._build import LIB

_c_double_p = C.POINTER(C.c_double)
_c_int32_p = C.POINTER(C.c_int32)

# name -> (restype, argtypes); kept in the same order as include/femo_hip.h
SIGNATURES = {
    "femo_version": (C.c_int, []),
    "femo_device_count": (C.c_int, []),
    "femo_last_error": (C.c_char_p, [C.c_void_p]),
    "femo_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                              _c_double_p, _c_int32_p, _c_int32_p, C.c_int, C.c_int, C.c_int]),
    "femo_create_ghost": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                    _c_double_p, _c_int32_p, _c_int32_p, C.c_int, C.c_int, C.c_int, C.c_int32]),
    "femo_create_element": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                      _c_double_p, _c_int32_p, _c_int32_p, C.c_int, C.c_int, C.c_int, C.c_int32, C.c_int]),
    "femo_destroy": (None, [C.c_void_p]),
    "femo_ndof": (C.c_int64, [C.c_void_p]),
    "femo_field_size": (C.c_int64, [C.c_void_p, C.c_char_p]),
    "femo_set_penalty_facets": (C.c_int, [C.c_void_p, C.c_int32, _c_int32_p, C.c_double]),
    "femo_set_strong_dofs": (C.c_int, [C.c_void_p, C.c_int32, _c_int32_p]),
    "femo_set_field": (C.c_int, [C.c_void_p, C.c_char_p, _c_double_p, C.c_int64]),
    "femo_get_field": (C.c_int, [C.c_void_p, C.c_char_p, _c_double_p, C.c_int64]),
    "femo_set_state": (C.c_int, [C.c_void_p, _c_double_p]),
    "femo_get_state": (C.c_int, [C.c_void_p, _c_double_p]),
    "femo_apply_K": (C.c_int, [C.c_void_p, _c_double_p, _c_double_p]),
    "femo_residual": (C.c_int, [C.c_void_p, _c_double_p, _c_double_p]),
    "femo_load_vector": (C.c_int, [C.c_void_p, _c_double_p]),
    "femo_diagonal": (C.c_int, [C.c_void_p, _c_double_p]),
    "femo_element_matrices": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _c_double_p]),
    "femo_set_frontal_plan": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _c_int32_p, _c_int32_p, C.POINTER(C.c_int64),
                                        C.POINTER(C.c_int64), _c_int32_p, _c_int32_p, _c_int32_p, _c_int32_p, _c_int32_p,
                                        _c_int32_p, _c_int32_p, _c_int32_p, _c_int32_p]),
    "femo_factorize": (C.c_int, [C.c_void_p]),
    "femo_factorize_profile": (C.c_int, [C.c_void_p, _c_double_p]),
    "femo_frontal_info": (C.c_int, [C.c_void_p, _c_double_p]),
    "femo_sweep_profile": (C.c_int, [C.c_void_p, _c_double_p, C.c_int64]),
    "femo_sweep_profile_multi": (C.c_int, [C.c_void_p, C.c_int32, _c_double_p, C.c_int64]),
    "femo_set_solver": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_int32, C.c_int32]),
    "femo_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_double]),
    "femo_set_krylov": (C.c_int, [C.c_void_p, C.c_int]),
    "femo_solve_state": (C.c_int, [C.c_void_p, C.c_int, _c_int32_p, _c_double_p]),
    "femo_solve_linear": (C.c_int, [C.c_void_p, _c_double_p, _c_double_p, _c_int32_p, _c_double_p]),
    "femo_solve_linear_multi": (C.c_int, [C.c_void_p, C.c_int32, _c_double_p, _c_double_p, _c_int32_p, _c_double_p]),
    "femo_total_gradients": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_char_p), _c_int32_p, C.c_char_p, _c_double_p, C.c_int64,
                                       _c_int32_p, _c_double_p]),
    "femo_force_to_pressure": (C.c_int, [C.c_void_p, _c_double_p, _c_double_p, C.c_double, C.c_int32, _c_int32_p, _c_double_p]),
    "femo_set_operator": (C.c_int, [C.c_void_p, C.c_double, C.c_double]),
    "femo_set_strain_quadrature": (C.c_int, [C.c_void_p, C.c_int32]),
    "femo_set_quadrature": (C.c_int, [C.c_void_p, C.c_int32]),
    "femo_get_quadrature": (C.c_int, [C.c_void_p, _c_int32_p, _c_int32_p]),
    "femo_quadrature_tables": (C.c_int, [C.c_int32] * 5 + [_c_int32_p] + [_c_double_p] * 8),
    "femo_op_apply_vec2": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_int]),
    "femo_solve_vec": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int, _c_int32_p, _c_double_p]),
    "femo_vec_mask_zero": (C.c_int, [C.c_void_p, C.c_int32]),
    "femo_grad_reset": (C.c_int, [C.c_void_p]),
    "femo_grad_add": (C.c_int, [C.c_void_p, C.c_int, C.c_int32, C.c_int32, C.c_double]),
    "femo_grad_get": (C.c_int, [C.c_void_p, _c_double_p, C.c_int64]),
    "femo_build_csr_map": (C.c_int, [C.c_void_p, _c_int32_p]),
    "femo_get_csr_pattern": (C.c_int, [C.c_void_p, _c_int32_p, _c_int32_p]),
    "femo_set_csr_map": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, _c_int32_p, _c_int32_p]),
    "femo_assemble_csr": (C.c_int, [C.c_void_p, _c_double_p, _c_double_p]),
    "femo_set_stress_params": (C.c_int, [C.c_void_p, C.c_double, C.c_double]),
    "femo_set_stress_alpha": (C.c_int, [C.c_void_p, C.c_int32, C.c_double]),
    "femo_set_cell_tags": (C.c_int, [C.c_void_p, _c_int32_p, C.c_int64, C.c_int32]),
    "femo_select_subdomain": (C.c_int, [C.c_void_p, C.c_int32]),
    "femo_field_output": (C.c_int, [C.c_void_p, C.c_char_p, _c_double_p, C.c_int64]),
    "femo_functional": (C.c_int, [C.c_void_p, C.c_char_p, _c_double_p]),
    "femo_dfunctional": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, _c_double_p, C.c_int64]),
    "femo_dRdarg_T": (C.c_int, [C.c_void_p, C.c_char_p, _c_double_p, _c_double_p, C.c_int64]),
    "femo_total_gradient": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, _c_double_p, C.c_int64, _c_int32_p,
                                      _c_double_p]),
    "femo_last_timing": (C.c_int, [C.c_void_p, _c_double_p]),
    "femo_bench_kernel": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32, _c_double_p]),
    "femo_vec_ptr": (C.c_void_p, [C.c_void_p, C.c_int32]),
    "femo_sync": (C.c_int, [C.c_void_p]),
    "femo_stream_ptr": (C.c_void_p, [C.c_void_p]),
    "femo_op_apply_vec": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "femo_load_vec": (C.c_int, [C.c_void_p, C.c_int32]),
    "femo_factorize_range": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int]),
    "femo_frontal_sweep": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int]),
    "femo_front_schur_get": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]),
    "femo_front_block_set": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "femo_functionals_partial": (C.c_int, [C.c_void_p, _c_double_p]),
    "femo_dfunctional_vec": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32]),
    "femo_field_gradient_vec": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int32, _c_double_p, C.c_int64]),
    "femo_dist_setup": (C.c_int, [C.c_void_p, C.c_int32, _c_int32_p, C.c_int32, C.c_int32, C.c_int32, _c_int32_p]),
    "femo_dist_ptr": (C.c_void_p, [C.c_void_p, C.c_int32]),
    "femo_dist_pack": (C.c_int, [C.c_void_p, C.c_int32]),
    "femo_dist_unpack": (C.c_int, [C.c_void_p, C.c_int32]),
    "femo_dist_pcg_start": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "femo_dist_precond_fwd": (C.c_int, [C.c_void_p]),
    "femo_dist_read": (C.c_int, [C.c_void_p, _c_double_p]),
    "femo_dist_precond_rest": (C.c_int, [C.c_void_p]),
    "femo_dist_direction_apply": (C.c_int, [C.c_void_p, C.c_int]),
    "femo_dist_update": (C.c_int, [C.c_void_p, C.c_int32]),
    "femo_dist_gradient": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int32, C.c_void_p, C.c_int64]),
    "femo_front_schur_pack": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]),
    "femo_front_block_unpack": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "femo_factorize_profile_get": (C.c_int, [C.c_void_p, _c_double_p]),
    "femo_newmark_setup": (C.c_int, [C.c_void_p, C.c_int32, C.c_double]),
    "femo_newmark_set_forces": (C.c_int, [C.c_void_p, _c_double_p, C.c_int32]),
    "femo_newmark_set_constant_load": (C.c_int, [C.c_void_p, _c_double_p]),
    "femo_newmark_march": (C.c_int, [C.c_void_p, C.c_int32, C.c_int, _c_int32_p, _c_double_p]),
    "femo_newmark_get_history": (C.c_int, [C.c_void_p, C.c_int32, _c_double_p]),
    "femo_newmark_set_history": (C.c_int, [C.c_void_p, C.c_int32, _c_double_p]),
    "femo_newmark_adjoint": (C.c_int, [C.c_void_p, _c_double_p, C.c_int32]),
    "femo_newmark_residual_T": (C.c_int, [C.c_void_p, C.c_int32, _c_double_p, _c_double_p]),
    "femo_newmark_jvp": (C.c_int, [C.c_void_p, C.c_int32, _c_double_p, _c_double_p, _c_double_p]),
    "femo_newmark_tangent": (C.c_int, [C.c_void_p, _c_double_p, C.c_int32]),
    "femo_newmark_ptr": (C.c_void_p, [C.c_void_p, C.c_int32]),
    "femo_device_ptr": (C.c_void_p, [C.c_void_p, C.c_char_p]),
}

_lib = None


class FemoHipError(RuntimeError):
    """A libfemo_hip call returned a non-zero status."""


def load():
    """Load libfemo_hip.so and bind every symbol of the header; raises if the library or a
    symbol is missing (never falls back to a CPU path)."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: the PyTorch wheel ships its own libamdhip64.  If this library initialised /opt/rocm's copy
    # first, a later ``import torch`` would bring a second runtime that finds no device ("No HIP GPUs are available").  With
    # torch imported first, libfemo_hip's dependency resolves to the copy already loaded.  torch is only plumbing here
    # (device tensors over the context's buffers, torch.distributed); the library itself does not need it.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    from . import _build
    if _build.needs_build():
        # missing, or compiled from other sources than the ones beside it (content digest): rebuild, or refuse
        try:
            _build.build(force=True)
        except RuntimeError as e:
            raise FemoHipError(
                f"{LIB} is missing or stale and cannot be rebuilt here ({e}); build it with "
                "`python -m femo_alpha_amd._build` (hipcc, gfx950). femo_alpha_amd has no CPU fallback.") from e
    lib = C.CDLL(LIB)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def dptr(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_c_double_p)


def iptr(a: np.ndarray):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_c_int32_p)
