"""ctypes binding of libmtip_hip.so (include/mtip_hip.h).

There is deliberately NO CPU fallback: if the HIP library is missing or no MI355X is visible the
import of the engine fails loudly (the reference would silently drop to its numpy path,
``xframe/projects/fxs/reconstruct.py:96-107``; a silent fallback here would void every parity claim).
"""
import ctypes as C
import os

import numpy as np

# HIP maps streams onto 4 hardware queues by default (one is the null stream's): with up to 3 engines per GPU give every
# engine its own queue.  Only effective if set before the HIP runtime initialises; harmless otherwise.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(os.path.dirname(_HERE), 'csrc', 'libmtip_hip.so')

c_double_p = C.POINTER(C.c_double)
c_u8_p = C.POINTER(C.c_uint8)
c_void = C.c_void_p


class MtipCfg(C.Structure):
    _fields_ = [('n_radial', C.c_int32), ('l_max', C.c_int32), ('n_theta', C.c_int32), ('n_phi', C.c_int32),
                ('n_batch', C.c_int32), ('hankel_trapz', C.c_int32), ('fused', C.c_int32), ('reserved', C.c_int32)]


class MtipError(RuntimeError):
    pass


_SIGNATURES = {
    'mtip_device_count': (C.c_int, []),
    'mtip_create': (c_void, [C.POINTER(MtipCfg), C.c_int]),
    'mtip_destroy': (None, [c_void]),
    'mtip_last_error': (C.c_char_p, [c_void]),
    'mtip_get_cfg': (C.c_int, [c_void, C.POINTER(MtipCfg)]),
    'mtip_synchronize': (C.c_int, [c_void]),
    'mtip_set_angular_grid': (C.c_int, [c_void, c_void, c_void]),
    'mtip_set_radial_grid': (C.c_int, [c_void, c_void, c_void]),
    'mtip_set_hankel_weights': (C.c_int, [c_void, c_void, C.c_double, C.c_double]),
    'mtip_set_projection_matrix': (C.c_int, [c_void, C.c_int, c_void, C.c_int, c_void, C.c_int]),
    'mtip_set_number_of_particles': (C.c_int, [c_void, C.c_double]),
    'mtip_set_so_freedom': (C.c_int, [c_void, C.c_int]),
    'mtip_set_invariant_metrics': (C.c_int, [c_void, C.c_uint32, c_void, c_void, c_void, c_void, c_void, C.c_double, c_void, c_void, c_void]),
    'mtip_fetch_invariant_metrics': (C.c_int, [c_void, C.c_int64, C.c_int64, c_void, c_void, c_void]),
    'mtip_set_reciprocal_l2_metric': (C.c_int, [c_void, c_void, c_void]),
    'mtip_fetch_reciprocal_l2_metric': (C.c_int, [c_void, C.c_int64, C.c_int64, c_void]),
    'mtip_op_invariant_metrics': (C.c_int, [c_void, c_void, c_void, c_void, c_void]),
    'mtip_set_deg2_metric': (C.c_int, [c_void, C.c_int]),
    'mtip_set_main_error': (C.c_int, [c_void, C.c_int, C.c_int]),
    'mtip_set_real_constraints': (C.c_int, [c_void, C.c_uint32, C.c_double, C.c_double, C.c_double, C.c_uint32]),
    'mtip_set_initial_support': (C.c_int, [c_void, c_void]),
    'mtip_set_error_weights': (C.c_int, [c_void, c_void, c_void, C.c_int]),
    'mtip_set_density': (C.c_int, [c_void, C.c_int, c_void]),
    'mtip_init_state': (C.c_int, [c_void]),
    'mtip_get_density': (C.c_int, [c_void, C.c_int, C.c_int, c_void]),
    'mtip_get_reciprocal_density': (C.c_int, [c_void, C.c_int, C.c_int, c_void]),
    'mtip_get_support': (C.c_int, [c_void, C.c_int, C.c_int, c_void]),
    'mtip_set_support': (C.c_int, [c_void, C.c_int, c_void, C.c_int]),
    'mtip_get_unknowns': (C.c_int, [c_void, C.c_int, C.c_int, c_void]),
    'mtip_get_best_error': (C.c_int, [c_void, c_void, C.POINTER(C.c_int64)]),
    'mtip_select_best': (C.c_int, [c_void]),
    'mtip_select_best_where': (C.c_int, [c_void, c_void]),
    'mtip_run': (C.c_int, [c_void, C.c_int, C.c_int, C.c_int, c_void, c_void, c_void]),
    'mtip_run_async': (C.c_int, [c_void, C.c_int, C.c_int, C.c_int, c_void]),
    'mtip_set_ft_stab_mask': (C.c_int, [c_void, c_void]),
    'mtip_run_group_async': (C.c_int, [c_void, C.c_int, C.c_int, C.c_int, C.c_int, c_void]),
    'mtip_fetch_errors': (C.c_int, [c_void, C.c_int64, C.c_int64, c_void, c_void]),
    'mtip_fetch_main_errors': (C.c_int, [c_void, C.c_int64, C.c_int64, c_void]),
    'mtip_shrinkwrap': (C.c_int, [c_void, C.c_double, C.c_double, C.c_double, c_void]),
    'mtip_begin_sub_loop': (C.c_int, [c_void]),
    'mtip_refresh_reciprocal_density': (C.c_int, [c_void]),
    'mtip_last_deg2_invariant': (C.c_int, [c_void, C.c_int, c_void]),
    'mtip_op_sht_forward': (C.c_int, [c_void, c_void, c_void, C.c_int]),
    'mtip_op_sht_inverse': (C.c_int, [c_void, c_void, c_void]),
    'mtip_op_sht_inverse_forward': (C.c_int, [c_void, c_void, c_void, c_void, C.c_int]),
    'mtip_op_hankel': (C.c_int, [c_void, c_void, c_void, C.c_int]),
    'mtip_op_fourier_transform': (C.c_int, [c_void, c_void, c_void, C.c_int]),
    'mtip_op_project_coefficients': (C.c_int, [c_void, c_void, c_void]),
    'mtip_op_project_real_intensity': (C.c_int, [c_void, c_void, c_void]),
    'mtip_op_apply_unknowns': (C.c_int, [c_void, c_void, c_void, c_void]),
    'mtip_op_modulus_replacement': (C.c_int, [c_void, c_void, c_void, c_void]),
    'mtip_op_real_space_update': (C.c_int, [c_void, c_void, c_void, C.c_int, C.c_double, c_void, c_void]),
    'mtip_op_deg2_invariants': (C.c_int, [c_void, c_void, c_void]),
    'mtip_op_apply_matrix': (C.c_int, [c_void, c_void, c_void, c_void, C.c_int, C.c_int, C.c_int]),
    'mtip_op_grid_stats': (C.c_int, [c_void, c_void, C.c_int, c_void, c_void, c_void, c_void]),
    'mtip_op_grid_phase_ramp': (C.c_int, [c_void, c_void, C.c_int, c_void, C.c_double]),
    'mtip_op_grid_combine': (C.c_int, [c_void, C.c_int, c_void, c_void, C.c_int, c_void]),
    'mtip_op_prtf': (C.c_int, [c_void, c_void, c_void, c_void, c_void, c_void, c_void]),
    'mtip_set_so3_tables': (C.c_int, [c_void, C.c_int, c_void]),
    'mtip_op_so3_correlation': (C.c_int, [c_void, c_void, c_void, C.c_int, C.c_int, c_void]),
    'mtip_op_rotate_coefficients': (C.c_int, [c_void, c_void, c_void, c_void]),
    'mtip2d_create': (c_void, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'mtip2d_destroy': (None, [c_void]),
    'mtip2d_last_error': (C.c_char_p, [c_void]),
    'mtip2d_set_hankel_weights': (C.c_int, [c_void, c_void, c_void, c_void]),
    'mtip2d_set_projection': (C.c_int, [c_void, C.c_int, c_void, c_void, c_void, c_void, C.c_double]),
    'mtip2d_op_harmonic': (C.c_int, [c_void, c_void, c_void, C.c_int]),
    'mtip2d_op_real_harmonic_forward': (C.c_int, [c_void, c_void, c_void]),
    'mtip2d_op_real_harmonic_inverse': (C.c_int, [c_void, c_void, c_void]),
    'mtip2d_op_hankel': (C.c_int, [c_void, c_void, c_void, C.c_int]),
    'mtip2d_op_fourier_transform': (C.c_int, [c_void, c_void, c_void, C.c_int]),
    'mtip2d_op_project': (C.c_int, [c_void, c_void, c_void, c_void]),
    'mtip2d_set_real_constraints': (C.c_int, [c_void, C.c_uint32, C.c_double, C.c_double, C.c_double, C.c_uint32]),
    'mtip2d_set_error_weights': (C.c_int, [c_void, c_void]),
    'mtip2d_set_so_freedom': (C.c_int, [c_void, C.c_int]),
    'mtip2d_op_step': (C.c_int, [c_void, C.c_int, C.c_int, C.c_double, c_void, c_void, c_void, c_void, c_void, c_void]),
    'mtip2d_op_step_ex': (C.c_int, [c_void, C.c_int, C.c_int, C.c_double, c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_void]),
    'mtip2d_op_shrinkwrap': (C.c_int, [c_void, c_void, C.c_double, C.c_double, c_void]),
    'mtip_op_so3_find_rotation': (C.c_int, [c_void, c_void, c_void, C.c_int, C.c_int, c_void, c_void, c_void]),
    'mtip_op_rotate_coefficients_grid': (C.c_int, [c_void, c_void, c_void, c_void, c_void, c_void]),
    'mtip_op_hermitian_eig': (C.c_int, [c_void, C.c_int, C.c_int, c_void, c_void, c_void]),
    'mtip_op_symmetric_eig': (C.c_int, [c_void, C.c_int, C.c_int, c_void, c_void, c_void]),
    'mtip_profile': (C.c_int, [c_void, C.c_int]),
    'mtip_profile_get': (C.c_int, [c_void, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    'mtip_profile_reset': (C.c_int, [c_void]),
    'mtip_debug_jacobi_sweeps': (C.c_int, [c_void, c_void]),
    'mtip_debug_projection_slots': (C.c_int, [c_void]),
    'mtip_debug_chain_timing': (C.c_int, [c_void, c_void]),
    'mtip_debug_check_jacobi_schedule': (C.c_int, [c_void, C.c_int]),
    'mtip_debug_polar_timing': (C.c_int, [c_void, c_void]),
    'mtip_debug_spin': (C.c_int, [c_void, C.c_double]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)
_libs = {}


def load(path=None):
    """Open the HIP library and declare every prototype of include/mtip_hip.h."""
    path = os.path.abspath(path or DEFAULT_LIB)
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise MtipError(f'{path} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                        '(hipcc --offload-arch=gfx950). There is no CPU fallback.')
    # One HIP runtime per process: the torch wheel bundles its own libamdhip64.so.7 (+ libhsa-runtime64) and /opt/rocm holds
    # another with the same soname.  Whichever is loaded first serves both; with this library first, torch's later import mixes
    # its bundled HSA runtime with /opt/rocm's HIP and finds "No HIP GPUs", and device memory of one runtime is unknown to the
    # other.  The averaging hands torch device tensors to the operators, so torch goes first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError = ABI mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _libs[path] = lib
    return lib


def ptr(a):
    """void* of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags['C_CONTIGUOUS']
    return a.ctypes.data_as(c_void)


def as_c128(a):
    return np.ascontiguousarray(a, dtype=np.complex128)


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def as_u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)
