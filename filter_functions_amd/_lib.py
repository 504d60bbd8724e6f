"""ctypes binding of ``libffk.so`` (C ABI declared in ``include/ffk.h``).

The product path has no CPU fallback: if the shared library is missing or a HIP
call fails, the error is raised -- never papered over.  Importing this module
only loads the library (no GPU is touched until the first call), so the
package imports, and the exported symbols can be checked, on a machine without
a GPU.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_size_t, c_uint, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('FFK_LIBRARY', os.path.join(_HERE, 'libffk.so'))

FFK_OK = 0
FFK_EINVAL = -1
FFK_EHIP = -2
FFK_ENOMEM = -3
FFK_ENOCONV = -4
FFK_EKERNEL = -5

WANT_NOISE_OPERATORS = 0x1
FF_FIDELITY = 0
FF_GENERALIZED = 1
MAX_D = 64             # include/ffk.h FFK_MAX_D: diagonalize, control matrix, filter function, Liouville
MAX_D_TEMPLATED = 16   # FFK_MAX_D_TEMPLATED: intermediates, second order, gradients, cumulant, resident passes

_dp = POINTER(c_double)
_ip = POINTER(c_int32)


class FFKError(RuntimeError):
    """A HIP runtime failure inside libffk."""


class FFKKernelFault(FFKError):
    """A kernel reported an internal fault (``FFK_EKERNEL``): a bounded wait between the wavefronts
    of the d = 4 accumulate kernel ran out and the launch's results are invalid."""


class ffk_stats(ctypes.Structure):
    _fields_ = [('accumulate_flops', c_double), ('accumulate_bytes', c_double),
                ('chunks', c_int), ('grid_x', c_int), ('grid_y', c_int), ('grid_z', c_int),
                ('block', c_int), ('lds_bytes', c_int)]


#: every symbol include/ffk.h declares: name -> (restype, argtypes)
SIGNATURES = {
    'ffk_last_error': (c_char_p, []),
    'ffk_version': (c_int, []),
    'ffk_device_count': (c_int, [POINTER(c_int)]),
    'ffk_set_device': (c_int, [c_int]),
    'ffk_get_device': (c_int, [POINTER(c_int)]),
    'ffk_device_info': (c_int, [ctypes.c_char_p, c_int, POINTER(c_int), POINTER(c_size_t)]),
    'ffk_malloc': (c_int, [POINTER(c_void_p), c_size_t]),
    'ffk_malloc_finegrained': (c_int, [POINTER(c_void_p), c_size_t]),
    'ffk_free': (c_int, [c_void_p]),
    'ffk_memset': (c_int, [c_void_p, c_int, c_size_t, c_void_p]),
    'ffk_memcpy_h2d': (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    'ffk_memcpy_d2h': (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    'ffk_memcpy_d2d': (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    'ffk_stream_create': (c_int, [POINTER(c_void_p)]),
    'ffk_stream_destroy': (c_int, [c_void_p]),
    'ffk_stream_synchronize': (c_int, [c_void_p]),
    'ffk_device_synchronize': (c_int, []),
    'ffk_event_create': (c_int, [POINTER(c_void_p)]),
    'ffk_event_destroy': (c_int, [c_void_p]),
    'ffk_event_record': (c_int, [c_void_p, c_void_p]),
    'ffk_event_synchronize': (c_int, [c_void_p]),
    'ffk_event_elapsed_ms': (c_int, [c_void_p, c_void_p, POINTER(c_float)]),
    'ffk_release_arena': (c_int, []),
    'ffk_diagonalize': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'ffk_diagonalize_workspace_bytes': (c_size_t, [c_int, c_int]),
    'ffk_diagonalize_dev': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_size_t, c_void_p]),
    'ffk_control_matrix': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                   c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                   c_uint, c_void_p, c_void_p]),
    'ffk_control_matrix_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    'ffk_control_matrix_dev': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                       c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                       c_int, c_uint, c_void_p, c_void_p, c_void_p, c_size_t,
                                       c_void_p]),
    'ffk_noise_operators_intermediates': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                                  c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                                  c_int, c_int, c_void_p, c_void_p, c_void_p,
                                                  c_void_p]),
    'ffk_control_matrix_intermediates': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                                 c_void_p, c_int, c_void_p, c_int, c_void_p,
                                                 c_void_p, c_void_p, c_int, c_int, c_void_p,
                                                 c_void_p, c_void_p, c_void_p, c_void_p,
                                                 c_void_p]),
    'ffk_control_matrix_from_atomic': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                               c_int, c_int, c_int, c_void_p]),
    'ffk_control_matrix_from_atomic_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int]),
    'ffk_control_matrix_from_atomic_dev': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                                   c_int, c_int, c_int, c_void_p, c_void_p,
                                                   c_size_t, c_void_p]),
    'ffk_control_matrix_from_atomic_indexed': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                                       c_int, c_int, c_int, c_int, c_int, c_int,
                                                       c_void_p]),
    'ffk_control_matrix_from_atomic_indexed_dev': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p,
                                                           c_int, c_int, c_int, c_int, c_int, c_int,
                                                           c_int, c_void_p, c_void_p, c_size_t,
                                                           c_void_p]),
    'ffk_concatenate_sequence': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                         c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                         c_void_p, c_void_p]),
    'ffk_concatenate_sequence_resident': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                                  c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                                  c_void_p]),
    'ffk_control_matrix_periodic': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                            c_int, c_void_p]),
    'ffk_control_matrix_periodic_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'ffk_control_matrix_periodic_dev': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                                c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    'ffk_filter_function': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'ffk_filter_function_dev': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'ffk_filter_function_weighted': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_double,
                                             c_void_p]),
    'ffk_filter_function_weighted_dev': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_double,
                                                 c_void_p, c_void_p]),
    'ffk_infidelity': (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int,
                               c_int, c_void_p]),
    'ffk_infidelity_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'ffk_infidelity_dev': (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                   c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    'ffk_infidelity_sharded_dev': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p,
                                           c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'ffk_noise_operators_from_atomic': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                                c_int, c_void_p]),
    'ffk_decay_amplitudes_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    'ffk_decay_amplitudes_dev': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int,
                                         c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t,
                                         c_void_p]),
    'ffk_decay_amplitudes_shard_dev': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                               c_int, c_void_p, c_int, c_int, c_void_p, c_int,
                                               c_void_p, c_void_p, c_size_t, c_void_p]),
    'ffk_decay_amplitudes': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int,
                                     c_void_p, c_void_p, c_int, c_void_p]),
    'ffk_cumulant_function_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'ffk_cumulant_function_dev': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p,
                                          c_void_p, c_size_t, c_void_p]),
    'ffk_cumulant_function': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    'ffk_second_order_filter_function': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                                 c_void_p, c_int, c_void_p, c_int, c_void_p,
                                                 c_void_p, c_void_p, c_int, c_int, c_void_p]),
    'ffk_second_order_filter_function_from_atomic': (c_int, [c_void_p, c_void_p, c_void_p, c_int,
                                                             c_int, c_int, c_int, c_void_p]),
    'ffk_frequency_shifts_from_scratch': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                                  c_void_p, c_int, c_void_p, c_int, c_void_p,
                                                  c_void_p, c_void_p, c_int, c_int, c_void_p,
                                                  c_int, c_void_p, c_int, c_void_p, c_void_p]),
    'ffk_frequency_shifts': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p,
                                     c_void_p, c_int, c_void_p]),
    'ffk_cumulant_function_second_order': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p,
                                                   c_void_p]),
    'ffk_second_order_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    'ffk_second_order_filter_function_dev': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                                     c_void_p, c_int, c_void_p, c_int, c_void_p,
                                                     c_void_p, c_void_p, c_int, c_int, c_void_p,
                                                     c_void_p, c_size_t, c_void_p]),
    'ffk_frequency_shifts_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'ffk_frequency_shifts_shard_dev': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int,
                                               c_void_p, c_int, c_int, c_void_p, c_int, c_void_p,
                                               c_void_p, c_size_t, c_void_p]),
    'ffk_cumulant_function_second_order_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'ffk_cumulant_function_second_order_dev': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p,
                                                       c_void_p, c_void_p, c_size_t, c_void_p]),
    'ffk_filter_function_derivative': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                               c_void_p, c_int, c_void_p, c_void_p, c_int,
                                               c_void_p, c_void_p, c_void_p, c_int, c_int,
                                               c_void_p, c_int, c_void_p, c_void_p]),
    'ffk_control_matrix_derivative': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                              c_void_p, c_int, c_void_p, c_int, c_void_p,
                                              c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                              c_int, c_int, c_void_p]),
    'ffk_filter_function_derivative_from_control_matrix': (c_int, [c_void_p, c_void_p, c_int, c_int,
                                                                   c_int, c_int, c_int, c_void_p]),
    'ffk_filter_function_derivative_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    'ffk_filter_function_derivative_shard_dev': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                                         c_void_p, c_int, c_void_p, c_void_p, c_int,
                                                         c_void_p, c_void_p, c_void_p, c_int, c_int,
                                                         c_void_p, c_int, c_void_p, c_int, c_int,
                                                         c_void_p, c_void_p, c_void_p, c_size_t,
                                                         c_void_p]),
    'ffk_expm_real': (c_int, [c_void_p, c_int, c_void_p]),
    'ffk_error_transfer_matrix_workspace_bytes': (c_size_t, [c_int]),
    'ffk_error_transfer_matrix_dev': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    'ffk_liouville': (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    'ffk_liouville_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'ffk_liouville_dev': (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p,
                                  c_void_p, c_size_t, c_void_p]),
    'ffk_pipeline_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    'ffk_pipeline_dev': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int,
                                 c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                 c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_size_t, c_void_p]),
    'ffk_eigensolver_status_dev': (c_int, [c_void_p, c_size_t, c_int, c_int, c_void_p, c_void_p]),
    'ffk_kernel_fault_status': (c_int, [POINTER(c_int32), c_int]),
    'ffk_resident_create': (c_int, [POINTER(c_void_p)]),
    'ffk_resident_destroy': (c_int, [c_void_p]),
    'ffk_resident_release_pools': (c_int, []),
    'ffk_resident_filter_function': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                             c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                                             c_void_p, POINTER(c_void_p), POINTER(c_void_p),
                                             POINTER(c_void_p), POINTER(c_void_p)]),
    'ffk_resident_filter_function_from_controls': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                                           c_void_p, c_int, c_int, c_void_p, c_int,
                                                           c_void_p, c_int, c_void_p, c_int, c_void_p,
                                                           POINTER(c_void_p), POINTER(c_void_p),
                                                           POINTER(c_void_p), POINTER(c_void_p)]),
    'ffk_resident_timing': (c_int, [c_void_p, c_void_p]),
    'ffk_resident_control_matrix': (c_int, [c_void_p, c_void_p]),
    'ffk_resident_control_matrix_dev': (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_void_p),
                                                POINTER(c_void_p)]),
    'ffk_resident_filter_function_infidelity': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                                        c_void_p, c_int, c_int, c_void_p, c_int,
                                                        c_void_p, c_int, c_void_p, c_int, c_void_p,
                                                        c_void_p, c_int, c_int, c_void_p, c_int, c_int,
                                                        POINTER(c_void_p), POINTER(c_void_p),
                                                        POINTER(c_void_p), POINTER(c_void_p), c_void_p]),
    'ffk_resident_infidelity': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int,
                                        c_void_p]),
    'ffk_ipc_get_handle': (c_int, [c_void_p, c_void_p]),
    'ffk_ipc_open_handle': (c_int, [c_void_p, POINTER(c_void_p)]),
    'ffk_ipc_close_handle': (c_int, [c_void_p]),
    'ffk_peer_push_dev': (c_int, [c_void_p, c_size_t, c_void_p, c_void_p, ctypes.c_int64, c_int, c_int,
                                  c_void_p, c_void_p]),
    'ffk_peer_signal_dev': (c_int, [c_void_p, c_void_p, c_int, ctypes.c_int64, ctypes.c_int64,
                                    c_void_p, c_void_p]),
    'ffk_peer_set_timeout_ms': (c_int, [c_double]),
    'ffk_peer_wait_dev': (c_int, [c_void_p, c_int, ctypes.c_int64, c_void_p, c_void_p]),
    'ffk_peer_step_dev': (c_int, [c_void_p, c_size_t, c_void_p, c_void_p, ctypes.c_int64, c_void_p, c_void_p,
                                  c_void_p, c_int, c_int, ctypes.c_int64, c_void_p, c_void_p]),
    'ffk_graph_capture_begin': (c_int, [c_void_p]),
    'ffk_graph_capture_end': (c_int, [c_void_p, POINTER(c_void_p)]),
    'ffk_graph_capture_abort': (c_int, [c_void_p]),
    'ffk_graph_launch': (c_int, [c_void_p, c_void_p]),
    'ffk_graph_node_count': (c_int, [c_void_p, POINTER(c_int)]),
    'ffk_graph_destroy': (c_int, [c_void_p]),
    'ffk_stream_wait_event': (c_int, [c_void_p, c_void_p]),
    'ffk_set_segment_chunks': (c_int, [c_int]),
    'ffk_set_accumulate_variant': (c_int, [c_int]),
    'ffk_get_stats': (c_int, [POINTER(ffk_stats)]),
    'ffk_set_accumulate_events': (c_int, [c_void_p, c_void_p]),
    'ffk_set_accumulate_gate': (c_int, [c_void_p]),
}

_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm bundles its own libamdhip64 / libhsa-runtime64 (torch/lib).  Two HSA runtimes in
    one process do not both see the GPU: if libffk pulled in /opt/rocm's copy first, a later
    ``torch.cuda`` initialisation (device.py, parallel.py) fails with "No HIP GPUs are available".
    Both copies carry the same SONAME, so loading torch's copy first -- without importing torch --
    makes libffk and torch share one runtime whatever the import order."""
    import importlib.util
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], 'lib', 'libamdhip64.so')
    if os.path.exists(path):
        try:
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load libffk.so (once) and declare the prototypes.  Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f'{LIB_PATH} not found: build the HIP extension first '
                "(python -c 'import __graft_entry__ as g; g.build()' or "
                'make -C filter_functions_amd/csrc).  There is no CPU fallback.')
        _share_hip_runtime_with_torch()
        lib = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
    return _lib


def check(status):
    """Map a libffk status code to the exception class the reference would raise."""
    if status == FFK_OK:
        return
    msg = load().ffk_last_error().decode(errors='replace')
    if status == FFK_EINVAL:
        raise ValueError(msg)
    if status == FFK_ENOMEM:
        raise MemoryError(msg)
    if status == FFK_ENOCONV:
        raise np.linalg.LinAlgError(msg)
    if status == FFK_EKERNEL:
        raise FFKKernelFault(msg)
    raise FFKError(msg)


def check_kernel_fault(clear=True):
    """For callers of the ``_dev`` flavour, AFTER they synchronised their stream: raise
    :class:`FFKKernelFault` if a kernel stored a code in the library's sticky fault word since it was
    last cleared (include/ffk.h: ``ffk_kernel_fault_status``)."""
    word = c_int32(0)
    check(load().ffk_kernel_fault_status(ctypes.byref(word), 1 if clear else 0))
    if word.value:
        raise FFKKernelFault(f'a flag wait inside the d = 4 accumulate kernel ran out (code {word.value}): '
                             'the results of the launches since the last check are invalid')


def ptr(arr):
    """void* of a C-contiguous ndarray (or None)."""
    return None if arr is None else arr.ctypes.data_as(c_void_p)


def as_c128(arr):
    return np.ascontiguousarray(arr, dtype=np.complex128)


def as_f64(arr):
    return np.ascontiguousarray(arr, dtype=np.float64)


def device_count():
    n = c_int(0)
    status = load().ffk_device_count(ctypes.byref(n))
    return n.value if status == FFK_OK else 0


def device_info():
    name = ctypes.create_string_buffer(256)
    cus = c_int(0)
    mem = c_size_t(0)
    check(load().ffk_device_info(name, 256, ctypes.byref(cus), ctypes.byref(mem)))
    return name.value.decode(), cus.value, mem.value


def stats():
    out = ffk_stats()
    check(load().ffk_get_stats(ctypes.byref(out)))
    return {k: getattr(out, k) for k, _ in ffk_stats._fields_}
