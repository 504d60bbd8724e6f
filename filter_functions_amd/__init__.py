"""filter_functions_amd -- the numeric hot path of qutech/filter_functions on AMD MI355X.

A drop-in for ``PulseSequence.get_filter_function()`` and ``ff.infidelity()``::

    import filter_functions_amd as ff

    pulse = ff.PulseSequence(H_c, H_n, dt, basis=ff.Basis.pauli(2))
    F = pulse.get_filter_function(omega)
    infid = ff.infidelity(pulse, spectrum, omega)

The arithmetic runs in hand-written HIP kernels for gfx950 behind a C ABI
(``include/ffk.h``, ``filter_functions_amd/libffk.so``); this package is the Python host
side: the reference's object model, argument checking, caching and exceptions.
See DESIGN.md for the scope, INTEGRATION.md for the boundary.
"""
from . import analytic, basis, gradient, numeric, pulse_sequence, superoperator, util
from .basis import Basis
from .gradient import infidelity_derivative
from .numeric import error_transfer_matrix, infidelity
from .pulse_sequence import (PulseSequence, concatenate, concatenate_periodic,
                             concatenate_without_filter_function, extend, remap)
from .superoperator import liouville_representation

__all__ = ['analytic', 'Basis', 'PulseSequence', 'basis', 'concatenate', 'concatenate_periodic',
           'concatenate_without_filter_function',
           'error_transfer_matrix', 'extend', 'gradient', 'infidelity', 'infidelity_derivative',
           'liouville_representation', 'numeric',
           'pulse_sequence', 'remap', 'superoperator', 'util']

__version__ = '0.1.0'
