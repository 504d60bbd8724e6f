"""Gradient of the filter function and the infidelity with respect to the control amplitudes,
served by libffk's K9 kernels (``csrc/grad.hip``).

Same user-facing surface as ``filter_functions/gradient.py`` for the path:

=============================================  =============================================
this module                                    reference (file:line)
=============================================  =============================================
:func:`infidelity_derivative`                  gradient.py:559-676
:func:`filter_function_derivative`             pulse_sequence.py:977-1054
                                               (``PulseSequence.get_filter_function_derivative``)
=============================================  =============================================

The reference materialises the derivative of the control matrix, ``(n_ctrl, n_omega, n_dt, n_nops,
d**2)``, including a pairwise sum over all later propagators, and contracts it with the control
matrix.  :func:`infidelity_derivative` and :func:`filter_function_derivative` never form that
tensor: the derivative of the filter function is evaluated directly in Hilbert space with one
generator per (control, segment) -- see ``grad.hip``.  The reference's two tensor-level functions
are served for callers that want the tensor itself:

=====================================================  =====================================
:func:`calculate_derivative_of_control_matrix_from_scratch`  gradient.py:384-523
:func:`calculate_filter_function_derivative`                 gradient.py:526-556
=====================================================  =====================================
"""

import numpy as np

from . import _lib, util
from ._lib import as_c128, as_f64, check, ptr

__all__ = ['calculate_derivative_of_control_matrix_from_scratch',
           'calculate_filter_function_derivative', 'filter_function_derivative',
           'infidelity_derivative']


def _derivative(pulse, omega, control_identifiers, n_oper_identifiers, n_coeffs_deriv, spectrum):
    c_idx = util.get_indices_from_identifiers(pulse.c_oper_identifiers, control_identifiers)
    n_idx = util.get_indices_from_identifiers(pulse.n_oper_identifiers, n_oper_identifiers)
    G = len(pulse)
    if n_coeffs_deriv is not None:
        actual_shape = np.shape(n_coeffs_deriv)
        required_shape = (len(n_idx), len(c_idx), G)
        if actual_shape != required_shape:
            raise ValueError(f'Expected n_coeffs_deriv to be of shape {required_shape}, '
                             f'not {actual_shape}. Did you forget to specify identifiers?')
    pulse.omega = omega
    omega = as_f64(pulse.omega)
    D, V, Q = as_f64(pulse.eigvals), as_c128(pulse.eigvecs), as_c128(pulse.propagators)
    d = D.shape[1]
    if not 2 <= d <= 8:
        raise ValueError(f'The gradient kernels support 2 <= d <= 8, not d={d}.')
    B = as_c128(pulse.n_opers[n_idx])
    s = as_f64(pulse.n_coeffs[n_idx])
    C = as_c128(pulse.c_opers[c_idx])
    dt = as_f64(pulse.dt)
    t = np.concatenate(([0.0], dt.cumsum()))
    A, H, W = len(B), len(C), len(omega)
    ratio = None
    if n_coeffs_deriv is not None:
        ratio = as_f64(np.asarray(n_coeffs_deriv, dtype=float)/s[:, None, :])
    # the filter-function derivative crosses PCIe only if it is what was asked for
    dF = np.empty((A, G, H, W), dtype=np.float64) if spectrum is None else None
    dI = None
    S = None
    if spectrum is not None:
        # the reference parses the spectrum against all noise operators (gradient.py:667)
        S = util.parse_spectrum(spectrum, omega, range(len(pulse.n_opers)))
        if S.ndim == 3:
            raise ValueError('Expected spectrum of shape (n_omega,) or (n_nops, n_omega) for the '
                             'infidelity derivative.')
        if S.ndim == 2:
            if len(n_idx) != len(pulse.n_opers):
                raise ValueError(f'Spectrum of shape {S.shape} does not match {len(n_idx)} selected '
                                 'noise operators.')
        S = as_c128(S)
        dI = np.empty((A, G, H), dtype=np.float64)
    check(_lib.load().ffk_filter_function_derivative(
        ptr(D), ptr(V), ptr(Q), ptr(omega), W, ptr(B), A, ptr(s), ptr(C), H,
        ptr(ratio) if ratio is not None else None, ptr(dt), ptr(t), G, d,
        ptr(S) if S is not None else None, S.ndim if S is not None else 0,
        ptr(dF) if dF is not None else None, ptr(dI) if dI is not None else None))
    return dF, dI


def filter_function_derivative(pulse, omega, control_identifiers=None, n_oper_identifiers=None,
                               n_coeffs_deriv=None):
    r"""Derivative of the fidelity filter function
    :math:`\partial F_\alpha(\omega)/\partial u_h(t_g)`, shape ``(n_nops, n_dt, n_ctrl, n_omega)``
    (reference ``PulseSequence.get_filter_function_derivative``, pulse_sequence.py:977-1054).

    control_identifiers / n_oper_identifiers select and order the control and noise operators;
    n_coeffs_deriv, shape ``(n_nops, n_ctrl, n_dt)``, are the derivatives of the noise
    sensitivities by the control amplitudes (None: the sensitivities do not depend on them)."""
    return _derivative(pulse, omega, control_identifiers, n_oper_identifiers, n_coeffs_deriv, None)[0]


def infidelity_derivative(pulse, spectrum, omega, control_identifiers=None,
                          n_oper_identifiers=None, n_coeffs_deriv=None):
    r"""Derivative of the entanglement infidelity, :math:`\partial\mathcal I_\alpha/\partial u_h(t_g)
    = \frac{1}{2\pi d}\int d\omega\,S_\alpha(\omega)\,\partial F_\alpha(\omega)/\partial u_h(t_g)`,
    shape ``(n_nops, n_dt, n_ctrl)`` (reference gradient.py:559-676).  The filter-function
    derivative stays on the device; only the integrals come back."""
    return _derivative(pulse, omega, control_identifiers, n_oper_identifiers, n_coeffs_deriv,
                       spectrum)[1]


def calculate_derivative_of_control_matrix_from_scratch(omega, propagators, eigvals, eigvecs, basis,
                                                        t, dt, n_opers, n_coeffs, c_opers,
                                                        n_coeffs_deriv=None, intermediates=None):
    r"""Derivative of the control matrix
    :math:`\partial\tilde{\mathcal B}_{\alpha k}(\omega)/\partial u_h(t_g)`, shape ``(n_ctrl,
    n_omega, n_dt, n_nops, d**2)`` complex (reference gradient.py:384-523, same positional
    arguments).  *n_coeffs_deriv*, shape ``(n_nops, n_ctrl, n_dt)``; *intermediates* is accepted
    for signature compatibility and not needed (the device recomputes the segment integrals in
    registers, cheaper than shipping the cache across PCIe)."""
    omega, dt = as_f64(omega), as_f64(dt)
    D, V, Q = as_f64(eigvals), as_c128(eigvecs), as_c128(propagators)
    C = as_c128(np.asarray(basis))
    B, s, Hc = as_c128(n_opers), as_f64(n_coeffs), as_c128(c_opers)
    G, d = D.shape
    if not 2 <= d <= 8:
        raise ValueError(f'The gradient kernels support 2 <= d <= 8, not d={d}.')
    A, H, W, N = len(B), len(Hc), len(omega), len(C)
    t = np.concatenate(([0.0], dt.cumsum())) if t is None else as_f64(t)
    if t.shape != (G + 1,) or Q.shape != (G + 1, d, d) or V.shape != (G, d, d) or s.shape != (A, G):
        raise ValueError('Inconsistent shapes of eigvecs, propagators, t, n_coeffs for '
                         f'{G} segments and dimension {d}.')
    ratio = None
    if n_coeffs_deriv is not None:
        if np.shape(n_coeffs_deriv) != (A, H, G):
            raise ValueError(f'Expected n_coeffs_deriv to be of shape {(A, H, G)}, '
                             f'not {np.shape(n_coeffs_deriv)}.')
        ratio = as_f64(np.asarray(n_coeffs_deriv, dtype=float)/s[:, None, :])
    out = np.empty((H, W, G, A, N), dtype=np.complex128)
    check(_lib.load().ffk_control_matrix_derivative(
        ptr(D), ptr(V), ptr(Q), ptr(omega), W, ptr(C), N, ptr(B), A, ptr(s), ptr(Hc), H,
        ptr(ratio) if ratio is not None else None, ptr(dt), ptr(t), G, d, ptr(out)))
    return out


def calculate_filter_function_derivative(ctrlmat, ctrlmat_deriv):
    r"""Derivative of the fidelity filter function from the control matrix and its derivative,
    :math:`2\mathrm{Re}\sum_k\tilde{\mathcal B}^\ast_{\alpha k}\,\partial\tilde{\mathcal
    B}_{\alpha k}/\partial u_h(t_g)`: ``(n_nops, d**2, n_omega)``, ``(n_ctrl, n_omega, n_dt,
    n_nops, d**2)`` -> ``(n_nops, n_dt, n_ctrl, n_omega)`` (reference gradient.py:526-556)."""
    R, dR = as_c128(ctrlmat), as_c128(ctrlmat_deriv)
    if R.ndim != 3 or dR.ndim != 5:
        raise ValueError('Expected ctrlmat of shape (n_nops, d**2, n_omega) and ctrlmat_deriv of '
                         f'shape (n_ctrl, n_omega, n_dt, n_nops, d**2), not {R.shape}, {dR.shape}.')
    A, N, W = R.shape
    H, _, G, _, _ = dR.shape
    if dR.shape != (H, W, G, A, N):
        raise ValueError(f'ctrlmat_deriv of shape {dR.shape} does not match ctrlmat of shape '
                         f'{R.shape}.')
    out = np.empty((A, G, H, W), dtype=np.float64)
    check(_lib.load().ffk_filter_function_derivative_from_control_matrix(
        ptr(R), ptr(dR), A, N, W, G, H, ptr(out)))
    return out
