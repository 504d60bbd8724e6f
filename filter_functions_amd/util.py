"""Host-side helpers of the filter-function path.

Mirrors the parts of ``filter_functions/util.py`` the hot path's callers use
(argument parsing, identifier lookup, spectrum validation, sample frequencies).
These are O(inputs) bookkeeping on tiny arrays; the numerics of the path itself
run in libffk (see :mod:`filter_functions_amd.numeric`).
"""
import functools
import inspect
from itertools import accumulate

import numpy as np

__all__ = ['paulis', 'abs2', 'cexp', 'cexpm1', 'get_indices_from_identifiers', 'parse_spectrum',
           'parse_operators', 'parse_optional_parameters', 'is_sequence_like', 'integrate',
           'get_sample_frequencies', 'mdot', 'adot', 'tensor', 'CalculationError',
           'progressbar_range']

#: identity and the three Pauli matrices (reference util.py:109-118)
paulis = np.array([[[1, 0], [0, 1]],
                   [[0, 1], [1, 0]],
                   [[0, -1j], [1j, 0]],
                   [[1, 0], [0, -1]]], dtype=complex)


class CalculationError(Exception):
    """Raised when a requested quantity cannot be computed from what is cached
    (reference util.py:1146-1150)."""


def abs2(x):
    """|x|**2 without the square root (reference util.py:120-133)."""
    x = np.asarray(x)
    return x.real**2 + x.imag**2


def cexp(x, out=None, where=True):
    """exp(ix) = cos x + i sin x (reference util.py:136-162); host helper for O(W) vectors."""
    x = np.asarray(x)
    out = np.empty(x.shape, dtype=np.complex128) if out is None else out
    out.real = np.cos(x, out=out.real, where=where)
    out.imag = np.sin(x, out=out.imag, where=where)
    return out


def cexpm1(x, out=None, where=True):
    """exp(ix) - 1 = -2 sin^2(x/2) + i sin x (reference util.py:165-182)."""
    x = np.asarray(x)
    out = np.empty(x.shape, dtype=np.complex128) if out is None else out
    half = np.divide(x, 2, where=where)
    half = np.sin(half, out=half, where=where)
    out.real = np.multiply(-2, np.square(half, where=where), where=where, out=out.real)
    out.imag = np.sin(x, out=out.imag, where=where)
    return out


def is_sequence_like(obj):
    """True for anything indexable with a length (lists, tuples, ndarrays, ...)."""
    try:
        len(obj)
    except TypeError:
        return False
    return hasattr(obj, '__getitem__')


def parse_optional_parameters(**allowed_kwargs):
    """Decorator: raise ``ValueError`` if a named argument is not one of the allowed values
    (same contract as reference util.py:185-211)."""
    def decorator(func):
        names = tuple(inspect.signature(func).parameters)
        defaults = {k: v.default for k, v in inspect.signature(func).parameters.items()}

        @functools.wraps(func)
        def wrapper(*args, **kwargs):
            for name, allowed in allowed_kwargs.items():
                pos = names.index(name)
                value = args[pos] if pos < len(args) else kwargs.get(name, defaults[name])
                if value not in allowed:
                    raise ValueError(f'Invalid value for {name}: {value}. '
                                     f'Should be one of {allowed}.')
            return func(*args, **kwargs)
        return wrapper
    return decorator


def parse_operators(opers, err_loc):
    """Turn a sequence of operators (ndarray or objects exposing ``full()`` / ``todense()``)
    into a (n, d, d) complex array (reference util.py:230-268)."""
    parsed = []
    for oper in opers:
        if isinstance(oper, np.ndarray):
            parsed.append(oper.squeeze())
        elif hasattr(oper, 'full'):
            parsed.append(oper.full())
        elif hasattr(oper, 'todense'):
            parsed.append(np.asarray(oper.todense()))
        else:
            raise TypeError(f'Expected operators in {err_loc} to be NumPy arrays or QuTiP Qobjs! '
                            f'Instead got {type(oper)}.')
    if not all(op.ndim == 2 for op in parsed):
        raise ValueError(f'Expected all operators in {err_loc} to be two-dimensional!')
    if len(set(op.shape for op in parsed)) != 1 or parsed[0].shape[0] != parsed[0].shape[1]:
        raise ValueError(f'Expected operators in {err_loc} to be square and of equal dimension!')
    return np.asarray(parsed, dtype=complex)


def get_indices_from_identifiers(all_identifiers, identifiers):
    """Positions of *identifiers* in *all_identifiers* (reference util.py:331-357)."""
    table = {identifier: i for i, identifier in enumerate(all_identifiers)}
    if identifiers is None:
        return np.arange(len(all_identifiers))
    try:
        if isinstance(identifiers, str):
            return np.array([table[identifiers]])
        return np.array([table[identifier] for identifier in identifiers])
    except KeyError:
        raise ValueError('Invalid identifiers given. All available ones '
                         f'are: {all_identifiers}') from None


def parse_spectrum(spectrum, omega, idx):
    """Broadcast a spectrum to ([[n_idx,] n_idx,] n_omega) and validate it
    (reference util.py:214-227)."""
    spectrum = np.asanyarray(spectrum)
    shape = (len(idx),)*(spectrum.ndim - 1) + (len(omega),)
    try:
        spectrum = np.broadcast_to(spectrum, shape)
    except ValueError as err:
        raise ValueError(f'Spectrum should be of shape {shape}, not {spectrum.shape}.') from err
    if spectrum.ndim == 3 and not np.allclose(spectrum, spectrum.conj().swapaxes(0, 1)):
        raise ValueError('Cross-spectra given but not Hermitian along first two axes')
    if spectrum.ndim > 3:
        raise ValueError(f'Expected spectrum to have < 4 dimensions, not {spectrum.ndim}')
    return spectrum


def integrate(f, x=None, dx=1.0):
    """Trapezoid rule along the last axis (reference util.py:880-906)."""
    f = np.asanyarray(f)
    steps = np.diff(x) if x is not None else dx
    summed = f[..., 1:] + f[..., :-1]
    summed = summed*steps
    return summed.sum(axis=-1)/2


def get_sample_frequencies(pulse, n_samples=300, spacing='log', include_quasistatic=False,
                           omega_min=None, omega_max=None):
    """Default frequency grid for a pulse (reference util.py:1054-1093)."""
    if omega_min is None:
        omega_min = 2*np.pi*1e-2/pulse.tau
    if omega_max is None:
        omega_max = 2*np.pi*1e+1/pulse.dt.min()
    space = np.geomspace if spacing == 'log' else np.linspace
    omega = space(omega_min, omega_max, n_samples - include_quasistatic)
    if include_quasistatic:
        omega = np.insert(omega, 0, 0)
    return omega


def mdot(arr, axis=0):
    """Product of the matrices stacked along *axis* (reference util.py:863-865)."""
    return functools.reduce(np.matmul, np.swapaxes(arr, 0, axis))


def adot(arr, axis=0):
    """Running left-products: out[i] = arr[i] @ ... @ arr[0] (reference util.py:868-877)."""
    arr = np.swapaxes(arr, 0, axis)
    return np.array(list(accumulate(arr, lambda acc, new: new @ acc))).swapaxes(0, axis)


def tensor(*args):
    """Kronecker product of matrices (the rank-2 case of reference util.py:360-457)."""
    return functools.reduce(np.kron, args)


def progressbar_range(*args, show_progressbar=False, **kwargs):
    """API-compatibility shim: the per-segment Python loop this used to decorate
    (reference numeric.py:846) no longer exists; there is nothing to show progress of."""
    return range(*args)


def dot_HS(U, V, eps=None):
    """Hilbert-Schmidt inner product tr(U^dag V) (reference util.py:909-956)."""
    res = np.einsum('...ij,...ij', np.conjugate(U), V)
    return res.real if np.isreal(res).all() else res

