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

__all__ = ['tensor_transpose', 'tensor_insert', 'tensor_merge', 'hash_array_along_axis',
           'progressbar', 'embed_in_register', 'all_array_equal', 'remove_float_errors',
           'oper_equiv', 'paulis', 'abs2', 'cexp', 'cexpm1', 'get_indices_from_identifiers', 'parse_spectrum',
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
    if out is None:
        out = np.empty(x.shape, dtype=np.complex128)
    # NumPy's own cos and sin (not exp): this is what the device sincos is checked against
    np.cos(x, out=out.real, where=where)
    np.sin(x, out=out.imag, where=where)
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


def tensor(*args, rank=2, optimize=False):
    """Tensor product over the last *rank* axes of the arguments (Kronecker product of vectors
    for rank=1, of matrices for rank=2, ...), broadcasting over the leading axes (reference
    util.py:360-463).  *optimize* is accepted for signature compatibility; the product is formed
    by broadcasting, there is no contraction order to optimise."""
    if not args:
        raise ValueError('Require nonzero number of args!')

    def pair(a, b):
        a, b = np.asarray(a), np.asarray(b)
        if a.ndim < rank or b.ndim < rank:
            raise ValueError(f'Incompatible shapes {a.shape} and {b.shape} for tensor product '
                             f'of rank {rank}.')
        da, db = a.shape[a.ndim - rank:], b.shape[b.ndim - rank:]
        # factor axes interleaved: a's at even positions, b's at odd ones
        a = a.reshape(a.shape[:a.ndim - rank] + tuple(x for n in da for x in (n, 1)))
        b = b.reshape(b.shape[:b.ndim - rank] + tuple(x for n in db for x in (1, n)))
        try:
            out = a*b
        except ValueError as err:
            raise ValueError(f'Incompatible shapes {a.shape} and {b.shape} for tensor product '
                             f'of rank {rank}.') from err
        return out.reshape(out.shape[:out.ndim - 2*rank] + tuple(x*y for x, y in zip(da, db)))
    return functools.reduce(pair, args)


def _check_dims(name, dims, rank):
    if not len(dims) == rank:
        raise ValueError(f'{name}_dims should be of length rank = {rank}, not {len(dims)}')
    if len({len(axis) for axis in dims}) != 1:
        raise ValueError(f'Require all lists in {name}_dims to be of same length!')


def _chain_after_insertion(n_arr, pos, n_ins, sort_wrapped):
    """Positions in the product chain ``[arr factors..., ins factors...]`` in the order they take
    after ins factor j has been inserted before arr factor ``pos[j]`` (negative positions count
    from the end).  Insertions happen in ascending order of position, each shifting the later
    ones by one -- the rule of numpy.insert.  The reference orders them by the wrapped position
    in tensor_insert (util.py:617-623) but by the position as given in tensor_merge (:748); the
    two differ for mixed signs, and both are reproduced."""
    wrapped = []
    for p in pos:
        if not -n_arr <= p <= n_arr:
            raise IndexError(f'Invalid position {p} specified. Must be between -{n_arr} and '
                             f'{n_arr}.')
        wrapped.append(p + n_arr if p < 0 else p)
    keys = wrapped if sort_wrapped else pos
    chain = list(range(n_arr))
    for shift, j in enumerate(sorted(range(n_ins), key=lambda j: (keys[j], j))):
        chain.insert(wrapped[j] + shift, n_arr + j)
    return chain


def tensor_merge(arr, ins, pos, arr_dims, ins_dims, rank=2, optimize=False, _sort_wrapped=False):
    """Merge the product chain *ins* into the product chain *arr*: constituent j of *ins* goes
    before the constituent ``pos[j]`` of *arr* (reference util.py:640-780), e.g.
    ``tensor_merge(tensor(X, Y, Z), tensor(I, I), pos=[1, 2], ...) == tensor(X, I, Y, I, Z)``.
    arr_dims / ins_dims: per tensor axis the dimensions of the constituents.

    Formed as the plain product ``arr (x) ins`` followed by one re-ordering of the factors
    (:func:`tensor_transpose`), instead of the reference's einsum over split axes."""
    _check_dims('arr', arr_dims, rank)
    _check_dims('ins', ins_dims, rank)
    n_arr, n_ins = len(arr_dims[0]), len(ins_dims[0])
    pos = [int(p) for p in pos]
    if len(pos) != n_ins:
        raise ValueError('Expected one position per constituent of ins, i.e. '
                         f'{n_ins}, not {len(pos)}')
    arr, ins = np.asarray(arr), np.asarray(ins)
    for name, a, dims in (('arr', arr, arr_dims), ('ins', ins, ins_dims)):
        if tuple(int(np.prod(axis)) for axis in dims) != a.shape[a.ndim - rank:]:
            raise ValueError(f'{name}_dims {dims} do not match the shape {a.shape} of {name}.')
    order = _chain_after_insertion(n_arr, pos, n_ins, _sort_wrapped)
    dims = [list(a) + list(i) for a, i in zip(arr_dims, ins_dims)]
    return tensor_transpose(tensor(arr, ins, rank=rank), order, dims, rank=rank)


def tensor_insert(arr, *args, pos, arr_dims, rank=2, optimize=False):
    """Insert *args* into the product chain *arr* (reference util.py:466-637): with an integer
    *pos* all of them, in a row, before constituent *pos*; with a sequence, ``args[j]`` before the
    constituent ``pos[j]`` of the original chain -- the rule of :func:`numpy.insert`."""
    if len(args) == 0:
        raise ValueError('Require nonzero number of args!')
    if np.issubdtype(type(pos), np.integer):
        args = (tensor(*args, rank=rank),)
        pos = (int(pos),)
    elif len(pos) != len(args):
        raise ValueError('Expected pos to be either an int or a sequence of the same length '
                         f'as the number of args, not length {len(pos)}')
    _check_dims('arr', arr_dims, rank)
    args = [np.asarray(a) for a in args]
    for k, a in enumerate(args):
        if a.ndim < rank:
            raise ValueError(f'Could not insert arg {k} with shape {a.shape} into the array '
                             f'with shape {np.shape(arr)} at position {pos[k]}.')
    ins_dims = [[a.shape[a.ndim - rank + axis] for a in args] for axis in range(rank)]
    try:
        return tensor_merge(arr, tensor(*args, rank=rank), pos, arr_dims, ins_dims, rank=rank,
                            _sort_wrapped=True)
    except ValueError as err:
        raise ValueError(f'Could not insert args with shapes {[a.shape for a in args]} into the '
                         f'array with shape {np.shape(arr)} at positions {list(pos)}.') from err


def tensor_transpose(arr, order, arr_dims, rank=2):
    """Re-order the factors of a tensor product: for ``arr == tensor(A, B, C)`` and
    ``order == (1, 2, 0)`` the result is ``tensor(B, C, A)`` (reference util.py:783-860).
    arr_dims: for each of the last *rank* axes the dimensions of the factors along it."""
    arr = np.asarray(arr)
    if len(arr_dims) != rank or len({len(dims) for dims in arr_dims}) != 1:
        raise ValueError(f'arr_dims should be {rank} lists of equal length')
    n = len(arr_dims[0])
    lead = arr.shape[:arr.ndim - rank]
    try:
        split = arr.reshape(lead + tuple(dim for dims in arr_dims for dim in dims))
    except ValueError as err:
        raise ValueError('arr_dims not compatible with arr.shape[-rank:] = '
                         f'{arr.shape[-rank:]}') from err
    try:
        order = [int(o) for o in order]
    except (TypeError, ValueError) as err:
        raise TypeError("Could not transpose the order. Are all elements of 'order' integers?") \
            from err
    if sorted(order) != list(range(n)):
        raise ValueError("Could not transpose the order. Are all elements of 'order' unique and "
                         "match the array?")
    axes = list(range(len(lead))) + [len(lead) + r*n + o for r in range(rank) for o in order]
    return split.transpose(axes).reshape(arr.shape)


def embed_in_register(arr, qubits, n_qubits, d_per_qubit=2, rank=2):
    """The stack of operators (rank=2) or diagonals (rank=1) *arr*, defined on the qubits
    *qubits* (ascending) of a register of *n_qubits*, on the whole register: identity (ones) on
    all other qubits, tensor factors in register order.  (What the reference assembles step by step
    with util.tensor_insert / tensor_merge, util.py:466-780.)"""
    qubits = [int(q) for q in qubits]
    rest = [q for q in range(n_qubits) if q not in qubits]
    arr = np.asarray(arr)
    if rest:
        filler = np.eye(d_per_qubit**len(rest)) if rank == 2 else np.ones(d_per_qubit**len(rest))
        arr = tensor(arr, filler, rank=rank)
    current = qubits + rest
    order = [current.index(q) for q in range(n_qubits)]
    return tensor_transpose(arr, order, [[d_per_qubit]*n_qubits]*rank, rank=rank)


def all_array_equal(it):
    """Are all arrays produced by the iterable equal (reference util.py:1103-1109)?"""
    arrays = list(it)
    return all(np.array_equal(arrays[0], a) for a in arrays[1:]) if arrays else True


def remove_float_errors(arr, eps_scale=None):
    """Set entries whose magnitude is below eps_scale*eps to zero, separately for real and
    imaginary parts (reference util.py:909-938)."""
    eps_scale = 1 if eps_scale is None else eps_scale
    atol = np.finfo(np.float64).eps*eps_scale
    arr = np.array(arr, copy=True)
    if np.iscomplexobj(arr):
        arr.real[np.abs(arr.real) <= atol] = 0
        arr.imag[np.abs(arr.imag) <= atol] = 0
    else:
        arr[np.abs(arr) <= atol] = 0
    return arr


def oper_equiv(psi, phi, eps=None, normalized=False):
    """Are two operators (or states) equal up to a global phase?  Returns (bool, phase) with
    ``psi == exp(-1j*phase)*phi`` -- the convention of reference util.py:941-1010."""
    psi, phi = np.atleast_2d(np.asarray(psi), np.asarray(phi))
    if eps is None:
        eps = max(np.finfo(psi.dtype).eps if np.issubdtype(psi.dtype, np.inexact) else 0,
                  np.finfo(phi.dtype).eps if np.issubdtype(phi.dtype, np.inexact) else 0,
                  np.finfo(float).eps)*np.prod(psi.shape)*phi.shape[-1]*2
        if not normalized:
            eps *= (np.prod(psi.shape[-2:])*phi.shape[-1]*2)**2
    try:
        inner = np.einsum('...ij,...ij', psi.conj(), phi)
    except ValueError as err:
        raise ValueError('psi and phi have incompatible dimensions!') from err
    norm = 1 if normalized else np.sqrt(np.einsum('...ij,...ij', psi.conj(), psi).real
                                        * np.einsum('...ij,...ij', phi.conj(), phi).real)
    return abs(norm - abs(inner)) <= eps, np.angle(inner)


def hash_array_along_axis(arr, axis=0):
    """Hashes of the slices of *arr* along *axis*; -0.0 and 0.0 hash alike (reference
    util.py:1096-1100)."""
    return [hash((a + 0.0).tobytes()) for a in np.swapaxes(np.asarray(arr), 0, axis)]


def progressbar(iterable, *args, **kwargs):
    """tqdm progress bar around *iterable* where tqdm is installed, the bare iterable otherwise
    (reference util.py:1112-1121)."""
    try:
        from tqdm import tqdm
    except ImportError:
        return iterable
    return tqdm(iterable, *args, **kwargs)


def progressbar_range(*args, show_progressbar=False, **kwargs):
    """API-compatibility shim: the per-segment Python loop this used to decorate
    (reference numeric.py:846) no longer exists; there is nothing to show progress of."""
    return range(*args)


def dot_HS(U, V, eps=None):
    """Hilbert-Schmidt inner product tr(U^dag V) (reference util.py:909-956)."""
    res = np.einsum('...ij,...ij', np.conjugate(U), V)
    return res.real if np.isreal(res).all() else res

