"""``PulseSequence``: the object model and memoising getters in front of the hot path.

Behavioural twin of the part of ``filter_functions/pulse_sequence.py`` that calls into the numeric
kernels (constructor and input validation 272-310, ``from_arrays`` 312-359, caches and
``is_cached`` 508-538, ``diagonalize`` 577-586, ``get_control_matrix`` / ``cache_control_matrix``
588-677, ``get_filter_function`` / ``cache_filter_function`` 691-902, total phases 1056-1084,
the cached-data properties 1086-1169 and ``cleanup`` 1188-1245): the same three cache
dictionaries and keys, the same invalidation rule (assigning a different ``omega`` drops all
frequency-dependent data) and the same exceptions.

The implementation is its own: the caches are :class:`~filter_functions_amd._resident.LazyCache`
dictionaries whose entries may still live on the GPU.  A pulse with nothing cached evaluates
``get_filter_function`` in ONE library call (``ffk_resident_filter_function``: one H2D, the fused
device pass, one D2H of the eigensystem and F); the control matrix stays in HBM and the
by-products the reference computes eagerly (total phases, Liouville propagator) are produced the
first time somebody reads them.  From the outside every entry is an ndarray, exactly as in the
reference.
"""
import copy
import functools
import os
import weakref
import zlib
from collections import OrderedDict
from collections.abc import Mapping
from types import MappingProxyType
from warnings import warn

import numpy as np

from . import numeric, util
from ._resident import Deferred, LazyCache, ResidentResult
from .basis import Basis
from .superoperator import liouville_representation

__all__ = ['PulseSequence', 'concatenate', 'concatenate_without_filter_function']

_DIAGONALIZATION = ('eigvals', 'eigvecs', 'propagators')

# human-readable names accepted by is_cached (reference pulse_sequence.py:508-538) -> cache key
_DATA_KEYS = {
    'eigenvalues': 'eigvals', 'eigenvectors': 'eigvecs', 'propagators': 'propagators',
    'total propagator': 'total_propagator',
    'total propagator liouville': 'total_propagator_liouville',
}
_FREQUENCY_KEYS = {
    'frequencies': 'omega', 'total phases': 'total_phases',
    'filter function': 'filter_function', 'fidelity filter function': 'filter_function',
    'generalized filter function': 'filter_function_gen',
    'pulse correlation filter function': 'filter_function_pc',
    'fidelity pulse correlation filter function': 'filter_function_pc',
    'generalized pulse correlation filter function': 'filter_function_pc_gen',
    'second order filter function': 'filter_function_2',
    'control matrix': 'control_matrix',
    'pulse correlation control matrix': 'control_matrix_pc',
}
# cleanup(method): keys dropped from (_data, _frequency_data), and whether intermediates go too
_CLEANUP = {
    'conservative': (_DIAGONALIZATION, (), False),
    'greedy': (_DIAGONALIZATION + ('total_propagator', 'total_propagator_liouville'),
               ('total_phases', 'control_matrix', 'control_matrix_pc'), True),
}
_NOT_CONCATENATED = ("Could not get the pulse correlation {} since it "
                     "was not computed during concatenation. Please run the "
                     "concatenation again with 'calc_pulse_correlation_FF' set to True.")


def _parse_hamiltonian(H, n_dt, H_str):
    """Unpack ``[[oper, coeffs(, identifier)], ...]`` into arrays sorted by identifier
    (contract of reference pulse_sequence.py:1286-1337)."""
    if not util.is_sequence_like(H):
        raise TypeError(f'Expected {H_str} to be a sequence, not of type {type(H)}!')
    opers, coeffs, identifiers = [], [], []
    for term in H:
        if not util.is_sequence_like(term):
            raise TypeError(f'Expected {H_str} to be a sequence of sequences but found at least one '
                            'item of H not a sequence!')
        term = tuple(term)
        opers.append(term[0] if term else None)
        coeffs.append(term[1] if len(term) > 1 else None)
        identifiers.append(term[2] if len(term) > 2 else None)
    for coeff in coeffs:
        if not util.is_sequence_like(coeff):
            raise TypeError(f'Expected coefficients in {H_str} to be a sequence')
    prefix = 'A' if H_str == 'H_c' else 'B'
    named = [ident is not None for ident in identifiers]
    identifiers = [ident if given else f'{prefix}_{i}'
                   for i, (ident, given) in enumerate(zip(identifiers, named))]
    if any(named) and len(set(identifiers)) != len(identifiers):
        raise ValueError(f'{H_str} identifiers should be unique')
    for coeff in coeffs:
        if len(coeff) != n_dt:
            raise ValueError(f'Expected all coefficients in {H_str} to be of len(dt) = {n_dt}!')
    order = np.argsort(identifiers)
    return (util.parse_operators(opers, H_str)[order], np.asarray(identifiers)[order],
            np.asarray(coeffs)[order])


def _checked_durations(dt):
    if not util.is_sequence_like(dt):
        raise TypeError(f'Expected a sequence of time steps, not {type(dt)}')
    dt = np.asarray(dt)
    if np.iscomplexobj(dt) and dt.imag.any():
        raise ValueError('Times dt are not (all) real!')
    if (dt.real < 0).any():
        raise ValueError('Time steps are not (all) positive!')
    return dt


@functools.lru_cache(maxsize=32)
def _default_basis(d):
    """The default basis of dimension d, ONE read-only object per d: pulses built without a basis
    share it, so that concatenating them compares bases by identity (the arrays are what
    ``Basis.ggm(d)`` returns; writing into a default basis in place raises)."""
    basis = Basis.ggm(d)
    basis.flags.writeable = False
    return basis


def _checked_basis(basis, d):
    if basis is None:
        return _default_basis(d)
    if not isinstance(basis, Basis):
        raise ValueError("Expected basis to be an instance of the "
                         f"'filter_functions_amd.basis.Basis' class, not {type(basis)}!")
    if basis.shape[1:] != (d, d):
        raise ValueError(f"Expected basis elements to be of shape ({d}, {d}), "
                         f"not {basis.shape[1:]}!")
    return basis


#: Default of ``PulseSequence.get_filter_function(..., writable=)``: hand out owned, writable arrays (the reference's
#: convention) instead of read-only views of the resident results.  From the environment variable FFK_WRITABLE_RESULTS.
WRITABLE_RESULTS = os.environ.get('FFK_WRITABLE_RESULTS', '') not in ('', '0')


class PulseSequence:
    r"""A piecewise-constant control pulse :math:`H_c(t)=\sum_i a_i(t)A_i` with noise
    :math:`H_n(t)=\sum_\alpha s_\alpha(t) b_\alpha(t) B_\alpha` (reference
    pulse_sequence.py:61-271).

    Parameters
    ----------
    H_c, H_n: ``[[operator, coefficients(, identifier)], ...]`` with operators (d, d) and
        coefficients of length ``len(dt)``.
    dt: sequence of segment durations.
    basis: :class:`~filter_functions_amd.basis.Basis`, optional (default ``Basis.ggm(d)``).
    """
    __array_interface__ = {'shape': (), 'typestr': '|O', 'version': 3}

    @classmethod
    def _blank(cls):
        """An instance with empty caches and no Hamiltonian yet."""
        new = object.__new__(cls)
        new._data, new._frequency_data, new._intermediates = LazyCache(), LazyCache(), LazyCache()
        new._resident = None
        return new

    def __new__(cls, *args, **kwargs):
        return cls._blank()

    def __init__(self, H_c, H_n, dt, basis=None):
        self.dt = _checked_durations(dt)
        control = _parse_hamiltonian(H_c, len(self.dt), 'H_c')
        noise = _parse_hamiltonian(H_n, len(self.dt), 'H_n')
        if control[0].shape[-2:] != noise[0].shape[-2:]:
            raise ValueError('Control and noise Hamiltonian not same dimension!')
        self.c_opers, self.c_oper_identifiers, self.c_coeffs = control
        self.n_opers, self.n_oper_identifiers, self.n_coeffs = noise
        self.d = self.c_opers.shape[-1]
        self.basis = _checked_basis(basis, self.d)

    @classmethod
    def from_arrays(cls, c_opers, c_oper_identifiers, c_coeffs, n_opers, n_oper_identifiers,
                    n_coeffs, dt, basis=None):
        """Alternative constructor from already-parsed arrays (contract of reference
        pulse_sequence.py:312-359): no sorting, consistency checks only."""
        new = cls._blank()
        given = dict(c_opers=c_opers, c_oper_identifiers=c_oper_identifiers, c_coeffs=c_coeffs,
                     n_opers=n_opers, n_oper_identifiers=n_oper_identifiers, n_coeffs=n_coeffs,
                     dt=dt)
        for name, value in given.items():
            setattr(new, name, np.asanyarray(value))
        new.d = new.c_opers.shape[-1]
        new.basis = _default_basis(new.d) if basis is None else np.asanyarray(basis).view(Basis)
        n_control = {len(new.c_opers), len(new.c_oper_identifiers), len(new.c_coeffs)}
        n_noise = {len(new.n_opers), len(new.n_oper_identifiers), len(new.n_coeffs)}
        extents = set(new.c_opers.shape[1:] + new.n_opers.shape[1:])
        n_steps = {new.dt.size, new.c_coeffs.shape[1], new.n_coeffs.shape[1]}
        for consistent, complaint in (
                (len(n_control) == 1, 'Control Hamiltonian not same length!'),
                (len(n_noise) == 1, 'Noise Hamiltonian not same length!'),
                (len(extents) == 1, 'Control and/or noise Hamiltonian not same, square dimension!'),
                (len(n_steps) == 1, 'Time steps not same length!'),
                (new.basis.d == new.d, 'Basis dimension not same as Hamiltonian dimension!')):
            if not consistent:
                raise ValueError(complaint)
        return new

    def __repr__(self):
        return f'PulseSequence with total duration {self.tau}'

    def __str__(self):
        return f'{repr(self)}\n\tof dimension {self.d} and duration {self.duration}'

    def __len__(self):
        return len(self.dt)

    def __getitem__(self, key):
        """A slice of the pulse as a new PulseSequence (reference pulse_sequence.py:440-484)."""
        durations = np.atleast_1d(self.dt[key])
        if durations.size == 0:
            raise IndexError('Cannot create empty PulseSequence')

        def columns(table):
            return np.atleast_2d(table.T[key]).T
        new = type(self).from_arrays(self.c_opers, self.c_oper_identifiers, columns(self.c_coeffs),
                                     self.n_opers, self.n_oper_identifiers, columns(self.n_coeffs),
                                     durations, self.basis)
        # a leading slice can reuse the running sum of the control matrix, if that was kept
        leading = isinstance(key, slice) and key.start in (None, 0) and key.step in (None, 1)
        if leading and 'control_matrix_step_cumulative' in self._intermediates:
            running = self._intermediates['control_matrix_step_cumulative']
            new.cache_control_matrix(self.omega, running[key.stop - 1])
        return new

    def __copy__(self):
        twin = self._blank()
        twin.__dict__.update(self.__dict__)
        for cache in ('_data', '_frequency_data', '_intermediates'):
            setattr(twin, cache, getattr(self, cache).copy())
        # by-products still due are re-deferred on the twin (their producers must not depend on
        # the original staying alive)
        due = [(cache, key) for cache, key in ((twin._frequency_data, 'total_phases'),
                                               (twin._data, 'total_propagator_liouville'))
               if key in cache and type(cache.peek(key)) is Deferred]
        for cache, key in due:
            dict.__delitem__(cache, key)
        if due:
            twin._defer_by_products()
        return twin

    def copy(self):
        return self.__copy__()

    def __matmul__(self, other):
        if not isinstance(other, self.__class__):
            raise TypeError(f'Incompatible type for concatenation: {type(other)}')
        return concatenate((self, other))

    def __imatmul__(self, other):
        raise NotImplementedError

    def __eq__(self, other):
        """Two sequences are equal if dt, operators, identifiers, coefficients and basis agree,
        after merging consecutive segments with identical control (reference
        pulse_sequence.py:363-440)."""
        if not isinstance(other, self.__class__):
            return NotImplemented
        atol = np.finfo(complex).eps*self.basis.shape[0]
        mine, theirs = _merge_constant_segments(self), _merge_constant_segments(other)
        if len(mine[2]) != len(theirs[2]) or not np.allclose(mine[2], theirs[2], 1e-10, atol):
            return False
        for opers, idents, k in (('c_opers', 'c_oper_identifiers', 0),
                                 ('n_opers', 'n_oper_identifiers', 1)):
            ia = np.argsort(getattr(self, idents))
            ib = np.argsort(getattr(other, idents))
            if len(ia) != len(ib):
                return False
            if not np.array_equal(getattr(self, opers)[ia], getattr(other, opers)[ib]):
                return False
            if not np.array_equal(getattr(self, idents)[ia], getattr(other, idents)[ib]):
                return False
            if not np.array_equal(mine[k][ia], theirs[k][ib]):
                return False
        return bool(self.basis == other.basis)

    __hash__ = None

    def propagator_at_arb_t(self, t):
        """Cumulative propagator Q(t) at arbitrary times: the segment propagator up to the
        enclosing step times the partial evolution within it (reference
        pulse_sequence.py:1247-1267).  Small host-side matrices, not on the accelerated path."""
        self.diagonalize()
        t = np.asarray(t, dtype=float)
        idx = np.clip(np.searchsorted(self.t, t) - 1, 0, len(self.dt) - 1)
        V = self.eigvecs[idx]
        phase = util.cexp((self.t[idx] - t)[:, None]*self.eigvals[idx])
        return (V*phase[:, None, :]) @ V.conj().swapaxes(-1, -2) @ self.propagators[idx]

    # ---- caches --------------------------------------------------------------------------
    def is_cached(self, attr):
        """True if *attr* is cached; accepts the reference's human-readable aliases
        (pulse_sequence.py:508-538).  Never triggers a computation or a transfer."""
        spoken = attr.lower().replace('_', ' ')
        if spoken in _DATA_KEYS:
            return _DATA_KEYS[spoken] in self._data
        if spoken in _FREQUENCY_KEYS:
            return _FREQUENCY_KEYS[spoken] in self._frequency_data
        return any(attr in cache for cache in (self._data, self._frequency_data,
                                               self._intermediates))

    @property
    def data(self):
        return MappingProxyType(self._data)

    @property
    def frequency_data(self):
        return MappingProxyType(self._frequency_data)

    @property
    def intermediates(self):
        return MappingProxyType(self._intermediates)

    @property
    def t(self):
        """Absolute segment times [0, cumsum(dt)] (reference pulse_sequence.py:541-544)."""
        if 't' not in self._data:
            self._data['t'] = np.concatenate(([0], self.dt.cumsum()))
        return self._data['t']

    @t.setter
    def t(self, val):
        self._data['t'] = val

    @property
    def tau(self):
        if 'tau' not in self._data:
            self._data['tau'] = self._data['t'][-1] if 't' in self._data else self.dt.sum()
        return self._data['tau']

    @tau.setter
    def tau(self, val):
        self._data['tau'] = val

    @property
    def duration(self):
        return self.tau

    # ---- the hot path ----------------------------------------------------------------------
    def _hamiltonian(self):
        """H[g] = sum_i a_i(t_g) A_i as one (n_dt x n_cops)(n_cops x d^2) matrix product (the
        einsum formulation of the same sum costs 55 us at config 2, half the device pass)."""
        # (the user may override self.d: take the shape from the operators.)  One REAL product against
        # the operators viewed as (n_cops, 2 d^2) doubles: the same numbers without first promoting
        # the (n_dt, n_cops) amplitudes to complex
        opers = np.ascontiguousarray(self.c_opers, dtype=np.complex128)
        flat = np.asarray(self.c_coeffs, dtype=np.float64).T @ opers.reshape(len(opers), -1).view(np.float64)
        return flat.view(np.complex128).reshape((-1,) + opers.shape[1:])

    def diagonalize(self):
        """Diagonalise the control Hamiltonian (reference pulse_sequence.py:577-586)."""
        if any(key not in self._data for key in _DIAGONALIZATION):
            results = numeric.diagonalize(self._hamiltonian(), self.dt)
            self._data.update(zip(_DIAGONALIZATION, results))
        self._data['total_propagator'] = self._data['propagators'][-1]

    def _resident_pass_applies(self, which, order, cache_intermediates):
        """The one-call evaluation serves the plain request on a pulse that has nothing to reuse."""
        return (order == 1 and which == 'fidelity' and not cache_intermediates
                and len(self.omega) > 0 and 2 <= self.c_opers.shape[-1] <= numeric._lib.MAX_D_TEMPLATED
                and self.d == self.c_opers.shape[-1]      # (a user-overridden d: the array route)
                and not any(key in self._data for key in _DIAGONALIZATION)
                and not any(key in self._frequency_data
                            for key in ('control_matrix', 'control_matrix_pc')))

    def _resident_pass(self, keep_filter_function=True, spectrum=None, idx=None):
        """diagonalize + control matrix + filter function in one library call; the control matrix
        stays on the device behind a :class:`Deferred` cache entry.  (``cache_control_matrix`` caches
        no filter function in the reference: *keep_filter_function* False leaves it out.)  With a
        validated *spectrum* and *idx* the infidelity integral rides in the same call and is
        returned."""
        result = ResidentResult()
        integral = {} if spectrum is None else dict(spectrum=spectrum, idx=idx, d_infidelity=self.d)
        out = result.evaluate(self.c_opers, self.dt, self.t, self.omega, np.asarray(self.basis),
                              self.n_opers, self.n_coeffs, c_coeffs=self.c_coeffs, **integral)
        D, V, Q, F = out[:4]
        self._data.update(eigvals=D, eigvecs=V, propagators=Q, total_propagator=Q[-1])
        self._frequency_data['control_matrix'] = Deferred(result.control_matrix,
                                                          result.control_matrix_nbytes())
        if keep_filter_function:
            self._frequency_data['filter_function'] = F
        self._defer_by_products()
        self._resident = result
        return out[4] if spectrum is not None else None

    def nothing_cached_for(self, omega, cache_intermediates=False):
        """Sets *omega* and says whether ``ff.infidelity`` can run the whole path and the integral in
        one library call (``_resident_pass(spectrum=..., idx=...)``): a pulse with nothing to reuse."""
        self.omega = omega
        return ('filter_function' not in self._frequency_data and len(self.omega) >= 2
                and self._resident_pass_applies('fidelity', 1, cache_intermediates))

    def _defer_by_products(self):
        """What the reference's cache_control_matrix computes on the spot -- total phase factors and
        the Liouville representation of the total propagator -- becomes due on first read."""
        # The producers close over the VALUES they need, as the reference's eager evaluation would
        # have seen them -- never over the pulse: a closure over `self` stored in the pulse's own
        # cache would be a reference cycle (the pulse, with the device and pinned blocks of its
        # resident result, would live until the next pass of the cyclic GC), and a weak reference
        # would dangle in a shallow copy that outlives the original.
        if 'total_phases' not in self._frequency_data:
            omega, tau = self.omega, self.tau
            self._frequency_data['total_phases'] = Deferred(
                lambda: util.cexp(np.asarray(omega)*tau), 16*len(omega))
        if 'total_propagator_liouville' not in self._data:
            basis = self.basis
            known = 'total_propagator' in self._data
            total = self._data['total_propagator'] if known else None
            if known:
                produce = lambda: liouville_representation(total, basis)     # noqa: E731
            else:
                # not diagonalised yet (a control matrix handed to cache_control_matrix): the
                # diagonalisation itself is deferred with it, through a weak reference that
                # __copy__ re-points at the twin
                me = weakref.ref(self)
                produce = lambda: liouville_representation(me().total_propagator, basis)  # noqa: E731
            self._data['total_propagator_liouville'] = Deferred(produce, 8*len(basis)**2)

    def _store_control_matrix(self, control_matrix):
        slot = 'control_matrix_pc' if control_matrix.ndim == 4 else 'control_matrix'
        self._frequency_data[slot] = control_matrix
        self._resident = None                 # (a resident result would describe another control matrix)
        self._defer_by_products()

    def get_control_matrix(self, omega, show_progressbar=False, cache_intermediates=False):
        """Control matrix (n_nops, d**2, n_omega) for *omega*, memoised
        (reference pulse_sequence.py:588-636)."""
        self.omega = omega
        known = self._frequency_data
        if 'control_matrix' not in known and 'control_matrix_pc' in known:
            known['control_matrix'] = np.sum(known['control_matrix_pc'], axis=0)
        if 'control_matrix' not in known:
            self.diagonalize()
            result = numeric.calculate_control_matrix_from_scratch(
                self.eigvals, self.eigvecs, self.propagators, self.omega, self.basis, self.n_opers,
                self.n_coeffs, self.dt, self.t, show_progressbar=show_progressbar,
                cache_intermediates=cache_intermediates)
            if cache_intermediates:
                result, by_products = result
                self._intermediates.update(by_products)
            self._store_control_matrix(result)
        return known['control_matrix']

    def cache_control_matrix(self, omega, control_matrix=None, show_progressbar=False,
                             cache_intermediates=False):
        """Cache the control matrix -- computed now if not given --, the total phases and the
        total Liouville propagator (reference pulse_sequence.py:638-677)."""
        self.omega = omega
        if control_matrix is not None:
            self._store_control_matrix(control_matrix)
        elif self._resident_pass_applies('fidelity', 1, cache_intermediates):
            self._resident_pass(keep_filter_function=False)   # stays in HBM until somebody reads it
        else:
            self.get_control_matrix(self.omega, show_progressbar, cache_intermediates)

    def get_pulse_correlation_control_matrix(self):
        if 'control_matrix_pc' not in self._frequency_data:
            raise util.CalculationError(_NOT_CONCATENATED.format('control matrix'))
        return self._frequency_data['control_matrix_pc']

    @util.parse_optional_parameters(which=('fidelity', 'generalized'), order=(1, 2))
    def get_filter_function(self, omega, which='fidelity', order=1, show_progressbar=False,
                            cache_intermediates=False, cache_second_order_cumulative=False, writable=None):
        """Filter function, memoised (reference pulse_sequence.py:691-805).  order=1: 'fidelity' ->
        (n_nops, n_nops, n_omega), 'generalized' -> (n_nops, n_nops, d², d², n_omega); order=2: the
        second-order filter function (n_nops, n_nops, d², d², n_omega), *which* ignored.

        ``writable`` (not in the reference; default :data:`WRITABLE_RESULTS`, i.e. the environment variable
        ``FFK_WRITABLE_RESULTS``, else False): the resident pass hands out a READ-ONLY view of pinned memory whose
        device copy ``ff.infidelity`` integrates in place.  With ``writable=True`` the cached entry becomes an
        ordinary owned array, as the reference returns it (pulse_sequence.py:772-783): edits are allowed and
        ``ff.infidelity`` integrates the array as the caller left it (one upload of F)."""
        self.omega = omega
        if writable is None:
            writable = WRITABLE_RESULTS
        wanted = ('filter_function_2' if order == 2 else
                  'filter_function' if which == 'fidelity' else 'filter_function_gen')
        if wanted not in self._frequency_data:
            if self._resident_pass_applies(which, order, cache_intermediates):
                self._resident_pass()
            else:
                self.cache_filter_function(
                    self.omega, which=which, order=order, show_progressbar=show_progressbar,
                    cache_intermediates=cache_intermediates,
                    cache_second_order_cumulative=cache_second_order_cumulative)
        result = self._frequency_data[wanted]
        if writable and not result.flags.writeable:
            # an owned copy takes the view's place in the cache (memoised by reference from now on); the resident F is
            # no longer "this array", so infidelity() takes the array route
            result = self._frequency_data[wanted] = np.array(result)
        return result

    @util.parse_optional_parameters(which=('fidelity', 'generalized'), order=(1, 2))
    def cache_filter_function(self, omega, control_matrix=None, filter_function=None,
                              which='fidelity', order=1, show_progressbar=False,
                              cache_intermediates=False, cache_second_order_cumulative=False):
        """Cache the filter function -- given, or computed from the (given or cached or computed)
        control matrix (reference pulse_sequence.py:807-902)."""
        self.omega = omega
        known = self._frequency_data
        if order == 2:
            if filter_function is None:
                filter_function = numeric.calculate_second_order_filter_function_from_scratch(
                    self.eigvals, self.eigvecs, self.propagators, self.omega, self.basis,
                    self.n_opers, self.n_coeffs, self.dt, self._intermediates, show_progressbar,
                    cache_intermediates, cache_second_order_cumulative)
                if cache_intermediates:
                    filter_function, by_products = filter_function
                    self._intermediates.update(by_products)
            known['filter_function_2'] = filter_function
            return
        generalized = which == 'generalized'
        if (filter_function is None and control_matrix is None
                and self._resident_pass_applies(which, order, cache_intermediates)):
            self._resident_pass()
            return
        if filter_function is None:
            if control_matrix is None:
                control_matrix = self.get_control_matrix(self.omega, show_progressbar,
                                                         cache_intermediates)
            else:
                self._store_control_matrix(control_matrix)
            if control_matrix.ndim == 3:
                filter_function = numeric.calculate_filter_function(control_matrix, which)
            else:
                # one control matrix per pulse of a concatenation: keep the pulse correlations too
                correlations = numeric.calculate_pulse_correlation_filter_function(control_matrix,
                                                                                   which)
                if generalized:
                    known['filter_function_pc_gen'] = correlations
                    known['filter_function_pc'] = correlations.trace(axis1=4, axis2=5)
                else:
                    known['filter_function_pc'] = correlations
                filter_function = correlations.sum(axis=(0, 1))
        if generalized:
            known['filter_function_gen'] = filter_function
            filter_function = filter_function.trace(axis1=2, axis2=3)
        known['filter_function'] = filter_function
        if self._resident is not None and self._resident.filter_function is not filter_function:
            self._resident = None

    def get_filter_function_derivative(self, omega, control_identifiers=None,
                                       n_oper_identifiers=None, n_coeffs_deriv=None):
        """Derivative of the filter function by the control amplitudes, shape (n_nops, n_dt,
        n_ctrl, n_omega) (reference pulse_sequence.py:977-1054); see
        :func:`filter_functions_amd.gradient.filter_function_derivative`."""
        from . import gradient
        return gradient.filter_function_derivative(self, omega, control_identifiers,
                                                   n_oper_identifiers, n_coeffs_deriv)

    @util.parse_optional_parameters(which=('fidelity', 'generalized'))
    def get_pulse_correlation_filter_function(self, which='fidelity'):
        known = self._frequency_data
        wanted = 'filter_function_pc' if which == 'fidelity' else 'filter_function_pc_gen'
        if wanted not in known:
            if 'control_matrix_pc' not in known:
                raise util.CalculationError(_NOT_CONCATENATED.format('filter function'))
            known[wanted] = numeric.calculate_pulse_correlation_filter_function(
                known['control_matrix_pc'], which=which)
        return known[wanted]

    def get_total_phases(self, omega):
        """exp(i omega tau), memoised (reference pulse_sequence.py:1056-1066)."""
        self.omega = omega
        if 'total_phases' not in self._frequency_data:
            self._frequency_data['total_phases'] = util.cexp(np.asarray(self.omega)*self.tau)
        return self._frequency_data['total_phases']

    def cache_total_phases(self, omega, total_phases=None):
        self.omega = omega
        if total_phases is None:
            self.get_total_phases(self.omega)
        else:
            self._frequency_data['total_phases'] = total_phases

    def resident_infidelity(self, filter_function, spectrum, idx):
        """Integrate *spectrum* against *filter_function* on the device if that array is the one a
        resident pass left in HBM (no upload of F); ``None`` if it is not."""
        resident = self._resident
        if resident is None or resident.filter_function is not filter_function:
            return None
        return resident.infidelity(spectrum, idx, self.d)

    # ---- cached-data properties ----------------------------------------------------------
    def _diagonalization_product(name):  # noqa: N805  (evaluated while the class body runs)
        def read(self):
            if name not in self._data:
                self.diagonalize()
            return self._data[name]

        def write(self, value):
            self._data[name] = value
        return property(read, write)

    eigvals = _diagonalization_product('eigvals')
    eigvecs = _diagonalization_product('eigvecs')
    propagators = _diagonalization_product('propagators')
    total_propagator = _diagonalization_product('total_propagator')
    del _diagonalization_product

    @property
    def total_propagator_liouville(self):
        if 'total_propagator_liouville' not in self._data:
            self._data['total_propagator_liouville'] = liouville_representation(
                self.total_propagator, self.basis)
        return self._data['total_propagator_liouville']

    @total_propagator_liouville.setter
    def total_propagator_liouville(self, value):
        self._data['total_propagator_liouville'] = value

    @property
    def omega(self):
        return self._frequency_data.get('omega')

    @omega.setter
    def omega(self, value):
        """Remember (a copy of) the frequencies; a grid that differs from the remembered one
        invalidates everything that depends on frequency (reference pulse_sequence.py:1158-1169)."""
        known = self._frequency_data.get('omega')
        if known is not None and _same_grid(known, value):
            return                           # the remembered copy is the same grid: keep it
        if _OWNED_GRIDS.get(id(value)) is value:
            # a grid this package copied and froze itself -- another pulse's remembered copy, handed
            # on by concatenate / remap / extend -- is shared, not copied again: the pulses of a long
            # sequence then hold ONE grid object and comparing their grids is an identity test.  (A
            # caller's own read-only array is NOT trusted: a read-only view of a writable base can
            # change under the pulse; the reference always copies, pulse_sequence.py:1166.)
            grid = value
        else:
            grid = _interned_grid(value)
        self.cleanup('frequency dependent')
        self._frequency_data['omega'] = grid

    @property
    def nbytes(self):
        """Bytes held by the caches (device-resident entries count with their host size)."""
        return sum(cache.stored_nbytes()
                   for cache in (self._data, self._frequency_data, self._intermediates))

    @util.parse_optional_parameters(method=('conservative', 'greedy', 'frequency dependent', 'all'))
    def cleanup(self, method='conservative'):
        """Drop cached by-products (reference pulse_sequence.py:1188-1245): 'conservative' the
        diagonalisation; 'greedy' also the total propagators, the control matrices, the total
        phases and the intermediates (the filter functions stay); 'frequency dependent' all that
        belongs to a frequency grid; 'all' everything.  Every method also drops what concatenations remember
        about this pulse (:func:`clear_merged_tables`)."""
        clear_merged_tables(self)
        if method in ('all', 'frequency dependent'):
            if method == 'all':
                self._data.clear()
            self._frequency_data.clear()
            self._intermediates.clear()
            self._resident = None
            return
        from_data, from_frequency_data, intermediates_too = _CLEANUP[method]
        for key in from_data:
            self._data.pop(key, None)
        for key in from_frequency_data:
            self._frequency_data.pop(key, None)
        if intermediates_too:
            self._intermediates.clear()


# --------------------------------------------------------------------------------------------
# Concatenation (reference pulse_sequence.py:1340-1483, 1599-1887).  Host-side bookkeeping on
# identifiers and coefficient tables; the arithmetic -- atomic control matrices, Liouville
# propagators, the concatenation rule and the filter functions -- runs in libffk.
# --------------------------------------------------------------------------------------------
_GRIDS = weakref.WeakValueDictionary()
_OWNED_GRIDS = weakref.WeakValueDictionary()     # id(grid) -> grid, for every grid _interned_grid froze


def _interned_grid(value):
    """The remembered copy of a frequency grid handed in by the user: read-only, and ONE object per
    distinct content for as long as some pulse holds it (pulses built separately on the same grid -- the
    X/2 and Y/2 atoms of a gate set -- then share it, and every later comparison is an identity test)."""
    grid = np.array(value, dtype=None, copy=True)
    if grid.dtype != np.float64 or grid.ndim != 1:
        return grid
    key = (grid.shape[0], hash(grid.tobytes()))
    known = _GRIDS.get(key)
    if known is not None and np.array_equal(known, grid):
        return known
    grid.flags.writeable = False
    _GRIDS[key] = grid
    _OWNED_GRIDS[id(grid)] = grid
    return grid


def _same_grid(a, b):
    """Two frequency grids are the same object or hold the same values."""
    return a is b or np.array_equal(a, b)


def _merge_constant_segments(pulse):
    """(c_coeffs, n_coeffs, dt) with runs of segments whose control amplitudes do not change
    merged into one (their durations added), so that equality does not depend on how a constant
    stretch was split (reference pulse_sequence.py:1270-1285)."""
    same = (np.diff(pulse.c_coeffs) == 0).all(axis=0).nonzero()[0]
    if same.size == 0:
        return pulse.c_coeffs, pulse.n_coeffs, pulse.dt
    dt = np.delete(pulse.dt, same)
    for old, new in zip(same, same - np.arange(len(same))):
        dt[new] += pulse.dt[old]
    return np.delete(pulse.c_coeffs, same, axis=1), np.delete(pulse.n_coeffs, same, axis=1), dt


def _running_products(matrices):
    """P[g] = M[g] M[g-1] ... M[0] for a stack (G, d, d), as a log-depth (Hillis-Steele) scan:
    ceil(log2 G) batched matrix products instead of G - 1 dependent ones."""
    P = np.array(matrices)
    shift = 1
    while shift < len(P):
        P[shift:] = P[shift:] @ P[:-shift]
        shift *= 2
    return P


def _all_bases_equal(pulses):
    first = pulses[0].basis
    # the same Basis object (or one already compared) needs no element-wise comparison: a long
    # sequence is typically built from a handful of distinct pulse objects
    others = {id(p.basis): p.basis for p in pulses[1:]}
    others.pop(id(first), None)
    if not others:
        return True
    if any(b.shape != first.shape for b in others.values()):
        return False
    # the remaining distinct objects in ONE comparison (24 small arrays: one call instead of 23)
    return bool((np.array([np.asarray(b) for b in others.values()]) == np.asarray(first)).all())


def _distinct_objects(objects):
    """Positions -> distinct objects (by identity, in order of first appearance): returns (distinct,
    first position of each, index) with objects[p] is distinct[index[p]]."""
    # (no Python-level loop over the positions: a 1000-gate sequence is the common case)
    ids = np.fromiter(map(id, objects), dtype=np.int64, count=len(objects))
    _, first, inverse = np.unique(ids, return_index=True, return_inverse=True)
    order = np.argsort(first, kind='stable')                      # distinct objects by first appearance
    rank = np.empty(len(order), dtype=np.int32)
    rank[order] = np.arange(len(order), dtype=np.int32)
    first = first[order].astype(np.intp)
    return [objects[p] for p in first], first, rank[inverse.reshape(-1)]


def _ragged_columns(lengths, index):
    """Column numbers that gather, from the blocks of the distinct entries laid side by side
    (entry k occupying ``lengths[k]`` columns), the blocks of the entries ``index`` in sequence."""
    lengths = np.asarray(lengths, dtype=np.intp)
    offsets = np.cumsum(lengths) - lengths
    lens = lengths[index]
    starts = np.cumsum(lens) - lens
    return np.repeat(offsets[index] - starts, lens) + np.arange(int(lens.sum()), dtype=np.intp)


def _same_operator_tables(opers, identifiers):
    """Do all entries hold identical operator arrays under identical identifiers, in the same order?"""
    n = len(opers[0])
    if any(len(o) != n for o in opers) or any(len(i) != n for i in identifiers):
        return False
    try:
        ids = np.array([np.asarray(i) for i in identifiers])
        ops = np.array([np.asarray(o) for o in opers])
    except ValueError:                       # ragged: operators of different dimension
        return False
    if ids.ndim != 2 or ops.ndim != 4:
        return False
    return bool((ids == ids[0]).all()) and bool((ops == ops[0]).all())


def _concatenate_hamiltonian(opers, identifiers, coeffs, kind, first=None, index=None, columns=None):
    """Merge the operator tables of several pulses (contract of reference
    pulse_sequence.py:1340-1483).

    Operators with identical matrices are one operator of the new pulse (they must then carry
    the same identifier in every pulse, else ValueError); an identifier that names different
    matrices in different pulses is disambiguated by appending the pulse position.  Returns the
    operators sorted by their new identifiers, the identifiers, the (n_opers, sum n_dt)
    coefficient table -- zero-filled for control terms a pulse lacks, filled with the common
    constant for noise sensitivities (ValueError if it is not constant) -- and, per pulse
    position, the map old identifier -> new identifier.

    A long sequence is typically drawn from a handful of pulse objects (1000 gates from 24
    Cliffords): the three lists hold one entry per DISTINCT pulse, ``first[k]`` is the sequence
    position where entry k first appears and ``index[p]`` the entry at position p (default: every
    entry once, in order).  The work splits into what depends on the distinct pulses only
    (:func:`_merge_hamiltonian`: names, order, error checks, the entries' coefficient blocks side by
    side) and one gather per sequence (:func:`_gather_hamiltonian`).
    """
    if index is None:
        first, index = np.arange(len(opers)), np.arange(len(opers))
    return _gather_hamiltonian(_merge_hamiltonian(opers, identifiers, coeffs, kind, first), index, columns)


def _merge_hamiltonian(opers, identifiers, coeffs, kind, first):
    """The part of :func:`_concatenate_hamiltonian` that does not depend on the order of the sequence
    (``merged['uses_first']``: unless an identifier had to be disambiguated by a position)."""
    n_entries = len(opers)
    lengths = np.array([np.shape(c)[1] for c in coeffs], dtype=np.intp)
    equal_lengths = bool((lengths == lengths[0]).all())
    if n_entries > 1 and _same_operator_tables(opers, identifiers):
        # Every pulse carries the same operators under the same names in the same order (a gate
        # sequence over one register): the bookkeeping of ONE entry holds for all of them --
        # names, order and error checks as below -- and the coefficient table is the entries'
        # tables side by side, rows permuted, blocks gathered.
        one = _merge_hamiltonian(opers[:1], identifiers[:1], coeffs[:1], kind, first[:1])
        one_map = one['maps'][0]
        names = [str(ident) for ident in identifiers[0]]
        perm = [names.index(old) for new in one['identifiers'] for old, mapped in one_map.items() if mapped == new]
        blocks = (np.array(coeffs, dtype=float)[:, perm] if equal_lengths
                  else np.concatenate(coeffs, axis=1)[perm])
        return dict(same=True, opers=one['opers'], identifiers=one['identifiers'], maps=[one_map]*n_entries,
                    blocks=blocks, lengths=lengths, equal_lengths=equal_lengths, kind=kind,
                    uses_first=one['uses_first'])
    # one record per (entry, operator): (entry, row in the pulse, matrix, name)
    records = [(k, i, np.ascontiguousarray(op).tobytes(), str(ident))
               for k in range(len(opers))
               for i, (op, ident) in enumerate(zip(opers[k], identifiers[k]))]
    idents_of_matrix, matrices_of_ident, first_seen = {}, {}, {}
    for k, i, key, ident in records:
        idents_of_matrix.setdefault(key, set()).add(ident)
        matrices_of_ident.setdefault(ident, set()).add(key)
        first_seen.setdefault(key, (k, i, ident))
    if any(len(v) > 1 for v in idents_of_matrix.values()):
        raise ValueError(f'Trying to concatenate pulses with equal {kind} operators but '
                         f'different identifiers. Please choose unique {kind} identifiers!')
    # new identifier of every distinct matrix
    uses_first = any(len(v) > 1 for v in matrices_of_ident.values())
    new_ident = {key: (f'{ident}_{first[k]}' if len(matrices_of_ident[ident]) > 1 else ident)
                 for key, (k, i, ident) in first_seen.items()}
    ordered = sorted(first_seen, key=new_ident.get)
    row = {key: r for r, key in enumerate(ordered)}
    concat_identifiers = np.array([new_ident[key] for key in ordered])
    concat_opers = np.array([np.asarray(opers[first_seen[key][0]][first_seen[key][1]])
                             for key in ordered])
    # per entry: its block of the coefficient table (NaN where it lacks an operator) -- all blocks
    # side by side in ONE array -- and its identifier map; per pulse position: a reference to those
    offsets = np.concatenate(([0], np.cumsum(lengths)))
    # (a control term a pulse lacks is zero; a noise sensitivity it lacks is to be inferred: NaN for now)
    side_by_side = np.full((len(ordered), int(offsets[-1])), np.nan if kind == 'noise' else 0.0)
    maps = [{} for _ in coeffs]
    carried = np.zeros((len(ordered), len(coeffs)), dtype=bool)
    for k, i, key, ident in records:
        maps[k][ident] = new_ident[key]
    if records:
        # all blocks in ONE scatter: the records are in (entry, row) order, i.e. the order of the
        # entries' coefficient arrays read row by row
        rec_k = np.array([rec[0] for rec in records], dtype=np.intp)
        rec_row = np.array([row[rec[2]] for rec in records], dtype=np.intp)
        rec_len = lengths[rec_k]
        starts = np.cumsum(rec_len) - rec_len
        values = np.concatenate([np.asarray(c, dtype=float).reshape(-1) for c in coeffs])
        dest = np.repeat(rec_row*int(offsets[-1]) + offsets[rec_k] - starts, rec_len) + np.arange(len(values))
        side_by_side.reshape(-1)[dest] = values
        carried[rec_row, rec_k] = True
    return dict(same=False, opers=concat_opers, identifiers=concat_identifiers, maps=maps,
                blocks=side_by_side, lengths=lengths, offsets=offsets, equal_lengths=equal_lengths,
                incomplete_rows=np.nonzero(~carried.all(axis=1))[0], kind=kind, uses_first=uses_first)


def _gather_hamiltonian(merged, index, columns=None):
    """The coefficient table of the sequence ``index`` from the merged tables of its distinct pulses."""
    lengths, blocks, n_entries = merged['lengths'], merged['blocks'], len(merged['maps'])
    mapping = _PositionMap(merged['maps'], index)
    if merged['same']:
        if merged['equal_lengths']:
            table = blocks[index].transpose(1, 0, 2).reshape(blocks.shape[1], -1)
        else:
            table = np.take(blocks, _ragged_columns(lengths, index) if columns is None else columns, axis=1)
        return merged['opers'].copy(), merged['identifiers'].copy(), table, mapping
    n_rows = len(merged['identifiers'])
    if merged['equal_lengths']:
        # equal segment counts: one gather (operators, entries, segments)[:, index] -> (operators, all segments)
        table = blocks.reshape(n_rows, n_entries, -1)[:, index].reshape(n_rows, -1)
    elif len(index) <= 4*n_entries:
        # few positions (possibly long pulses): plain block copies
        offsets = merged['offsets']
        table = np.concatenate([blocks[:, offsets[k]:offsets[k + 1]] for k in index], axis=1)
    else:
        # many positions drawn from few pulses, ragged: one gather of columns (no Python-level loop
        # over the positions)
        # (np.take along the axis: 5x faster than the equivalent fancy index on a few long rows)
        table = np.take(blocks, _ragged_columns(lengths, index) if columns is None else columns, axis=1)
    if merged['kind'] == 'noise':
        for r in merged['incomplete_rows']:                    # rows that some pulse does not carry
            missing = np.isnan(table[r])
            known = table[r][~missing]
            if not (known == known[0]).all():
                raise ValueError('Not all pulses have the same noise operators and '
                                 'non-trivial noise sensitivities so I cannot infer them.')
            table[r, missing] = known[0]
    # (the merged tables may be remembered and serve further sequences: hand out copies)
    return merged['opers'].copy(), merged['identifiers'].copy(), table, mapping


class _PositionMap(Mapping):
    """position -> identifier map of the pulse standing there, without materialising one dict
    entry per position: a read-only ``Mapping`` (``mapping[p]``, ``len``, iteration over the
    positions, ``keys/values/items/get``, ``==`` against a dict) over the maps of the DISTINCT pulses."""

    def __init__(self, maps, index):
        self._maps, self._index = maps, index

    def __getitem__(self, position):
        if not isinstance(position, (int, np.integer)) or not -len(self._index) <= position < len(self._index):
            raise KeyError(position)
        return self._maps[self._index[position]]

    def __len__(self):
        return len(self._index)

    def __iter__(self):
        return iter(range(len(self._index)))

    def __repr__(self):
        return repr(dict(self.items()))


def _validated_sequence(pulses):
    """The pulses as a tuple, their distinct objects, first positions and position -> object index
    (TypeError / ValueError of reference pulse_sequence.py:1626-1640)."""
    try:
        pulses = tuple(pulses)           # any iterable, also a generator: consumed exactly once
    except TypeError:
        raise TypeError(f'Expected pulses to be iterable, not {type(pulses)}') from None
    distinct, first, index = _distinct_objects(pulses)
    for pulse in distinct:
        if not isinstance(pulse, PulseSequence):
            raise TypeError('Can only concatenate PulseSequences!')
    return pulses, distinct, first, index


#: Merged tables of sets of distinct pulses (see :func:`_merged_tables`): few entries, oldest dropped first.  A
#: process-wide cache of COPIES keyed on the pulse objects; an entry is dropped when one of its pulses dies, is
#: ``cleanup()``-ed (any method), or no longer holds the arrays -- object AND content -- it was built from.
#: :func:`clear_merged_tables` empties it.
_MERGED = OrderedDict()
_MERGED_MAX = 8


def clear_merged_tables(pulse=None):
    """Forget the remembered merged tables of concatenations: all of them, or those that involve ``pulse``
    (``PulseSequence.cleanup`` calls this with itself)."""
    if pulse is None:
        _MERGED.clear()
        return
    for key in [k for k, hit in _MERGED.items() if any(r() is pulse for r in hit['refs'])]:
        del _MERGED[key]


def _content_stamp(pulse):
    """A checksum of the arrays a concatenation reads from ``pulse``: a pulse whose coefficients, time steps or
    operators were modified IN PLACE no longer matches its remembered tables (the reference rebuilds from the live
    arrays on every call, pulse_sequence.py:1599-1665).  ~0.3 us per small array."""
    crc = 0
    for a in (pulse.c_coeffs, pulse.n_coeffs, pulse.dt, pulse.c_opers, pulse.n_opers):
        a = np.asarray(a)
        crc = zlib.crc32(a if a.flags.c_contiguous else np.ascontiguousarray(a), crc)
    return crc


def _merged_tables(distinct, first):
    """What a concatenation needs of its DISTINCT pulses and not of their order: dimension and basis checks,
    the merged control and noise tables, segment counts, durations.  Randomized benchmarking evaluates many
    sequences drawn from one gate set: the result is remembered per set of pulse OBJECTS and reused while they
    are alive and still hold the same arrays with the same content (identity of every attribute the tables are
    built from -- the entry keeps those objects alive, so an ``id`` cannot be recycled -- plus a checksum of the
    numeric ones); 0.22 -> 0.09 ms of the 0.49 ms a 1000-gate sequence takes (profiles/r05_h_*)."""
    for key in [k for k, hit in _MERGED.items() if any(r() is None for r in hit['refs'])]:
        del _MERGED[key]                  # a pulse of the set has died
    key = tuple(map(id, distinct))
    held = tuple(a for p in distinct
                 for a in (p.c_opers, p.c_coeffs, p.n_opers, p.n_coeffs, p.dt, p.basis, p.c_oper_identifiers,
                           p.n_oper_identifiers))
    stamp = tuple(map(id, held)) + tuple(_content_stamp(p) for p in distinct)
    hit = _MERGED.get(key)
    if hit is not None and hit['stamp'] == stamp and all(r() is p for r, p in zip(hit['refs'], distinct)):
        _MERGED.move_to_end(key)
        return hit
    if any(pulse.d != distinct[0].d for pulse in distinct):
        raise ValueError('Trying to concatenate PulseSequence instances with different dimension!')
    if not _all_bases_equal(distinct):
        raise ValueError('Trying to concatenate PulseSequence instances with different bases!')
    lengths = np.array([len(p.dt) for p in distinct])
    equal = bool((lengths == lengths[0]).all())
    merged = dict(
        stamp=stamp, held=held, refs=[weakref.ref(p) for p in distinct], lengths=lengths, equal_lengths=equal,
        control=_merge_hamiltonian([p.c_opers for p in distinct], [p.c_oper_identifiers for p in distinct],
                                   [p.c_coeffs for p in distinct], 'control', first),
        noise=_merge_hamiltonian([p.n_opers for p in distinct], [p.n_oper_identifiers for p in distinct],
                                 [p.n_coeffs for p in distinct], 'noise', first),
        dt=np.stack([p.dt for p in distinct]) if equal else np.concatenate([p.dt for p in distinct]),
        tau=np.array([p.tau for p in distinct], dtype=float))
    if not (merged['control']['uses_first'] or merged['noise']['uses_first']):
        _MERGED[key] = merged
        while len(_MERGED) > _MERGED_MAX:
            _MERGED.popitem(last=False)
    return merged


def _concatenate_distinct(pulses, distinct, first, index):
    """concatenate_without_filter_function on an already validated sequence."""
    merged = _merged_tables(distinct, first)
    lengths = merged['lengths']
    # (ragged pulses, many positions: the three gathers below share their column numbers)
    ragged = not merged['equal_lengths'] and len(index) > 4*len(distinct)
    columns = _ragged_columns(lengths, index) if ragged else None
    c_opers, c_ids, c_coeffs, c_map = _gather_hamiltonian(merged['control'], index, columns)
    n_opers, n_ids, n_coeffs, n_map = _gather_hamiltonian(merged['noise'], index, columns)
    if merged['equal_lengths']:
        dt = merged['dt'][index].reshape(-1)
    elif not ragged:
        dt = np.concatenate([distinct[k].dt for k in index])
    else:
        dt = merged['dt'].take(columns)
    newpulse = PulseSequence.from_arrays(c_opers, c_ids, c_coeffs, n_opers, n_ids, n_coeffs, dt,
                                         distinct[0].basis)
    # (summed position by position like the reference's sum over the pulses: a running sum, not
    # NumPy's pairwise reduction)
    newpulse.tau = float(np.cumsum(merged['tau'][index])[-1])
    return newpulse, c_map, n_map


def concatenate_without_filter_function(pulses, return_identifier_mappings=False):
    """Concatenate pulses without touching any filter function
    (reference pulse_sequence.py:1534-1665)."""
    newpulse, c_map, n_map = _concatenate_distinct(*_validated_sequence(pulses))
    if return_identifier_mappings:
        return newpulse, c_map, n_map
    return newpulse


@util.parse_optional_parameters(which=('fidelity', 'generalized'))
def concatenate(pulses, calc_pulse_correlation_FF=False, calc_filter_function=None,
                which='fidelity', omega=None, show_progressbar=False, calc_second_order_FF=False):
    r"""Concatenate pulses and, where it pays, their filter functions by the concatenation rule
    :math:`\tilde{\mathcal B}(\omega)=\sum_g e^{i\omega t_{g-1}}\tilde{\mathcal B}^{(g)}(\omega)
    \mathcal Q^{(g-1)}` (reference pulse_sequence.py:1668-1887).

    Same decision logic as the reference: the filter function is computed if
    ``calc_filter_function`` is True, or left out if False, or -- by default -- computed only if
    at least one pulse has a cached control matrix, all cached frequencies agree and at least
    two pulses share a noise operator.  ``calc_pulse_correlation_FF`` keeps every summand and
    caches the pulse correlation filter function.
    """
    pulses, distinct, first_position, index = _validated_sequence(pulses)
    if len(pulses) == 1:
        return copy.deepcopy(pulses[0])  # nothing to concatenate: an independent copy, caches kept
    newpulse, _, n_map = _concatenate_distinct(pulses, distinct, first_position, index)

    cumulative = []

    def cumulative_propagators():
        """U_g ... U_2 U_1 for every g, once (log-depth scan over the sequence, on the host: the
        routes that do not end in the fused device call)."""
        if not cumulative:
            cumulative.append(_running_products(
                np.array([pls.total_propagator for pls in distinct])[index]))
        return cumulative[0]

    def finished(pulse):
        """The new pulse with its total propagator, if every pulse knows its own (contract of the
        reference, pulse_sequence.py:1753-1754)."""
        if 'total_propagator' not in pulse._data and all('total_propagator' in pls._data
                                                         for pls in distinct):
            pulse.total_propagator = cumulative_propagators()[-1]
        return pulse
    if calc_pulse_correlation_FF or calc_second_order_FF is True:
        calc_filter_function = True         # both need every summand of the control matrix
    elif calc_filter_function is False:
        return finished(newpulse)

    # distinct pulse objects (a randomized-benchmarking sequence draws 1000 gates from 24
    # Cliffords): every per-pulse question below is asked once per object
    first_position = [int(p) for p in first_position]

    # which noise operators of the new pulse does each pulse carry?
    new_ids = list(newpulse.n_oper_identifiers)
    column = {ident: c for c, ident in enumerate(new_ids)}
    carried = np.zeros((len(distinct), len(new_ids)), dtype=bool)
    # (one scatter; pulses that share their identifier map -- all of them when the sequence carries
    # one operator table -- share their columns)
    columns_of, rows_k, cols_k = {}, [], []
    for k, p in enumerate(first_position):
        this_map = n_map[p]
        cols = columns_of.get(id(this_map))
        if cols is None:
            cols = columns_of[id(this_map)] = [column[ident] for ident in this_map.values()]
        rows_k += [k]*len(cols)
        cols_k += cols
    carried[rows_k, cols_k] = True
    if carried.all():
        present = np.broadcast_to(True, (len(index), len(new_ids)))
        shared_n_opers = len(index) > 1
    else:
        present = carried[index]
        shared_n_opers = bool((present.sum(axis=0) > 1).any())
    if calc_second_order_FF and not present.all():
        warn('Second order FF requested but not all pulses have the same n_opers. '
             'Not implemented.', UserWarning)
        calc_second_order_FF = False

    if omega is None:
        cached_R = [pls.is_cached('control_matrix') for pls in distinct]
        cached_w = [pls.is_cached('omega') for pls in distinct]
        candidates = [pls.omega for pls, c in zip(distinct, cached_R if any(cached_R) else cached_w)
                      if c]
        equal_omega = all(_same_grid(candidates[0], w) for w in candidates[1:])
        if not equal_omega or not candidates:
            if calc_filter_function:
                raise ValueError('Calculation of filter function forced but not all pulses '
                                 'have the same frequencies cached and none were supplied!')
            if calc_pulse_correlation_FF:
                raise ValueError('Cannot compute the pulse correlation filter functions; do not '
                                 'have the frequencies at which to evaluate.')
            return finished(newpulse)
        if calc_filter_function is None and (not shared_n_opers or not any(cached_R)):
            return finished(newpulse)
        omega = candidates[0]

    if not shared_n_opers:
        # nothing to reuse: plain from-scratch evaluation of the long sequence
        finished(newpulse)
        newpulse.cache_filter_function(omega, which=which)
        if calc_second_order_FF:
            newpulse.cache_filter_function(omega, order=2)
        return newpulse

    # every distinct control matrix is evaluated / fetched once
    seg = np.concatenate(([0], np.cumsum(np.array([len(pls.dt) for pls in distinct])[index])))

    own_rows_of = {}

    def own_rows(i):
        """Rows of pulse i's control matrix in the order of the new pulse's noise operators."""
        pls = pulses[i]
        # (pulses that share identifier map, identifiers and presence give the same answer)
        key = (id(n_map[i]), tuple(pls.n_oper_identifiers), present[i].tobytes())
        rows = own_rows_of.get(key)
        if rows is None:
            rows = own_rows_of[key] = [list(pls.n_oper_identifiers).index(old)
                                       for new in np.asarray(new_ids)[present[i]]
                                       for old, mapped in n_map[i].items() if mapped == new]
        return rows

    def atomic_control_matrix(i):
        """Control matrix of the pulse at position i in the new pulse's operator order."""
        pls, here = pulses[i], present[i]
        own_order = own_rows(i)
        own = pls.get_control_matrix(omega, show_progressbar)
        if here.all() and own_order == list(range(len(own))):
            return own                       # same operators in the same order: no copy
        R = np.empty((len(newpulse.n_opers), len(newpulse.basis), len(omega)), dtype=complex)
        R[here] = own[own_order]
        if not here.all():
            # noise operators this pulse does not know: evaluate them on its control Hamiltonian
            R[~here] = numeric.calculate_control_matrix_from_scratch(
                pls.eigvals, pls.eigvecs, pls.propagators, omega, pls.basis,
                newpulse.n_opers[~here], newpulse.n_coeffs[~here, seg[i]:seg[i + 1]], pls.dt,
                t=pls.t)
        return R

    newpulse.omega = omega
    newpulse._defer_by_products()           # total phases, Liouville propagator: on first read
    mode = 'correlations' if calc_pulse_correlation_FF or calc_second_order_FF else 'total'
    # the table rule assumes a pulse contributes the same rows wherever it stands, which holds
    # when every pulse carries every noise operator (else fall back to the plain rule)
    if present.all():
        # the whole rule in one library call: cumulative propagators, their Liouville
        # representations, cumulative phases and the sum stay on the device
        with_F = mode == 'total' and which == 'fidelity'
        wanted = dict(which=mode, return_liouville=bool(calc_second_order_FF), return_filter_function=with_F)
        residents = [pls._resident for pls in distinct]
        if (all(res is not None and res.shape is not None and res.shape[1:] == residents[0].shape[1:]
                and _same_grid(pls.omega, omega) for res, pls in zip(residents, distinct))
                and all(own_rows(i) == list(range(len(new_ids))) for i in first_position)):
            # every distinct pulse still has its control matrix in HBM (evaluated by the resident
            # pass on this grid): the table is assembled there, nothing but index, basis and the
            # durations goes in
            keep = ResidentResult() if with_F else None       # the result stays resident as well
            control_matrix, total_propagator, propagators_liouville, *F = numeric.concatenate_sequence_resident(
                residents, [pls.tau for pls in distinct], index, newpulse.basis, keep=keep, **wanted)
            if keep is not None:
                newpulse.total_propagator = total_propagator
                newpulse._frequency_data['control_matrix'] = Deferred(keep.control_matrix,
                                                                      keep.control_matrix_nbytes())
                newpulse.cache_filter_function(omega, filter_function=F[0])
                newpulse._resident = keep
                return newpulse
        else:
            table = np.array([atomic_control_matrix(i) for i in first_position])
            control_matrix, total_propagator, propagators_liouville, *F = numeric.concatenate_sequence_indexed(
                np.array([pls.total_propagator for pls in distinct]),
                np.array([pls.get_total_phases(omega) for pls in distinct]), table, index, newpulse.basis,
                **wanted)
        newpulse.total_propagator = total_propagator
        if with_F:
            newpulse._store_control_matrix(control_matrix)
            newpulse.cache_filter_function(omega, filter_function=F[0])
            return newpulse
    else:
        # Liouville representation of the propagators accumulated before each pulse: the
        # cumulative propagators in one batched device call (the representation is a
        # homomorphism; the reference multiplies the pulses' Liouville propagators up instead,
        # pulse_sequence.py:1827)
        propagators_liouville = liouville_representation(cumulative_propagators()[:-1], newpulse.basis)
        newpulse.total_propagator = cumulative_propagators()[-1]
        phases = np.array([pls.get_total_phases(omega) for pls in pulses[:-1]]).cumprod(axis=0)
        R_atomic = np.array([atomic_control_matrix(i) for i in range(len(pulses))])
        control_matrix = numeric.calculate_control_matrix_from_atomic(
            phases, R_atomic, propagators_liouville, which=mode)
    if calc_second_order_FF:
        # each pulse's own second-order filter function (cached or computed now), in the new
        # pulse's noise-operator order; control_matrix holds the summands of the sequence's
        def atomic_second_order(i):
            pls = pulses[i]
            order = [list(pls.n_oper_identifiers).index(old) for new in new_ids
                     for old, mapped in n_map[i].items() if mapped == new]
            F2 = pls.get_filter_function(omega, order=2, show_progressbar=show_progressbar)
            return F2[np.ix_(order, order)]
        cache = {}
        for i, k in enumerate(index):
            if k not in cache:
                cache[k] = atomic_second_order(i)
        F2_atomic = np.array([cache[k] for k in index])
        filter_function_2 = numeric.calculate_second_order_filter_function_from_atomic(
            F2_atomic, control_matrix, propagators_liouville)
        newpulse.cache_filter_function(omega, filter_function=filter_function_2, order=2)
        if not calc_pulse_correlation_FF:
            control_matrix = control_matrix.sum(axis=0)
    newpulse.cache_filter_function(omega, control_matrix, which=which)
    return newpulse


def concatenate_periodic(pulse, repeats, check_invertible=True):
    r"""Concatenate *repeats* periods of *pulse* (reference pulse_sequence.py:1890-1973).  If the
    pulse has a cached control matrix, the new pulse's filter function is computed too.

    The reference sums the geometric series :math:`\tilde{\mathcal B}^{(1)}\sum_g(e^{i\omega T}
    \mathcal Q^{(1)})^g` in closed form with one matrix inverse per frequency (and falls back to
    the plain sum where the inverse fails, *check_invertible*); here the series is summed on the
    device by doubling (:func:`numeric.calculate_control_matrix_periodic`), which needs no inverse
    and no fallback -- *check_invertible* is accepted and ignored."""
    if not isinstance(pulse, PulseSequence):
        raise TypeError('Can only concatenate PulseSequences!')
    repeats = int(repeats)
    newpulse = PulseSequence.from_arrays(
        c_opers=pulse.c_opers, c_oper_identifiers=pulse.c_oper_identifiers,
        c_coeffs=np.tile(pulse.c_coeffs, (1, repeats)),
        n_opers=pulse.n_opers, n_oper_identifiers=pulse.n_oper_identifiers,
        n_coeffs=np.tile(pulse.n_coeffs, (1, repeats)),
        dt=np.tile(pulse.dt, repeats), basis=pulse.basis)
    newpulse.tau = repeats*pulse.tau
    if not pulse.is_cached('control_matrix'):
        return newpulse
    omega = pulse.omega
    newpulse.total_propagator = np.linalg.matrix_power(pulse.total_propagator, repeats)
    newpulse.cache_total_phases(omega)
    control_matrix = numeric.calculate_control_matrix_periodic(
        pulse.get_total_phases(omega), pulse.get_control_matrix(omega),
        pulse.total_propagator_liouville, repeats, check_invertible)
    newpulse.cache_filter_function(omega, control_matrix)
    return newpulse


# ---- qubit registers: remap and extend (reference pulse_sequence.py:1976-2625) ------------------
def _map_identifiers(identifiers, mapping):
    """New identifiers and the permutation that sorts them (identity if no mapping)."""
    if mapping is None:
        return np.asarray(identifiers), np.arange(len(identifiers))
    mapped = np.array([mapping[identifier] for identifier in identifiers])
    return mapped, np.argsort(mapped)


def _default_extend_mapping(identifiers, mapping, qubits):
    """Default identifier mapping of extend(): the qubit indices appended, 'X' -> 'X_0', 'XY_12'."""
    if mapping is not None:
        return mapping
    suffix = ''.join(str(q) for q in qubits) if np.ndim(qubits) else str(qubits)
    return {identifier: f'{identifier}_{suffix}' for identifier in identifiers}


def remap(pulse, order, d_per_qubit=2, oper_identifier_mapping=None):
    """Permute the qubits of *pulse*'s register: the factor at position ``j`` of every tensor
    product becomes the one at ``order[j]`` (reference pulse_sequence.py:1976-2120).  Cached
    attributes are carried over: the diagonalisation by permuting tensor factors; the filter
    function as is; the control matrix and Liouville propagator -- for a Pauli basis, whose elements
    are tensor products -- by permuting basis elements."""
    from .basis import remap_pauli_basis_elements
    N = int(round(np.log(pulse.d)/np.log(d_per_qubit)))
    dims = [[d_per_qubit]*N]*2
    c_opers = util.tensor_transpose(pulse.c_opers, order, dims)
    n_opers = util.tensor_transpose(pulse.n_opers, order, dims)
    c_ids, c_sort = _map_identifiers(pulse.c_oper_identifiers, oper_identifier_mapping)
    n_ids, n_sort = _map_identifiers(pulse.n_oper_identifiers, oper_identifier_mapping)
    remapped = PulseSequence.from_arrays(
        c_opers=c_opers[c_sort], c_oper_identifiers=c_ids[c_sort], c_coeffs=pulse.c_coeffs[c_sort],
        n_opers=n_opers[n_sort], n_oper_identifiers=n_ids[n_sort], n_coeffs=pulse.n_coeffs[n_sort],
        dt=pulse.dt, basis=pulse.basis)
    for attr in ('t', 'tau'):
        if attr in pulse._data:
            setattr(remapped, attr, getattr(pulse, attr))
    if pulse.is_cached('eigvals'):
        remapped.eigvals = util.tensor_transpose(pulse.eigvals, order, dims[:1], rank=1)
    for attr in ('eigvecs', 'propagators', 'total_propagator'):
        if pulse.is_cached(attr):
            setattr(remapped, attr, util.tensor_transpose(getattr(pulse, attr), order, dims))
    if not pulse.is_cached('omega'):
        return remapped
    omega = pulse.omega
    if pulse.is_cached('total_phases'):
        remapped.cache_total_phases(omega, pulse.get_total_phases(omega))
    if pulse.is_cached('filter_function'):
        remapped.cache_filter_function(
            omega, filter_function=pulse.get_filter_function(omega)[np.ix_(n_sort, n_sort)])
    if pulse.is_cached('total_propagator_liouville') or pulse.is_cached('control_matrix'):
        if pulse.basis.btype != 'Pauli':
            warn('pulse does not have a separable basis which is needed to retain cached control '
                 'matrices.')
            return remapped
        perm = remap_pauli_basis_elements(order, N)
        if pulse.is_cached('total_propagator_liouville'):
            L = np.empty_like(pulse.total_propagator_liouville)
            L[np.ix_(perm, perm)] = pulse.total_propagator_liouville
            remapped.total_propagator_liouville = L
        if pulse.is_cached('control_matrix'):
            R_old = pulse.get_control_matrix(omega)
            R = np.empty_like(R_old)
            R[np.ix_(np.argsort(n_sort), perm)] = R_old
            remapped.cache_control_matrix(omega, R)
    return remapped


def extend(pulse_to_qubit_mapping, N=None, d_per_qubit=2, additional_noise_Hamiltonian=None,
           cache_diagonalization=None, cache_filter_function=None, omega=None,
           show_progressbar=False):
    r"""Map pulses defined on one or few qubits onto a register of *N* qubits (reference
    pulse_sequence.py:2123-2625).

    pulse_to_qubit_mapping: sequence of ``(pulse, qubit)`` or ``(pulse, qubit, identifier_mapping)``
    with *qubit* an int or a tuple of ints (a multi-qubit pulse; its register is permuted first if
    the tuple is not ascending).  Operators are padded with identities on all other qubits and get
    the qubit indices appended to their identifiers unless a mapping is given.
    additional_noise_Hamiltonian: noise operators on the whole register (e.g. crosstalk).

    For Pauli bases -- whose elements are tensor products -- cached data survive: the
    diagonalisation as products of the embedded factors, control matrices and filter functions
    by placing each pulse's rows at the equivalent basis elements of the register's Pauli basis
    (scaled by the dimension of the padding); rows of the additional noise operators are computed
    from scratch on the device."""
    from .basis import equivalent_pauli_basis_elements
    multi, single = [], []           # (pulse, qubits, id_mapping)
    taken = []
    for entry in pulse_to_qubit_mapping:
        pulse, qubit = entry[0], entry[1]
        id_mapping = entry[2] if len(entry) > 2 else None
        if not isinstance(pulse, PulseSequence):
            raise TypeError('Can only extend PulseSequences!')
        if np.ndim(qubit):
            qubit = tuple(int(q) for q in qubit)
            taken.extend(qubit)
            ascending = tuple(sorted(qubit))
            if qubit != ascending:
                try:
                    pulse = remap(pulse, np.argsort(qubit), d_per_qubit)
                except ValueError as err:
                    raise ValueError(f'Could not remap {pulse!r} mapped to qubits {qubit}. Do the '
                                     'dimensions match?') from err
            multi.append((pulse, list(ascending), id_mapping))
        else:
            taken.append(int(qubit))
            single.append((pulse, int(qubit), id_mapping))
    if not all(pulse.d == d_per_qubit for pulse, _, _ in single):
        raise ValueError(f'Not all single-qubit pulses have dimension d_per_qubit = {d_per_qubit}.')
    if not all(pulse.d == d_per_qubit**len(qubits) for pulse, qubits, _ in multi):
        raise ValueError('Not all multi-qubit pulses have correct dimension!')
    entries = multi + single
    pulses = [entry[0] for entry in entries]
    if not util.all_array_equal(pulse.dt for pulse in pulses):
        raise ValueError('All pulses should be defined on the same time steps')
    if len(set(taken)) != len(taken):
        raise ValueError('Qubit clash: multiple pulses mapped to same qubit!')
    if N is None:
        N = max(taken) + 1
    elif max(taken) + 1 > N:
        raise ValueError('Number of qubits N smaller than highest qubit index + 1 = '
                         f'{max(taken) + 1}')
    if len(entries) == 1:
        if multi and N == len(multi[0][1]):
            warn('Single multi-qubit pulse given and mapped to its original qubits. Returning the '
                 'same.')
            return multi[0][0]
        if single and N == 1:
            warn('Single single-qubit pulse given and mapped to its original qubit. Returning the '
                 'same.')
            return single[0][0]

    if cache_filter_function is not False:
        # the grid the pulses agree on, if they do
        grids = [pulse._frequency_data.get('omega') for pulse in pulses]
        common = grids[0] if all(g is not None and _same_grid(g, grids[0]) for g in grids) \
            else None
        if cache_filter_function is None:
            # by default the filter function is carried over exactly when every pulse brings one
            cache_filter_function = common is not None and all(
                'control_matrix' in pulse._frequency_data for pulse in pulses)
            omega = common if cache_filter_function else omega
        elif omega is None:
            if common is None:
                raise ValueError('Filter function should be cached but omega was not provided and '
                                 'could not be inferred.')
            omega = common
    needs_eigensystem = cache_filter_function and additional_noise_Hamiltonian is not None
    if cache_diagonalization is None:
        cache_diagonalization = bool(needs_eigensystem) or all(
            key in pulse._data for pulse in pulses for key in _DIAGONALIZATION)
    elif not cache_diagonalization and additional_noise_Hamiltonian is not None:
        raise ValueError('Additional noise Hamiltonian given and cache_diagonalization set to '
                         'False but required.')

    def register_qubits(qubits):
        return list(qubits) if np.ndim(qubits) else [qubits]

    d = d_per_qubit**N
    n_dt = len(pulses[0].dt)
    c_opers, c_ids, c_coeffs, n_opers, n_ids, n_coeffs = [], [], [], [], [], []
    for pulse, qubits, id_mapping in entries:
        where = register_qubits(qubits)
        c_map = _default_extend_mapping(pulse.c_oper_identifiers, id_mapping, qubits)
        n_map = _default_extend_mapping(pulse.n_oper_identifiers, id_mapping, qubits)
        c_ids.extend(_map_identifiers(pulse.c_oper_identifiers, c_map)[0])
        n_ids.extend(_map_identifiers(pulse.n_oper_identifiers, n_map)[0])
        c_opers.extend(util.embed_in_register(pulse.c_opers, where, N, d_per_qubit))
        n_opers.extend(util.embed_in_register(pulse.n_opers, where, N, d_per_qubit))
        c_coeffs.extend(pulse.c_coeffs)
        n_coeffs.extend(pulse.n_coeffs)
    n_pulse_nops = len(n_ids)
    if additional_noise_Hamiltonian is not None:
        add_opers, add_ids, add_coeffs = _parse_hamiltonian(additional_noise_Hamiltonian, n_dt,
                                                            'H_n')
        if add_opers.shape[1:] != (d, d):
            raise ValueError(f'Expected additional noise operators to have dimensions {(d, d)}, '
                             f'not {add_opers.shape[1:]}.')
        duplicates = set(n_ids).intersection(add_ids)
        if duplicates:
            raise ValueError(f'Found duplicate noise operator identifiers: {duplicates}')
        n_opers.extend(add_opers)
        n_coeffs.extend(add_coeffs)
        n_ids.extend(add_ids)

    btypes = {pulse.basis.btype for pulse in pulses}
    if len(btypes) != 1:
        warn('Not all pulses had the same basis type. Cannot retain cached control matrices.')
        basis = Basis.ggm(d)
    elif btypes == {'Pauli'}:
        basis = Basis.pauli(N)
    elif btypes == {'GGM'}:
        warn('Original pulses had GGM basis which is not separable into a tensor product. Cannot '
             'retain cached control matrices.')
        basis = Basis.ggm(d)
    else:
        warn('Original pulses had custom basis which I cannot extend.')
        basis = Basis.ggm(d)

    c_sort, n_sort = np.argsort(c_ids), np.argsort(n_ids)

    def in_order(items, order):
        return np.asarray(items)[order]
    newpulse = PulseSequence.from_arrays(
        in_order(c_opers, c_sort), in_order(c_ids, c_sort), in_order(c_coeffs, c_sort),
        in_order(n_opers, n_sort), in_order(n_ids, n_sort), in_order(n_coeffs, n_sort),
        pulses[0].dt, basis)
    newpulse._data.update((key, pulses[0]._data[key]) for key in ('t', 'tau')
                          if key in pulses[0]._data)
    if newpulse.basis.btype != 'Pauli':
        # no tensor-product structure to exploit: evaluate what was asked for from scratch
        # (the filter function brings the diagonalisation with it)
        if cache_filter_function:
            newpulse.cache_filter_function(omega)
        elif cache_diagonalization:
            newpulse.diagonalize()
        return newpulse

    def embedded_product(attr):
        """Product over the pulses of their embedded attribute: the factors act on disjoint qubits,
        so this is the tensor product in register order."""
        out = None
        for pulse, qubits, _ in entries:
            factor = util.embed_in_register(getattr(pulse, attr), register_qubits(qubits), N,
                                            d_per_qubit)
            out = factor if out is None else out @ factor
        return out

    if cache_diagonalization:
        eigvals = np.zeros((n_dt, d))
        for pulse, qubits, _ in entries:
            eigvals += util.embed_in_register(pulse.eigvals, register_qubits(qubits), N, d_per_qubit,
                                              rank=1)
        newpulse.eigvals = eigvals
        newpulse.eigvecs = embedded_product('eigvecs')
        newpulse.propagators = embedded_product('propagators')
        newpulse.total_propagator = newpulse.propagators[-1]
    elif all(pulse.is_cached('total_propagator') for pulse in pulses):
        newpulse.total_propagator = embedded_product('total_propagator')

    if cache_filter_function:
        newpulse.omega = omega
        W = len(newpulse.omega)
        control_matrix = np.zeros((len(n_ids), d**2, W), dtype=complex)
        filter_function = np.zeros((len(n_ids), len(n_ids), W), dtype=complex)
        row = 0
        for pulse, qubits, _ in entries:
            where = register_qubits(qubits)
            rows = slice(row, row + len(pulse.n_opers))
            row += len(pulse.n_opers)
            scale = d_per_qubit**(N - len(where))
            control_matrix[rows, equivalent_pauli_basis_elements(where, N)] = \
                pulse.get_control_matrix(omega, show_progressbar)*np.sqrt(scale)
            filter_function[rows, rows] = pulse.get_filter_function(
                omega, show_progressbar=show_progressbar)*scale
        if additional_noise_Hamiltonian is not None:
            add_idx = util.get_indices_from_identifiers(newpulse.n_oper_identifiers,
                                                        n_ids[n_pulse_nops:])
            control_matrix[n_pulse_nops:] = numeric.calculate_control_matrix_from_scratch(
                newpulse.eigvals, newpulse.eigvecs, newpulse.propagators, omega, newpulse.basis,
                newpulse.n_opers[add_idx], newpulse.n_coeffs[add_idx], newpulse.dt, t=newpulse.t)
            filter_function[n_pulse_nops:, n_pulse_nops:] = numeric.calculate_filter_function(
                control_matrix[n_pulse_nops:])
        newpulse.cache_total_phases(omega)
        newpulse.total_propagator_liouville = liouville_representation(newpulse.total_propagator,
                                                                       newpulse.basis)
        newpulse.cache_control_matrix(omega, control_matrix[n_sort])
        newpulse.cache_filter_function(omega,
                                       filter_function=filter_function[np.ix_(n_sort, n_sort)])
    return newpulse
