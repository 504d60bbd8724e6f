"""The numeric hot path, served by hand-written HIP kernels behind libffk's C ABI.

Same free-function surface as ``filter_functions/numeric.py`` for the path:

====================================================  ==============================
this module                                           reference (file:line)
====================================================  ==============================
:func:`diagonalize`                                   numeric.py:1886-1935
:func:`calculate_control_matrix_from_scratch`         numeric.py:707-881
:func:`calculate_noise_operators_from_scratch`        numeric.py:456-618
:func:`calculate_filter_function`                     numeric.py:1413-1467
:func:`infidelity`                                    numeric.py:2062-2334
:func:`calculate_decay_amplitudes`                    numeric.py:1194-1337
:func:`calculate_cumulant_function`                   numeric.py:957-1191
:func:`calculate_second_order_filter_function_from_scratch`  numeric.py:1470-1699 (:170-256)
:func:`calculate_second_order_filter_function_from_atomic`   numeric.py:1702-1818
:func:`calculate_frequency_shifts`                    numeric.py:1340-1410
:func:`error_transfer_matrix`                         numeric.py:1938-2059
====================================================  ==============================

Inputs are borrowed NumPy arrays, outputs are fresh C-contiguous ``complex128`` /
``float64`` arrays with the reference's shapes.  Every function raises if the HIP
library or a GPU is missing -- there is no CPU fallback in the product.
"""
import ctypes
from warnings import warn

import numpy as np

from . import _lib, util
from ._lib import as_c128, as_f64, check, ptr

__all__ = ['diagonalize', 'calculate_control_matrix_from_scratch',
           'calculate_noise_operators_from_scratch', 'calculate_filter_function', 'infidelity',
           'calculate_control_matrix_from_atomic', 'calculate_control_matrix_from_atomic_indexed',
           'calculate_noise_operators_from_atomic', 'calculate_control_matrix_periodic',
           'calculate_pulse_correlation_filter_function', 'calculate_decay_amplitudes',
           'calculate_cumulant_function', 'error_transfer_matrix',
           'calculate_second_order_filter_function_from_scratch',
           'calculate_second_order_filter_function_from_atomic', 'calculate_frequency_shifts']


def _check_d(d, templated=False, what=None):
    """The main path (diagonalize, control matrix / noise operators, filter function, Liouville
    representation) serves d <= 64; *templated* entry points (kernels compiled per dimension) d <= 16."""
    limit = _lib.MAX_D_TEMPLATED if templated else _lib.MAX_D
    if not 2 <= d <= limit:
        hint = ''
        if templated and 2 <= d <= _lib.MAX_D:
            hint = (f'  (The kernels behind it are compiled per dimension up to {limit}; the control matrix, noise '
                    f'operators, filter function, infidelity, decay amplitudes, cumulant function and error transfer '
                    f'matrix themselves are served up to d = {_lib.MAX_D}: call them without it.)')
        raise ValueError(f'Hilbert space dimension d={d} unsupported' + (f' for {what}' if what else '')
                         + f': need 2 <= d <= {limit}.' + hint)


def diagonalize(hamiltonian, dt):
    r"""Diagonalise a piecewise-constant Hamiltonian (reference numeric.py:1886-1935).

    Parameters
    ----------
    hamiltonian: array_like, shape (n_dt, d, d)
        Only the lower triangle is read (``numpy.linalg.eigh`` default).
    dt: array_like, shape (n_dt,)

    Returns
    -------
    eigvals (n_dt, d) ascending; eigvecs (n_dt, d, d), eigenvectors in columns (defined up to
    the usual phase / degenerate-subspace gauge); propagators (n_dt+1, d, d) with
    ``propagators[0] = 1`` and ``propagators[g+1] = V_g exp(-i D_g dt_g) V_g^dag propagators[g]``.
    """
    H = as_c128(hamiltonian)
    dt = as_f64(dt)
    if H.ndim != 3 or H.shape[1] != H.shape[2]:
        raise ValueError(f'Expected hamiltonian of shape (n_dt, d, d), not {H.shape}.')
    G, d, _ = H.shape
    _check_d(d)
    if dt.shape != (G,):
        raise ValueError(f'Expected dt of shape ({G},), not {dt.shape}.')
    eigvals = np.empty((G, d), dtype=np.float64)
    eigvecs = np.empty((G, d, d), dtype=np.complex128)
    propagators = np.empty((G + 1, d, d), dtype=np.complex128)
    check(_lib.load().ffk_diagonalize(ptr(H), ptr(dt), G, d, ptr(eigvals), ptr(eigvecs),
                                      ptr(propagators)))
    return eigvals, eigvecs, propagators


def _prepare(eigvals, eigvecs, propagators, omega, n_opers, n_coeffs, dt, t):
    eigvals = as_f64(eigvals)
    eigvecs = as_c128(eigvecs)
    propagators = as_c128(propagators)
    omega = as_f64(omega)
    n_opers = as_c128(n_opers)
    n_coeffs = as_f64(n_coeffs)
    dt = as_f64(dt)
    if eigvals.ndim != 2:
        raise ValueError(f'Expected eigvals of shape (n_dt, d), not {eigvals.shape}.')
    G, d = eigvals.shape
    _check_d(d)
    if t is None:
        # same expression as the reference (numeric.py:796-797) so that omega*t rounds identically
        t = np.concatenate(([0], np.asarray(dt).cumsum()))
    t = as_f64(t)
    A = len(n_opers)
    for name, arr, shape in (('eigvecs', eigvecs, (G, d, d)), ('propagators', propagators, (G + 1, d, d)),
                             ('n_opers', n_opers, (A, d, d)), ('n_coeffs', n_coeffs, (A, G)),
                             ('dt', dt, (G,)), ('t', t, (G + 1,))):
        if arr.shape != shape:
            raise ValueError(f'Expected {name} of shape {shape}, not {arr.shape}.')
    if omega.ndim != 1:
        raise ValueError(f'Expected omega to be one-dimensional, not {omega.shape}.')
    return eigvals, eigvecs, propagators, omega, n_opers, n_coeffs, dt, t, G, d, A


def calculate_control_matrix_from_scratch(eigvals, eigvecs, propagators, omega, basis, n_opers,
                                          n_coeffs, dt, t=None, show_progressbar=False,
                                          cache_intermediates=False, out=None):
    r"""Control matrix :math:`\tilde{\mathcal B}_{\alpha k}(\omega)` of a pulse, from scratch
    (reference numeric.py:707-881).

    Returns ``control_matrix`` of shape (n_nops, n_basis, n_omega); with
    ``cache_intermediates=True`` returns ``(control_matrix, intermediates)`` where
    *intermediates* has the reference's keys (numeric.py:871-878).  ``out`` (if given) is
    overwritten in place and returned.  ``show_progressbar`` is accepted and ignored: the
    per-segment loop it decorated runs inside one kernel.
    """
    (eigvals, eigvecs, propagators, omega, n_opers, n_coeffs, dt, t,
     G, d, A) = _prepare(eigvals, eigvecs, propagators, omega, n_opers, n_coeffs, dt, t)
    basis_arr = as_c128(np.asarray(basis))
    if basis_arr.ndim != 3 or basis_arr.shape[1:] != (d, d):
        raise ValueError(f'Expected basis of shape (n_basis, {d}, {d}), not {basis_arr.shape}.')
    N = len(basis_arr)
    W = len(omega)
    if out is None:
        R = np.empty((A, N, W), dtype=np.complex128)
    else:
        if out.shape != (A, N, W) or out.dtype != np.complex128 or not out.flags.c_contiguous:
            raise ValueError(f'out must be a C-contiguous complex128 array of shape {(A, N, W)}.')
        R = out
    lib = _lib.load()
    if W == 0:
        R[...] = 0
    else:
        if cache_intermediates:
            _check_d(d, templated=True, what='cache_intermediates=True')
        check(lib.ffk_control_matrix(ptr(eigvals), ptr(eigvecs), ptr(propagators), ptr(omega), W,
                                     ptr(basis_arr), N, ptr(n_opers), A, ptr(n_coeffs), ptr(dt),
                                     ptr(t), G, d, 0, ptr(R), None))
    if not cache_intermediates:
        return R

    inter = dict(
        n_opers_transformed=np.empty((A, G, d, d), dtype=np.complex128),
        eigvecs_propagated=np.empty((G, d, d), dtype=np.complex128),
        basis_transformed=np.empty((G, N, d, d), dtype=np.complex128),
        phase_factors=np.empty((G, W), dtype=np.complex128),
        first_order_integral=np.empty((G, W, d, d), dtype=np.complex128),
        control_matrix_step=np.empty((G, A, N, W), dtype=np.complex128),
    )
    if W > 0:
        check(lib.ffk_control_matrix_intermediates(
            ptr(eigvals), ptr(eigvecs), ptr(propagators), ptr(omega), W, ptr(basis_arr), N,
            ptr(n_opers), A, ptr(n_coeffs), ptr(dt), ptr(t), G, d,
            ptr(inter['n_opers_transformed']), ptr(inter['eigvecs_propagated']),
            ptr(inter['basis_transformed']), ptr(inter['phase_factors']),
            ptr(inter['first_order_integral']), ptr(inter['control_matrix_step'])))
    # running sum over segments, g = 1 .. G-1 (numeric.py:856-861)
    inter['control_matrix_step_cumulative'] = np.cumsum(inter['control_matrix_step'][:-1], axis=0)
    return R, inter


def calculate_noise_operators_from_scratch(eigvals, eigvecs, propagators, omega, n_opers,
                                           n_coeffs, dt, t=None, show_progressbar=False,
                                           cache_intermediates=False):
    r"""Interaction-picture noise operators :math:`\tilde B_\alpha(\omega)`, shape
    (n_omega, n_nops, d, d) (reference numeric.py:456-618)."""
    (eigvals, eigvecs, propagators, omega, n_opers, n_coeffs, dt, t,
     G, d, A) = _prepare(eigvals, eigvecs, propagators, omega, n_opers, n_coeffs, dt, t)
    W = len(omega)
    B = np.empty((W, A, d, d), dtype=np.complex128)
    if W > 0:
        check(_lib.load().ffk_control_matrix(
            ptr(eigvals), ptr(eigvecs), ptr(propagators), ptr(omega), W, None, 1, ptr(n_opers), A,
            ptr(n_coeffs), ptr(dt), ptr(t), G, d, _lib.WANT_NOISE_OPERATORS, None, ptr(B)))
    if not cache_intermediates:
        return B
    # the reference's step caches (numeric.py:586-615), materialised by a second pass
    inter = dict(n_opers_transformed=np.empty((A, G, d, d), dtype=np.complex128),
                 first_order_integral=np.empty((G, W, d, d), dtype=np.complex128),
                 phase_factors=np.empty((G, W), dtype=np.complex128),
                 noise_operators_step=np.empty((G, W, A, d, d), dtype=np.complex128))
    if W > 0:
        check(_lib.load().ffk_noise_operators_intermediates(
            ptr(eigvals), ptr(eigvecs), ptr(propagators), ptr(omega), W, ptr(n_opers), A,
            ptr(n_coeffs), ptr(dt), ptr(t), G, d, ptr(inter['n_opers_transformed']),
            ptr(inter['phase_factors']), ptr(inter['first_order_integral']),
            ptr(inter['noise_operators_step'])))
    return B, inter


@util.parse_optional_parameters(which=('fidelity', 'generalized'))
def calculate_filter_function(control_matrix, which='fidelity'):
    r"""Filter function from the control matrix (reference numeric.py:1413-1467):
    'fidelity' -> (n_nops, n_nops, n_omega), 'generalized' -> (n_nops, n_nops, d², d², n_omega)."""
    R = as_c128(control_matrix)
    if R.ndim != 3:
        raise ValueError(f'Expected control_matrix of shape (n_nops, n_basis, n_omega), not {R.shape}.')
    A, N, W = R.shape
    shape = (A, A, W) if which == 'fidelity' else (A, A, N, N, W)
    F = np.empty(shape, dtype=np.complex128)
    if F.size:
        check(_lib.load().ffk_filter_function(
            ptr(R), A, N, W, _lib.FF_FIDELITY if which == 'fidelity' else _lib.FF_GENERALIZED, ptr(F)))
    return F


def _integrate_filter_function(filter_function, spectrum, omega, idx, d, pulse=None):
    """(1 / 2 pi d) int dw Re(S F): the filter-function branch of ``_get_integrand``
    (numeric.py:323-325, 351-352, 374) and ``util.integrate`` (util.py:903-906), on the device.
    If *filter_function* is the array a resident pass of *pulse* left in HBM it is integrated
    there (no upload)."""
    omega = as_f64(omega)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    spectrum = util.parse_spectrum(spectrum, omega, idx)
    if pulse is not None:
        resident = pulse.resident_infidelity(filter_function, spectrum, idx)
        if resident is not None:
            return resident
    F = as_c128(filter_function)
    if F.ndim != 3 or F.shape[0] != F.shape[1] or F.shape[2] != len(omega):
        raise ValueError(f'Expected filter_function of shape (n_nops, n_nops, {len(omega)}), '
                         f'not {F.shape}.')
    S = as_c128(spectrum)
    n_idx = len(idx)
    out = np.empty((n_idx, n_idx) if S.ndim == 3 else (n_idx,), dtype=np.float64)
    if len(omega) < 2:
        out[...] = 0.0
        return out
    check(_lib.load().ffk_infidelity(ptr(F), F.shape[0], len(omega), ptr(S), S.ndim, ptr(omega),
                                     idx.ctypes.data_as(ctypes.c_void_p), n_idx, int(d), ptr(out)))
    return out


def _convergence_study(pulse, spectrum, settings, n_oper_identifiers):
    """``infidelity(..., test_convergence=True)``: the infidelity on frequency grids of growing
    size (contract of reference numeric.py:2250-2292).  *spectrum* is a callable ``S(omega)``,
    *settings* a mapping with optional keys omega_IR, omega_UV (default 2 pi/tau x 1e-2, 1e2),
    spacing ('linear' | 'log'), n_min, n_max, n_points.  Returns (n_samples, infidelities)."""
    if not callable(spectrum):
        raise TypeError('Spectrum should be callable when test_convergence == True.')
    if not hasattr(settings, 'get'):
        raise TypeError('omega should be dictionary with parameters when test_convergence == True.')
    scale = 2*np.pi/pulse.tau
    options = dict(omega_IR=scale*1e-2, omega_UV=scale*1e+2, spacing='linear', n_min=100,
                   n_max=500, n_points=10)
    options.update((key, settings.get(key)) for key in options if settings.get(key) is not None)
    make_grid = {'linear': np.linspace, 'log': np.geomspace}.get(options['spacing'])
    if make_grid is None:
        raise ValueError("spacing should be either 'linear' or 'log'.")
    stride = (options['n_max'] - options['n_min'])//(options['n_points'] - 1)
    n_samples = np.arange(options['n_min'], options['n_max'] + stride, stride)
    results = []
    for n in n_samples:
        grid = make_grid(options['omega_IR'], options['omega_UV'], n)
        results.append(infidelity(pulse, spectrum(grid), grid,
                                  n_oper_identifiers=n_oper_identifiers, which='total'))
    return n_samples, np.array(results)


@util.parse_optional_parameters(which=('total', 'correlations'))
def infidelity(pulse, spectrum, omega, n_oper_identifiers=None, which='total',
               show_progressbar=False, cache_intermediates=False, return_smallness=False,
               test_convergence=False):
    r"""Leading-order entanglement infidelity
    :math:`\frac{1}{2\pi d}\int d\omega\, S(\omega) F(\omega)` (reference numeric.py:2062-2334).

    Same arguments, return shapes and exceptions as the reference: spectrum of shape
    ([[n_nops,] n_nops,] n_omega); ``test_convergence`` expects a callable spectrum and a dict
    for *omega*; ``return_smallness`` additionally returns :math:`\xi`.
    """
    idx = util.get_indices_from_identifiers(pulse.n_oper_identifiers, n_oper_identifiers)

    if test_convergence:
        return _convergence_study(pulse, spectrum, omega, n_oper_identifiers)

    spectrum = np.asanyarray(spectrum)
    if which == 'total':
        if not pulse.basis.istraceless:
            # Fidelity is not simply the sum over the diagonal of the decay amplitudes: the trace
            # tensor enters (reference numeric.py:2295-2305).  Its two partial traces are formed
            # here without the N^4 tensor, the contraction with the control matrix runs on device.
            C = np.asarray(pulse.basis)
            P = np.einsum('mab,mbc->ac', C, C)                       # sum_m C_m C_m
            Q = np.einsum('mab,lbc,mcd->lad', C, C, C)               # sum_m C_m C_l C_m
            weights = as_c128(np.einsum('kab,lbc,ca->kl', C, C, P) - np.einsum('kab,lba->kl', C, Q))
            R = as_c128(pulse.get_control_matrix(omega, show_progressbar, cache_intermediates))
            A, N, W = R.shape
            filter_function = np.empty((A, A, W), dtype=np.complex128)
            check(_lib.load().ffk_filter_function_weighted(ptr(R), A, N, W, ptr(weights),
                                                           1.0/pulse.d, ptr(filter_function)))
        else:
            if pulse.nothing_cached_for(omega, cache_intermediates):
                # filter function AND integral in one library call (the filter function ends up
                # cached as after get_filter_function)
                parsed = util.parse_spectrum(spectrum, as_f64(omega), np.asarray(idx))
                infid = pulse._resident_pass(spectrum=parsed, idx=idx)
                filter_function = None
            else:
                filter_function = pulse.get_filter_function(omega, which='fidelity',
                                                            show_progressbar=show_progressbar,
                                                            cache_intermediates=cache_intermediates)
        if filter_function is not None:
            infid = _integrate_filter_function(filter_function, spectrum, omega, idx, pulse.d, pulse)
    else:
        if pulse.is_cached('omega') and not np.array_equal(pulse.omega, omega):
            raise ValueError('Pulse correlation infidelities requested '
                             'but omega not equal to cached frequencies.')
        F_pc = pulse.get_pulse_correlation_filter_function()
        # (n_pls, n_pls, n_nops, n_nops, n_omega): one device integral per pulse pair
        n_pls = F_pc.shape[0]
        first = _integrate_filter_function(F_pc[0, 0], spectrum, omega, idx, pulse.d)
        infid = np.empty((n_pls, n_pls) + first.shape)
        for g in range(n_pls):
            for h in range(n_pls):
                infid[g, h] = _integrate_filter_function(F_pc[g, h], spectrum, omega, idx, pulse.d)

    if return_smallness:
        if spectrum.ndim > 2:
            raise NotImplementedError('Smallness parameter only implemented '
                                      'for uncorrelated noise sources')
        T1 = util.integrate(spectrum, omega)/(2*np.pi)
        T2 = (pulse.dt*pulse.n_coeffs[idx]).sum(axis=-1)**2
        T3 = util.abs2(pulse.n_opers[idx]).sum(axis=(1, 2))
        return infid, np.sqrt((T1*T2*T3).sum())
    return infid


@util.parse_optional_parameters(which=('total', 'correlations'))
def calculate_control_matrix_from_atomic(phases, control_matrix_atomic, propagators_liouville,
                                         show_progressbar=False, which='total'):
    r"""Control matrix of a concatenated sequence from those of its pulses
    (reference numeric.py:621-704):

    .. math:: \tilde{\mathcal B}(\omega) = \sum_g e^{i\omega t_{g-1}}
              \tilde{\mathcal B}^{(g)}(\omega)\mathcal Q^{(g-1)}.

    phases: (G-1, n_omega) cumulated phase factors; control_matrix_atomic: (G, n_nops, d², n_omega);
    propagators_liouville: (G-1, d², d²) cumulated Liouville propagators.  Returns (n_nops, d²,
    n_omega) for ``which='total'``, every summand (G, n_nops, d², n_omega) for 'correlations'.
    """
    R_atomic = as_c128(control_matrix_atomic)
    if R_atomic.ndim != 4:
        raise ValueError('Expected control_matrix_atomic of shape (G, n_nops, n_basis, n_omega), '
                         f'not {R_atomic.shape}.')
    G, A, N, W = R_atomic.shape
    phases = as_c128(phases)
    L = np.asarray(propagators_liouville)
    l_is_complex = np.iscomplexobj(L)
    L = as_c128(L) if l_is_complex else as_f64(L)
    if G > 1:
        if phases.shape[0] < G - 1 or phases.shape[1:] != (W,):
            raise ValueError(f'Expected phases of shape ({G - 1}, {W}), not {phases.shape}.')
        if L.shape[0] < G - 1 or L.shape[1:] != (N, N):
            raise ValueError(f'Expected propagators_liouville of shape ({G - 1}, {N}, {N}), '
                             f'not {L.shape}.')
        phases = np.ascontiguousarray(phases[:G - 1])
        L = np.ascontiguousarray(L[:G - 1])
    out = np.empty((G, A, N, W) if which == 'correlations' else (A, N, W), dtype=np.complex128)
    if out.size:
        check(_lib.load().ffk_control_matrix_from_atomic(
            ptr(phases) if G > 1 else None, ptr(R_atomic), ptr(L) if G > 1 else None,
            int(l_is_complex), G, A, N, W, int(which == 'correlations'), ptr(out)))
    return out


def calculate_noise_operators_from_atomic(phases, noise_operators_atomic, propagators,
                                          show_progressbar=False):
    r"""Interaction-picture noise operators of a sequence from those of its atomic pulses, the
    Hilbert-space twin of :func:`calculate_control_matrix_from_atomic` (reference
    numeric.py:377-453): :math:`\tilde B(\omega) = \sum_g e^{i\omega t_{g-1}}
    Q_{g-1}^\dagger \tilde B^{(g)}(\omega) Q_{g-1}`.

    phases (G-1, n_omega) (extra rows are ignored like in the reference), noise_operators_atomic
    (G, n_omega, n_nops, d, d), propagators (G-1, d, d) -> (n_omega, n_nops, d, d)."""
    Ba = as_c128(noise_operators_atomic)
    if Ba.ndim != 5 or Ba.shape[-1] != Ba.shape[-2]:
        raise ValueError(f'Expected noise_operators_atomic of shape (G, n_omega, n_nops, d, d), '
                         f'not {Ba.shape}.')
    G, W, A, d = Ba.shape[:4]
    _check_d(d, templated=True)
    ph = as_c128(np.asarray(phases)[:max(G - 1, 0)])
    props = as_c128(np.asarray(propagators)[:max(G - 1, 0)])
    if G > 1 and (ph.shape != (G - 1, W) or props.shape != (G - 1, d, d)):
        raise ValueError(f'phases {np.shape(phases)} / propagators {np.shape(propagators)} do not '
                         f'match {G} atomic pulses, {W} frequencies and d = {d}.')
    out = np.empty((W, A, d, d), dtype=np.complex128)
    check(_lib.load().ffk_noise_operators_from_atomic(ptr(ph) if G > 1 else None, ptr(Ba),
                                                      ptr(props) if G > 1 else None, G, W, A, d,
                                                      ptr(out)))
    return out


@util.parse_optional_parameters(which=('total', 'correlations'))
def calculate_control_matrix_from_atomic_indexed(total_phases, control_matrix_table, index,
                                                 propagators_liouville, which='total'):
    """The concatenation rule for a sequence drawn from few distinct pulses (no reference
    counterpart as a free function; it is what ``concatenate`` does when the same PulseSequence
    objects repeat, cf. examples/randomized_benchmarking.py:76-81).

    total_phases: (T, n_omega) total phase factors of the T distinct pulses;
    control_matrix_table: (T, n_nops, d², n_omega); index: (G,) position -> distinct pulse;
    propagators_liouville: (G-1, d², d²) cumulated.  The cumulated phase factors are formed on
    the device, the tables stay cache resident."""
    tp = as_c128(total_phases)
    table = as_c128(control_matrix_table)
    index = np.ascontiguousarray(index, dtype=np.int32)
    if table.ndim != 4 or tp.shape != (table.shape[0], table.shape[3]):
        raise ValueError('Expected control_matrix_table (T, n_nops, n_basis, n_omega) and '
                         f'total_phases (T, n_omega), not {table.shape} and {tp.shape}.')
    T, A, N, W = table.shape
    G = len(index)
    if G < 1 or index.min() < 0 or index.max() >= T:
        raise ValueError('index must be a non-empty sequence of values in [0, T).')
    L = np.asarray(propagators_liouville)
    l_is_complex = np.iscomplexobj(L)
    L = as_c128(L) if l_is_complex else as_f64(L)
    if G > 1 and (L.shape[0] < G - 1 or L.shape[1:] != (N, N)):
        raise ValueError(f'Expected propagators_liouville of shape ({G - 1}, {N}, {N}), not {L.shape}.')
    L = np.ascontiguousarray(L[:max(G - 1, 0)])
    out = np.empty((G, A, N, W) if which == 'correlations' else (A, N, W), dtype=np.complex128)
    check(_lib.load().ffk_control_matrix_from_atomic_indexed(
        ptr(tp), ptr(table), index.ctypes.data_as(ctypes.c_void_p), ptr(L) if G > 1 else None,
        int(l_is_complex), T, G, A, N, W, int(which == 'correlations'), ptr(out)))
    return out


def concatenate_sequence_indexed(total_propagators, total_phases, control_matrix_table, index, basis,
                                 which='total', return_liouville=False, return_filter_function=False):
    """The concatenation rule for a sequence drawn from T distinct pulses in ONE library call
    (reference pulse_sequence.py:1812-1840, spread there over ``util.adot``,
    ``liouville_representation``, ``cumprod`` and ``calculate_control_matrix_from_atomic``): the
    cumulative propagators of the sequence, their Liouville representations, the cumulative phases
    and the sum all stay on the device.

    total_propagators: (T, d, d); total_phases: (T, n_omega); control_matrix_table:
    (T, n_nops, d², n_omega); index: (G,) position -> distinct pulse.  Returns the control matrix
    ((n_nops, d², n_omega), or (G, ...) for which='correlations'), the sequence's total propagator
    and -- with *return_liouville* -- the (G-1, d², d²) cumulative Liouville propagators (else
    None); with *return_filter_function* (which='total') a fourth item, the fidelity filter function
    of the result."""
    U = as_c128(total_propagators)
    tp = as_c128(total_phases)
    table = as_c128(control_matrix_table)
    barr = as_c128(np.asarray(basis))
    index = np.ascontiguousarray(index, dtype=np.int32)
    if (table.ndim != 4 or tp.shape != (table.shape[0], table.shape[3]) or U.ndim != 3
            or len(U) != len(table) or U.shape[1] != U.shape[2]):
        raise ValueError('Expected total_propagators (T, d, d), total_phases (T, n_omega) and '
                         f'control_matrix_table (T, n_nops, n_basis, n_omega), not {U.shape}, '
                         f'{tp.shape} and {table.shape}.')
    T, A, N, W = table.shape
    d = U.shape[-1]
    G = len(index)
    if barr.shape != (N, d, d):
        raise ValueError(f'Expected basis of shape ({N}, {d}, {d}), not {barr.shape}.')
    if G < 1 or index.min() < 0 or index.max() >= T:
        raise ValueError('index must be a non-empty sequence of values in [0, T).')
    hermitian = getattr(basis, 'isherm', None)
    if hermitian is None:
        hermitian = np.allclose(barr, barr.conj().swapaxes(-1, -2), atol=np.finfo(complex).eps*d**3, rtol=0)
    out = np.empty((G, A, N, W) if which == 'correlations' else (A, N, W), dtype=np.complex128)
    total = np.empty((d, d), dtype=np.complex128)
    L = None
    if return_liouville:
        L = np.empty((max(G - 1, 0), N, N), dtype=np.float64 if hermitian else np.complex128)
    F = np.empty((A, A, W), dtype=np.complex128) if return_filter_function else None
    check(_lib.load().ffk_concatenate_sequence(
        ptr(U), ptr(tp), ptr(table), index.ctypes.data_as(ctypes.c_void_p), ptr(barr), int(bool(hermitian)),
        T, G, d, A, N, W, int(which == 'correlations'), ptr(out), ptr(total),
        ptr(L) if L is not None and G > 1 else None, ptr(F) if F is not None else None))
    if return_filter_function:
        return out, total, L, F
    return out, total, L


def concatenate_sequence_resident(residents, tau, index, basis, which='total', return_liouville=False,
                                  return_filter_function=False, keep=None):
    """:func:`concatenate_sequence_indexed` for distinct pulses whose control matrices are still
    resident in HBM (``_resident.ResidentResult`` objects, all on one frequency grid): the table
    never crosses PCIe, the total phases ``exp(i omega tau_k)`` are formed on the device.  *tau*:
    (T,) total durations.  Same return values -- except with *keep* (a fresh ``ResidentResult``;
    which='total' with the filter function): the summed control matrix then STAYS in HBM, owned by
    *keep* (fetch it with ``keep.control_matrix()``), and None is returned in its place."""
    index = np.ascontiguousarray(index, dtype=np.int32)
    tau = as_f64(tau)
    barr = as_c128(np.asarray(basis))
    T, G = len(residents), len(index)
    _, d, W, N, A = residents[0].shape
    if barr.shape != (N, d, d) or tau.shape != (T,):
        raise ValueError(f'Expected basis of shape ({N}, {d}, {d}) and tau ({T},), not {barr.shape} and {tau.shape}.')
    if G < 1 or index.min() < 0 or index.max() >= T:
        raise ValueError('index must be a non-empty sequence of values in [0, T).')
    hermitian = getattr(basis, 'isherm', None)
    if hermitian is None:
        hermitian = np.allclose(barr, barr.conj().swapaxes(-1, -2), atol=np.finfo(complex).eps*d**3, rtol=0)
    handles = (ctypes.c_void_p*T)(*(r.handle for r in residents))
    if keep is not None and (which != 'total' or not return_filter_function):
        raise ValueError("keep needs which='total' and return_filter_function=True")
    out = None
    if keep is None:
        out = np.empty((G, A, N, W) if which == 'correlations' else (A, N, W), dtype=np.complex128)
    total = np.empty((d, d), dtype=np.complex128)
    L = None
    if return_liouville:
        L = np.empty((max(G - 1, 0), N, N), dtype=np.float64 if hermitian else np.complex128)
    F = np.empty((A, A, W), dtype=np.complex128) if return_filter_function else None
    check(_lib.load().ffk_concatenate_sequence_resident(
        handles, ptr(tau), index.ctypes.data_as(ctypes.c_void_p), ptr(barr), int(bool(hermitian)), T, G,
        int(which == 'correlations'), ptr(out) if out is not None else None, ptr(total),
        ptr(L) if L is not None and G > 1 else None, ptr(F) if F is not None else None,
        keep.handle if keep is not None else None))
    if keep is not None:
        keep.adopt((1, d, W, N, A), F)
    if return_filter_function:
        return out, total, L, F
    return out, total, L


def calculate_control_matrix_periodic(phases, control_matrix, total_propagator_liouville, repeats,
                                      check_invertible=True):
    r"""Control matrix of *repeats* periods of a pulse from the control matrix (n_nops, d**2,
    n_omega), total phase factors (n_omega,) and Liouville total propagator (d**2, d**2) of one
    period (reference numeric.py:886-954): :math:`\tilde{\mathcal B}^{(1)}\sum_{g<G}(e^{i\omega T}
    \mathcal Q^{(1)})^g`.

    The reference sums the series in closed form with one inverse per frequency, falling back to
    the explicit sum where :math:`\mathbb I - e^{i\omega T}\mathcal Q^{(1)}` is ill conditioned
    (*check_invertible*).  Here the series is summed on the device by doubling -- about
    :math:`2\log_2 G` passes over the control matrix, no inverse -- so *check_invertible* has
    nothing to check and is ignored."""
    L = np.asarray(total_propagator_liouville)
    z, R = as_c128(phases), as_c128(control_matrix)
    repeats = int(repeats)
    if repeats < 1:
        raise ValueError('repeats must be a positive integer')
    if R.ndim != 3 or z.shape != R.shape[2:] or L.shape != (R.shape[1],)*2:
        raise ValueError('Expected control_matrix (n_nops, n_basis, n_omega), phases (n_omega,) and '
                         f'a (n_basis, n_basis) propagator, not {R.shape}, {z.shape} and {L.shape}.')
    l_is_complex = np.iscomplexobj(L)
    L = as_c128(L) if l_is_complex else as_f64(L)
    A, N, W = R.shape
    out = np.empty_like(R)
    check(_lib.load().ffk_control_matrix_periodic(ptr(z), ptr(R), ptr(L), int(l_is_complex), repeats,
                                                  A, N, W, ptr(out)))
    return out


@util.parse_optional_parameters(which=('fidelity', 'generalized'))
def calculate_pulse_correlation_filter_function(control_matrix, which='fidelity'):
    r"""Pulse-correlation filter function
    :math:`F^{(gg')}_{\alpha\beta}(\omega)` from the pulse-resolved control matrix
    (G, n_nops, d², n_omega) (reference numeric.py:1821-1883): (G, G, n_nops, n_nops, n_omega),
    or (G, G, n_nops, n_nops, d², d², n_omega) for 'generalized'."""
    R = as_c128(control_matrix)
    if R.ndim != 4:
        raise ValueError('Expected control_matrix.ndim == 4.')
    G, A, N, W = R.shape
    # 'gako,hbko->ghabo' is the fidelity filter function of the (G*A)-row control matrix
    F = calculate_filter_function(R.reshape(G*A, N, W), which)
    if which == 'fidelity':
        return np.ascontiguousarray(F.reshape(G, A, G, A, W).transpose(0, 2, 1, 3, 4))
    return np.ascontiguousarray(F.reshape(G, A, G, A, N, N, W).transpose(0, 2, 1, 3, 4, 5, 6))


@util.parse_optional_parameters(which=('total', 'correlations'))
def calculate_decay_amplitudes(pulse, spectrum, omega, n_oper_identifiers=None, which='total',
                               show_progressbar=False, cache_intermediates=False,
                               memory_parsimonious=False):
    r"""Decay amplitudes :math:`\Gamma_{\alpha\beta,kl} = \int\frac{d\omega}{2\pi}
    \tilde{\mathcal{B}}^\ast_{\alpha k}(\omega) S_{\alpha\beta}(\omega)
    \tilde{\mathcal{B}}_{\beta l}(\omega)` (reference numeric.py:1194-1337).

    Returns an array of shape ``([[G, G,] n_nops,] n_nops, d**2, d**2)`` exactly like the
    reference.  The reference builds the ``(n_nops, d**2, d**2, n_omega)`` integrand and integrates
    it (or loops over ``k`` if *memory_parsimonious*); here the integral is one FP64 matrix
    product over the frequency axis on the matrix cores, so *memory_parsimonious* changes nothing.
    """
    idx = util.get_indices_from_identifiers(pulse.n_oper_identifiers, n_oper_identifiers)
    if which == 'total':
        control_matrix = pulse.get_control_matrix(omega, show_progressbar, cache_intermediates)
    else:
        if pulse.is_cached('omega') and not np.array_equal(pulse.omega, omega):
            raise ValueError('Pulse correlation decay amplitudes requested but omega not '
                             'equal to cached frequencies.')
        control_matrix = pulse.get_pulse_correlation_control_matrix()
    return _decay_amplitudes(control_matrix, spectrum, omega, idx, which)


def _decay_amplitudes(control_matrix, spectrum, omega, idx, which):
    omega = as_f64(omega)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    S = as_c128(util.parse_spectrum(spectrum, omega, idx))
    R = as_c128(control_matrix)
    if R.ndim != (3 if which == 'total' else 4) or R.shape[-1] != len(omega):
        raise ValueError(f'control matrix of shape {R.shape} does not match which={which!r} and '
                         f'{len(omega)} frequencies.')
    n_pls = 1 if which == 'total' else R.shape[0]
    A, N, W = R.shape[-3:]
    n_idx = len(idx)
    shape = (n_idx, n_idx, N, N) if S.ndim == 3 else (n_idx, N, N)
    if which == 'correlations':
        shape = (n_pls, n_pls) + shape
    out = np.empty(shape, dtype=np.float64)
    check(_lib.load().ffk_decay_amplitudes(ptr(R), n_pls, A, N, W, ptr(S), S.ndim, ptr(omega),
                                           idx.ctypes.data_as(ctypes.c_void_p), n_idx, ptr(out)))
    return out


@util.parse_optional_parameters(which=('total', 'correlations'))
def calculate_cumulant_function(pulse, spectrum=None, omega=None, n_oper_identifiers=None,
                                which='total', second_order=False, decay_amplitudes=None,
                                frequency_shifts=None, show_progressbar=False,
                                memory_parsimonious=False, cache_intermediates=None):
    r"""Cumulant function :math:`\mathcal{K}(\tau)`, first order in the Magnus expansion
    (reference numeric.py:957-1191):

    .. math:: K_{\alpha\beta,ij} = -\frac{1}{2}\sum_{kl}\Gamma_{\alpha\beta,kl}
              \left(T_{klji} - T_{kjli} - T_{kilj} + T_{kijl}\right).

    Same arguments, shapes and errors as the reference.  The contraction with the four-element
    trace tensor is evaluated on the device without forming the tensor.  ``second_order=True`` adds
    the frequency-shift terms (reference numeric.py:1139-1141, 1166-1190), evaluated as the
    commutator with the effective Hamiltonian :math:`\sum_{kl}(\Delta_{kl}-\Delta_{lk})C_kC_l`.
    """
    have_spectrum = spectrum is not None or omega is not None
    if not have_spectrum and (decay_amplitudes is None or (second_order and frequency_shifts is None)):
        raise ValueError('Require either spectrum and frequencies or precomputed '
                         'decay amplitudes (frequency shifts)')
    if second_order and which == 'correlations':
        raise ValueError('Cannot compute correlation cumulant function for second order terms')
    if decay_amplitudes is None:
        decay_amplitudes = calculate_decay_amplitudes(
            pulse, spectrum, omega, n_oper_identifiers, which, show_progressbar,
            second_order if cache_intermediates is None else cache_intermediates,
            memory_parsimonious)
    if second_order and frequency_shifts is None:
        if memory_parsimonious:
            warn('Memory parsimonious calculation not implemented for frequency shifts.')
        frequency_shifts = calculate_frequency_shifts(pulse, spectrum, omega, n_oper_identifiers,
                                                      show_progressbar)
    if second_order and np.shape(frequency_shifts) != np.shape(decay_amplitudes):
        raise ValueError('Frequency shifts not same shape as decay amplitudes')
    return _cumulant_function(decay_amplitudes, pulse.basis,
                              frequency_shifts if second_order else None)


def calculate_frequency_shifts(pulse, spectrum, omega, n_oper_identifiers=None,
                               show_progressbar=False):
    r"""Frequency shifts :math:`\Delta_{\alpha\beta,kl} = \int\frac{d\omega}{2\pi}
    S_{\alpha\beta}(\omega) F^{(2)}_{\alpha\beta,kl}(\omega)` (reference numeric.py:1340-1410):
    shape ``([n_nops,] n_nops, d**2, d**2)``.  The second-order filter function comes from (and is
    cached on) *pulse*; the integral is one reduction kernel over the frequency axis."""
    idx = util.get_indices_from_identifiers(pulse.n_oper_identifiers, n_oper_identifiers)
    pulse.omega = omega
    if pulse.is_cached('filter_function_2'):
        F2 = pulse.get_filter_function(omega, order=2, show_progressbar=show_progressbar)
        return _frequency_shifts(F2, spectrum, omega, idx)
    # not cached yet: one device-resident pass produces F2 (returned once, for the pulse's cache,
    # like the reference's get_filter_function(order=2)) and the integral
    omega = as_f64(pulse.omega)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    S = as_c128(util.parse_spectrum(spectrum, omega, idx))
    D, V, Q = as_f64(pulse.eigvals), as_c128(pulse.eigvecs), as_c128(pulse.propagators)
    C, B = as_c128(np.asarray(pulse.basis)), as_c128(pulse.n_opers)
    s, dt = as_f64(pulse.n_coeffs), as_f64(pulse.dt)
    G, d = D.shape
    _check_d(d, templated=True)
    A, N, W, n_idx = len(B), len(C), len(omega), len(idx)
    t = np.concatenate(([0.0], dt.cumsum()))
    F2 = np.empty((A, A, N, N, W), dtype=np.complex128)
    out = np.empty((n_idx, n_idx, N, N) if S.ndim == 3 else (n_idx, N, N), dtype=np.float64)
    check(_lib.load().ffk_frequency_shifts_from_scratch(
        ptr(D), ptr(V), ptr(Q), ptr(omega), W, ptr(C), N, ptr(B), A, ptr(s), ptr(dt), ptr(t), G, d,
        ptr(S), S.ndim, idx.ctypes.data_as(ctypes.c_void_p), n_idx, ptr(F2), ptr(out)))
    pulse.cache_filter_function(omega, filter_function=F2, order=2)
    return out


def _frequency_shifts(filter_function_2, spectrum, omega, idx):
    omega = as_f64(omega)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    S = as_c128(util.parse_spectrum(spectrum, omega, idx))
    F2 = as_c128(filter_function_2)
    if F2.ndim != 5 or F2.shape[0] != F2.shape[1] or F2.shape[2] != F2.shape[3] \
            or F2.shape[-1] != len(omega):
        raise ValueError(f'Expected a second-order filter function of shape (n_nops, n_nops, d**2, '
                         f'd**2, {len(omega)}), not {F2.shape}.')
    A, N, W = F2.shape[0], F2.shape[2], F2.shape[4]
    n_idx = len(idx)
    out = np.empty((n_idx, n_idx, N, N) if S.ndim == 3 else (n_idx, N, N), dtype=np.float64)
    check(_lib.load().ffk_frequency_shifts(ptr(F2), A, N, W, ptr(S), S.ndim, ptr(omega),
                                           idx.ctypes.data_as(ctypes.c_void_p), n_idx, ptr(out)))
    return out


def calculate_second_order_filter_function_from_scratch(eigvals, eigvecs, propagators, omega, basis,
                                                        n_opers, n_coeffs, dt, intermediates=None,
                                                        show_progressbar=False,
                                                        cache_intermediates=False,
                                                        cache_cumulative=False):
    r"""Second-order filter function :math:`F^{(2)}_{\alpha\beta,kl}(\omega)`, shape
    ``(n_nops, n_nops, d**2, d**2, n_omega)`` (reference numeric.py:1470-1699, nested integral
    :170-256), one device pass: the segment loop, the control-matrix steps and their cumulative sums
    live inside the kernel, so the step caches the reference recycles (*intermediates*) are not
    needed and are ignored.  The reference's own caches of this function (``second_order_integral``
    with ``n_dt*n_omega*d**4`` entries, ``filter_function_2_step_cumulative``) are not materialised;
    with *cache_intermediates* the (unchanged) dict is returned alongside for signature parity."""
    D = as_f64(eigvals)
    V = as_c128(eigvecs)
    Q = as_c128(propagators)
    omega = as_f64(omega)
    C = as_c128(np.asarray(basis))
    B = as_c128(n_opers)
    s = as_f64(n_coeffs)
    dt = as_f64(dt)
    G, d = D.shape
    _check_d(d, templated=True)
    if cache_cumulative:
        raise NotImplementedError('cache_cumulative: the per-segment cumulative second-order filter '
                                  'function is not materialised by the device path.')
    A, N, W = len(B), len(C), len(omega)
    t = np.concatenate(([0.0], dt.cumsum()))
    out = np.empty((A, A, N, N, W), dtype=np.complex128)
    check(_lib.load().ffk_second_order_filter_function(
        ptr(D), ptr(V), ptr(Q), ptr(omega), W, ptr(C), N, ptr(B), A, ptr(s), ptr(dt), ptr(t), G, d,
        ptr(out)))
    if cache_intermediates:
        return out, (intermediates if intermediates is not None else dict())
    return out


def calculate_second_order_filter_function_from_atomic(filter_function_atomic,
                                                       control_matrix_atomic_step,
                                                       propagators_liouville):
    r"""Second-order filter function of a sequence of pulses from those of the pulses (reference
    numeric.py:1702-1818), in the rotated form of the concatenation rule

    .. math:: F^{(2)}_{\alpha\beta,kl} = \sum_g\Big[\sum_{pq}\mathcal Q^{(g-1)}_{pk}
              F^{(2,g)}_{\alpha\beta,pq}\mathcal Q^{(g-1)}_{ql} + \mathcal G^{(g)\ast}_{\alpha k}
              \sum_{g'<g}\mathcal G^{(g')}_{\beta l}\Big],

    which needs only each pulse's own second-order filter function (the reference re-evaluates the
    incomplete steps from its ``second_order_integral`` caches instead; same result).

    filter_function_atomic: (G, n_nops, n_nops, d**2, d**2, n_omega); control_matrix_atomic_step:
    (G, n_nops, d**2, n_omega), the summands of the sequence's control matrix
    (``calculate_control_matrix_from_atomic(..., which='correlations')``); propagators_liouville:
    (G-1, d**2, d**2), real (Hermitian basis)."""
    Fa = as_c128(filter_function_atomic)
    Rs = as_c128(control_matrix_atomic_step)
    if Fa.ndim != 6 or Rs.ndim != 4 or Fa.shape[0] != Rs.shape[0]:
        raise ValueError(f'Expected filter_function_atomic (G, A, A, N, N, W) and '
                         f'control_matrix_atomic_step (G, A, N, W), not {Fa.shape} and {Rs.shape}.')
    G, A, N, W = Rs.shape
    if Fa.shape[1:] != (A, A, N, N, W):
        raise ValueError(f'Expected filter_function_atomic of shape {(G, A, A, N, N, W)}, '
                         f'not {Fa.shape}.')
    if np.iscomplexobj(propagators_liouville):
        raise NotImplementedError('second-order concatenation needs a Hermitian basis (real '
                                  'Liouville representation)')
    L = as_f64(propagators_liouville)
    if G > 1 and (L.ndim != 3 or L.shape[0] < G - 1 or L.shape[1:] != (N, N)):
        raise ValueError(f'Expected propagators_liouville of shape ({G - 1}, {N}, {N}), not {L.shape}.')
    L = np.ascontiguousarray(L[:max(G - 1, 0)])
    out = np.empty((A, A, N, N, W), dtype=np.complex128)
    check(_lib.load().ffk_second_order_filter_function_from_atomic(
        ptr(Fa), ptr(Rs), ptr(L) if G > 1 else None, G, A, N, W, ptr(out)))
    return out


def _cumulant_function(decay_amplitudes, basis, frequency_shifts=None):
    N, d = basis.shape[:2]
    _check_d(d)
    G = as_f64(decay_amplitudes)
    if G.ndim < 2 or G.shape[-2:] != (N, N):
        raise ValueError(f'Expected decay amplitudes of shape (..., {N}, {N}), not {G.shape}.')
    single_qubit = int(d == 2 and basis.btype in ('Pauli', 'GGM'))
    flat = G.reshape(-1, N, N)
    out = np.empty_like(flat)
    C = as_c128(np.asarray(basis))
    step = 16384
    for lo in range(0, len(flat), step):
        part = np.ascontiguousarray(flat[lo:lo + step])
        res = np.empty_like(part)
        check(_lib.load().ffk_cumulant_function(ptr(part), len(part), N, d, ptr(C), single_qubit,
                                                ptr(res)))
        if frequency_shifts is not None:
            delta = np.ascontiguousarray(as_f64(frequency_shifts).reshape(-1, N, N)[lo:lo + step])
            check(_lib.load().ffk_cumulant_function_second_order(ptr(delta), len(delta), N, d,
                                                                 ptr(C), ptr(res)))
        out[lo:lo + step] = res
    return out.reshape(G.shape)


def error_transfer_matrix(pulse=None, spectrum=None, omega=None, n_oper_identifiers=None,
                          second_order=False, cumulant_function=None, show_progressbar=False,
                          memory_parsimonious=False, cache_intermediates=False):
    r"""Error transfer matrix :math:`\langle\tilde{\mathcal{U}}\rangle = \exp\mathcal{K}`
    (reference numeric.py:1938-2059): the cumulant function summed over the noise operators,
    exponentiated (one ``d**2 x d**2`` real matrix: scaling and squaring on the GPU, ``ffk_expm_real``; the
    reference calls ``scipy.linalg.expm``)."""
    from scipy import linalg as sla

    if cumulant_function is None:
        if any(arg is None for arg in (pulse, spectrum, omega)):
            raise ValueError('Require either precomputed cumulant function '
                             'or pulse, spectrum, and omega as arguments.')
        cumulant_function = calculate_cumulant_function(
            pulse, spectrum, omega, n_oper_identifiers, 'total', second_order,
            show_progressbar=show_progressbar, memory_parsimonious=memory_parsimonious,
            cache_intermediates=cache_intermediates)
    if not hasattr(cumulant_function, 'sum'):
        raise TypeError(f'cumulant_function invalid type: {type(cumulant_function)}')
    if np.ndim(cumulant_function) < 2:
        raise ValueError(f'cumulant_function invalid shape: {cumulant_function.shape}')
    # sum over everything but the two Liouville-space axes (noise operators, pulse pairs)
    K = cumulant_function.sum(axis=tuple(range(cumulant_function.ndim - 2)))
    if K.shape[0] != K.shape[1]:
        raise ValueError(f'cumulant_function invalid shape: {cumulant_function.shape}')
    if np.iscomplexobj(K) or not np.isfinite(K).all():
        # not a cumulant function (those are real and finite): what the reference's call does with it
        return sla.expm(K)
    K = as_f64(K)
    out = np.empty_like(K)
    check(_lib.load().ffk_expm_real(ptr(K), K.shape[0], ptr(out)))
    return out
