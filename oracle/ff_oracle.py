"""CPU oracle: a NumPy restatement of the filter_functions hot path.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module, and only as the checker / the timed CPU baseline.  The product
package ``filter_functions_amd`` never imports it and has no CPU fallback.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function
here against fixtures generated from the upstream reference itself
(``oracle/make_golden.py``, run in the build container against
/root/reference) and against the reference's own golden vector
(tests/test_precision.py:510-529 ``ref_infids``, atol 1e-12) and closed-form
dynamical-decoupling filter functions (filter_functions/analytic.py:59-88).

Every function cites the reference ``file:line`` (relative to /root/reference)
whose arithmetic it follows.  The algorithm is deliberately the reference's own
(per-segment loop, ``exp(ix)-1 = -2 sin^2(x/2) + i sin x``, Liouville-space
contraction with the generated integral), so that it doubles as the same-box
CPU baseline ("port") next to the GPU measurement.

All arrays are C-contiguous float64 / complex128.
"""
from itertools import product

import numpy as np

__all__ = [
    'paulis', 'basis_pauli', 'basis_ggm', 'basis_expand', 'ggm_expand', 'cexp', 'cexpm1',
    'integrate', 'get_sample_frequencies', 'hamiltonian', 'diagonalize',
    'first_order_integral', 'control_matrix_from_scratch', 'noise_operators_from_scratch',
    'filter_function', 'liouville_representation', 'parse_spectrum', 'infidelity_from_filter_function',
    'pauli_labels', 'equivalent_pauli_basis_elements', 'remap_pauli_basis_elements',
    'control_matrix_from_atomic',
]

# --------------------------------------------------------------------------------------
# Basis construction and indexing (bit-exact contract)
# --------------------------------------------------------------------------------------

#: Identity and Pauli matrices, filter_functions/util.py:109-118
paulis = np.array([[[1, 0], [0, 1]],
                   [[0, 1], [1, 0]],
                   [[0, -1j], [1j, 0]],
                   [[1, 0], [0, -1]]], dtype=complex)


def pauli_labels(n):
    """Labels in element order, filter_functions/basis.py:425-426."""
    return [''.join(t) for t in product('IXYZ', repeat=n)]


def basis_pauli(n):
    """n-qubit Pauli basis, shape (4**n, 2**n, 2**n).

    Element order is that of ``np.indices((4,)*n)``: last qubit fastest
    (filter_functions/basis.py:393-426).  Normalised by sqrt(2**n).
    """
    d = 2**n
    out = np.empty((4**n, d, d), dtype=complex)
    for flat, combo in enumerate(product(range(4), repeat=n)):
        elem = np.ones((1, 1), dtype=complex)
        for c in combo:
            elem = np.kron(elem, paulis[c])
        out[flat] = elem
    out /= np.sqrt(2**n)
    return out


def _ggm_offdiag_indices(d):
    """Row-major enumeration of the strict upper triangle (j < k),
    filter_functions/basis.py:460-467."""
    j = [a for a in range(d - 1) for _ in range(d - 1 - a)]
    k = [b for a in range(d - 1) for b in range(a + 1, d)]
    return np.array(j, dtype=int), np.array(k, dtype=int)


def basis_ggm(d):
    """Generalised Gell-Mann basis, shape (d**2, d, d).

    Order: identity/sqrt(d); n_sym symmetric; n_sym antisymmetric (-i at (j,k),
    +i at (k,j)); d-1 diagonal (filter_functions/basis.py:428-489).
    """
    n_sym = d*(d - 1)//2
    j, k = _ggm_offdiag_indices(d)
    inv_sqrt2 = 1/np.sqrt(2)
    lam = np.zeros((d*d, d, d), dtype=complex)
    lam[0] = np.eye(d)/np.sqrt(d)
    for s in range(n_sym):
        lam[1 + s, j[s], k[s]] = inv_sqrt2
        lam[1 + s, k[s], j[s]] = inv_sqrt2
        lam[1 + n_sym + s, j[s], k[s]] = -1j*inv_sqrt2
        lam[1 + n_sym + s, k[s], j[s]] = 1j*inv_sqrt2
    for l in range(1, d):
        elem = lam[2*n_sym + l]
        for i in range(l):
            elem[i, i] = 1
        elem[l, l] = -l
        # basis.py:484-486 divides the diagonal by sqrt(l (l+1))
        elem[range(d), range(d)] /= np.sqrt(l*(l + 1))
    return lam


def basis_expand(M, basis, hermitian=False):
    """c_j = tr(M C_j) for a normalised basis, filter_functions/basis.py:650-698."""
    coeffs = np.tensordot(M, basis, axes=[(-2, -1), (-1, -2)])
    return coeffs.real if hermitian else coeffs


def ggm_expand(M, hermitian=False):
    """Closed-form GGM expansion, filter_functions/basis.py:701-787."""
    M = np.asarray(M)
    d = M.shape[-1]
    n_sym = d*(d - 1)//2
    j, k = _ggm_offdiag_indices(d)
    cast = (lambda a: a.real) if hermitian else (lambda a: a)
    coeffs = np.zeros(M.shape[:-2] + (d*d,), dtype=float if hermitian else complex)
    coeffs[..., 0] = cast(np.trace(M, axis1=-2, axis2=-1))/np.sqrt(d)
    up = M[..., j, k]
    lo = M[..., k, j]
    coeffs[..., 1:1 + n_sym] = cast(up + lo)/np.sqrt(2)
    coeffs[..., 1 + n_sym:1 + 2*n_sym] = cast(1j*(up - lo))/np.sqrt(2)
    diag = np.diagonal(M, axis1=-2, axis2=-1)
    l = np.arange(1, d)
    coeffs[..., 1 + 2*n_sym:] = cast(np.cumsum(diag[..., :-1], axis=-1) - l*diag[..., 1:])
    coeffs[..., 1 + 2*n_sym:] /= np.sqrt(l*(l + 1))
    return coeffs


def equivalent_pauli_basis_elements(idx, N):
    """filter_functions/basis.py:790-800."""
    idx = [idx] if isinstance(idx, int) else list(idx)
    grids = np.ix_(*[range(4) if i in idx else [0] for i in range(N)])
    return np.ravel_multi_index(grids, [4]*N).ravel()


def remap_pauli_basis_elements(order, N):
    """filter_functions/basis.py:803-815."""
    tuples = np.indices((4,)*N).reshape(N, 4**N).T
    return np.array([np.ravel_multi_index([tup[i] for i in order], (4,)*N) for tup in tuples])


# --------------------------------------------------------------------------------------
# Small numeric helpers
# --------------------------------------------------------------------------------------

def cexp(x):
    """exp(ix) as cos + i sin, filter_functions/util.py:136-162."""
    x = np.asarray(x, dtype=float)
    out = np.empty(x.shape, dtype=complex)
    out.real = np.cos(x)
    out.imag = np.sin(x)
    return out


def cexpm1(x):
    """exp(ix) - 1 = -2 sin^2(x/2) + i sin(x), filter_functions/util.py:165-182."""
    x = np.asarray(x, dtype=float)
    out = np.empty(x.shape, dtype=complex)
    half = np.sin(x/2)
    out.real = -2*np.square(half)
    out.imag = np.sin(x)
    return out


def integrate(f, x):
    """Trapezoid over the last axis, filter_functions/util.py:880-906:
    sum((f[1:] + f[:-1]) * diff(x)) / 2."""
    dx = np.diff(x)
    ret = f[..., 1:] + f[..., :-1]
    ret = ret*dx
    return ret.sum(axis=-1)/2


def get_sample_frequencies(tau, dt, n_samples=300, spacing='log', include_quasistatic=False,
                           omega_min=None, omega_max=None):
    """filter_functions/util.py:1054-1093."""
    if omega_min is None:
        omega_min = 2*np.pi*1e-2/tau
    if omega_max is None:
        omega_max = 2*np.pi*1e+1/np.min(dt)
    if spacing == 'linear':
        xspace = np.linspace
    else:
        xspace = np.geomspace
    if include_quasistatic:
        return np.insert(xspace(omega_min, omega_max, n_samples - 1), 0, 0)
    return xspace(omega_min, omega_max, n_samples)


# --------------------------------------------------------------------------------------
# The hot path
# --------------------------------------------------------------------------------------

def hamiltonian(c_opers, c_coeffs):
    """H_g = sum_i a_i(g) A_i, filter_functions/pulse_sequence.py:582."""
    return np.einsum('ijk,il->ljk', c_opers, c_coeffs)


def diagonalize(H, dt):
    """Eigen-decomposition, segment propagators and cumulative propagators.

    filter_functions/numeric.py:1886-1935: ``eigh`` (lower triangle, ascending),
    P_g = V_g exp(-i D_g dt_g) V_g^dag (:1928), Q = [1, P_0, P_1 P_0, ...]
    accumulated left-to-right in segment order (:1933, util.py:868-877).
    """
    H = np.asarray(H)
    dt = np.asarray(dt, dtype=float)
    G, d, _ = H.shape
    eigvals, eigvecs = np.linalg.eigh(H)
    phases = cexp(-dt[:, None]*eigvals)                       # (G, d)
    P = (eigvecs*phases[:, None, :]) @ eigvecs.conj().transpose(0, 2, 1)
    Q = np.empty((G + 1, d, d), dtype=complex)
    Q[0] = np.identity(d)
    for g in range(G):
        Q[g + 1] = P[g] @ Q[g]
    return eigvals, eigvecs, Q


def first_order_integral(omega, eigvals_g, dt_g):
    """I[o,m,n] = (exp(i x dt) - 1)/(i x), x = omega_o + D_m - D_n; dt where x == 0.

    filter_functions/numeric.py:144-167 (argument (omega + dE)*dt, exact x != 0
    mask) with util.cexpm1 (util.py:165-182).
    """
    dE = np.subtract.outer(eigvals_g, eigvals_g)              # dE[m,n] = D_m - D_n
    x = np.add.outer(np.asarray(omega, dtype=float), dE)      # (W, d, d)
    mask = x != 0
    out = np.full(x.shape, dt_g, dtype=complex)
    num = cexpm1(x[mask]*dt_g)
    out[mask] = num/(1j*x[mask])
    return out


def _prologue(eigvals, eigvecs, propagators, n_opers, n_coeffs):
    """Q_{g}^dag V_g (numeric.py:93-95, :818) and s_a(g) V_g^dag B_a V_g
    (numeric.py:98-141, :819)."""
    n_opers = np.asarray(n_opers)
    n_coeffs = np.asarray(n_coeffs, dtype=float)
    QdV = propagators[:-1].conj().transpose(0, 2, 1) @ eigvecs              # (G,d,d)
    Vd = eigvecs.conj().transpose(0, 2, 1)
    Bbar = Vd[None] @ (n_opers[:, None] @ eigvecs[None])                   # (A,G,d,d)
    Bbar = Bbar*n_coeffs[:, :, None, None]
    return QdV, Bbar


def control_matrix_from_scratch(eigvals, eigvecs, propagators, omega, basis, n_opers, n_coeffs,
                                dt, t=None, cache_intermediates=False):
    """R[a,k,o] = sum_g e^{i w_o t_g} sum_mn Bbar^{(g)}_{a,mn} I^{(g)}_{o,mn} Cbar^{(g)}_{k,nm}.

    filter_functions/numeric.py:707-881.  The einsum 'o,jmn,omn,knm->jko' (:843) is
    restated as the (A N x d^2)(d^2 x W) product it is, segment by segment, with
    the phase factor exp(i omega t_g) (:865) applied to the generated integral.
    """
    dt = np.asarray(dt, dtype=float)
    omega = np.asarray(omega, dtype=float)
    basis = np.asarray(basis)
    if t is None:
        t = np.concatenate(([0.0], dt.cumsum()))
    G, d = eigvals.shape
    A = len(n_opers)
    N = len(basis)
    W = len(omega)
    QdV, Bbar = _prologue(eigvals, eigvecs, propagators, n_opers, n_coeffs)

    R = np.zeros((A, N, W), dtype=complex)
    if cache_intermediates:
        basis_cache = np.empty((G, N, d, d), dtype=complex)
        phase_cache = np.empty((G, W), dtype=complex)
        int_cache = np.empty((G, W, d, d), dtype=complex)
        step_cache = np.empty((G, A, N, W), dtype=complex)
        cumulative_cache = np.zeros((max(G - 1, 0), A, N, W), dtype=complex)
    for g in range(G):
        if cache_intermediates and g > 0:
            cumulative_cache[g - 1] = R
        Cbar = QdV[g].conj().T @ basis @ QdV[g]                 # (N,d,d), numeric.py:863-864
        phase = cexp(omega*t[g])                                # numeric.py:865
        integral = first_order_integral(omega, eigvals[g], dt[g])   # (W,d,d)
        # M[(a,k),(m,n)] = Bbar[a,m,n] * Cbar[k,n,m]
        M = (Bbar[:, g, None, :, :]*Cbar.transpose(0, 2, 1)[None]).reshape(A*N, d*d)
        step = (M @ (integral.reshape(W, d*d)*phase[:, None]).T).reshape(A, N, W)
        R += step
        if cache_intermediates:
            basis_cache[g] = Cbar
            phase_cache[g] = phase
            int_cache[g] = integral
            step_cache[g] = step
    if cache_intermediates:
        return R, dict(n_opers_transformed=Bbar, eigvecs_propagated=QdV,
                       basis_transformed=basis_cache, phase_factors=phase_cache,
                       first_order_integral=int_cache, control_matrix_step=step_cache,
                       control_matrix_step_cumulative=cumulative_cache)
    return R


def noise_operators_from_scratch(eigvals, eigvecs, propagators, omega, n_opers, n_coeffs, dt,
                                 t=None, cache_intermediates=False):
    """Hilbert-space twin, result (W, A, d, d).

    filter_functions/numeric.py:456-618:
    B~_a(w) = sum_g e^{i w t_g} P_g^dag [Bbar_a^{(g)} o I^{(g)}(w)] P_g, P_g = V_g^dag Q_g
    (:577 swaps the argument order of _propagate_eigenvectors).
    """
    dt = np.asarray(dt, dtype=float)
    omega = np.asarray(omega, dtype=float)
    if t is None:
        t = np.concatenate(([0.0], dt.cumsum()))
    G, d = eigvals.shape
    A = len(n_opers)
    W = len(omega)
    _, Bbar = _prologue(eigvals, eigvecs, propagators, n_opers, n_coeffs)
    P = eigvecs.conj().transpose(0, 2, 1) @ propagators[:-1]       # (G,d,d)
    out = np.zeros((W, A, d, d), dtype=complex)
    steps = np.empty((G, W, A, d, d), dtype=complex) if cache_intermediates else None
    for g in range(G):
        phase = cexp(omega*t[g])
        integral = first_order_integral(omega, eigvals[g], dt[g])*phase[:, None, None]
        X = Bbar[None, :, g]*integral[:, None]                     # (W,A,d,d)
        step = P[g].conj().T @ X @ P[g]
        if cache_intermediates:
            steps[g] = step                                         # 'noise_operators_step', :611-615
        out += step
    if cache_intermediates:
        return out, dict(noise_operators_step=steps)
    return out


def filter_function(R, which='fidelity'):
    """F[a,b,o] = sum_k conj(R[a,k,o]) R[b,k,o]  (numeric.py:1461-1467)."""
    if which == 'fidelity':
        return np.einsum('ako,bko->abo', R.conj(), R)
    return np.einsum('ako,blo->abklo', R.conj(), R)


def liouville_representation(U, basis):
    """L[i,j] = tr(U^dag C_i U C_j); real part if the basis is Hermitian.

    filter_functions/superoperator.py:51-84 followed by Basis.expand
    (basis.py:650-698).
    """
    U = np.asarray(U)
    basis = np.asarray(basis)
    conj_basis = np.einsum('...ba,ibc,...cd->...iad', U.conj(), basis, U)
    L = np.tensordot(conj_basis, basis, axes=[(-2, -1), (-1, -2)])
    herm = np.allclose(basis, basis.conj().transpose(0, 2, 1),
                       atol=np.finfo(complex).eps*basis.shape[-1]**3, rtol=0)
    return L.real if herm else L


def parse_spectrum(spectrum, omega, idx):
    """Broadcast/validate the spectrum, filter_functions/util.py:214-227."""
    spectrum = np.asarray(spectrum)
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


def infidelity_from_filter_function(F, spectrum, omega, idx, d):
    """(1/2 pi d) int S F dw, filter_functions/numeric.py:323-325, 351-352, 374, 2318-2320."""
    omega = np.asarray(omega, dtype=float)
    idx = np.asarray(idx)
    spectrum = parse_spectrum(spectrum, omega, idx)
    if spectrum.ndim in (1, 2):
        integrand = F[idx, idx, :]*spectrum
    else:
        integrand = F[idx[:, None], idx, :]*spectrum
    return integrate(integrand.real, omega)/(2*np.pi*d)


def noise_operators_from_atomic(phases, B_atomic, propagators):
    """B = B^(0) + sum_g phases[g-1] P_{g-1}^dag B^(g) P_{g-1}, filter_functions/numeric.py:377-453
    (_transform_by_unitary :126-141 is U^dag X U)."""
    B_atomic = np.asarray(B_atomic)
    out = B_atomic[0].copy()
    for g in range(1, len(B_atomic)):
        P = np.asarray(propagators[g - 1])
        out += P.conj().T @ (B_atomic[g]*np.asarray(phases[g - 1])[:, None, None, None]) @ P
    return out


def infidelity_nontraceless(control_matrix, basis, spectrum, omega, idx, d):
    """infidelity() for a basis that is not traceless, filter_functions/numeric.py:2295-2305:
    F_ab = sum_kl R*_ak R_bl (sum_m T_klmm - sum_m T_kmlm)/d, then the usual integral."""
    T = four_element_traces(basis)
    traces_diag = np.einsum('klmm->kl', T) - np.einsum('kmlm->kl', T)
    R = np.asarray(control_matrix)
    F = np.einsum('ako,blo,kl->abo', R.conj(), R, traces_diag)/d
    return infidelity_from_filter_function(F, spectrum, omega, idx, d)


def control_matrix_from_atomic(phases, R_atomic, Q_liouville, which='total'):
    """R = sum_g e^{i w t_{g-1}} R^{(g)} Q^{(g-1)}  (numeric.py:621-704)."""
    G = len(R_atomic)
    steps = np.empty((G,) + R_atomic[0].shape, dtype=complex)
    steps[0] = R_atomic[0]
    for g in range(1, G):
        # (A,N,W) -> contraction over the basis index with Q^{(g-1)} (N,N)
        steps[g] = np.einsum('o,ako,kl->alo', phases[g - 1], R_atomic[g], Q_liouville[g - 1])
    if which == 'correlations':
        return steps
    return steps.sum(axis=0)


def control_matrix_periodic(phases, control_matrix, total_propagator_liouville, repeats):
    """R_G = R_1 sum_{g<G} (e^{i w T} Q_1)^g, the geometric series in closed form with one linear
    solve per frequency: (I - T) S = I - T^G  (numeric.py:886-954, the well-conditioned branch;
    where cond(I - T) >= 1e8 the reference sums the series term by term, as done here too)."""
    L = np.asarray(total_propagator_liouville)
    eye = np.eye(len(L))
    T = np.multiply.outer(np.asarray(phases), L)
    M = eye - T
    good = np.linalg.cond(M) < 1e8
    S = np.empty_like(T)
    S[good] = np.linalg.solve(M[good], eye - np.linalg.matrix_power(T[good], repeats))
    for w in np.flatnonzero(~good):
        term, total = eye.astype(complex), eye.astype(complex)
        for _ in range(repeats - 1):
            term = term @ T[w]
            total = total + term
        S[w] = total
    return (np.asarray(control_matrix).transpose(2, 0, 1) @ S).transpose(1, 2, 0)


# ---------------------------------------------------------------------------------------------
# Decay amplitudes -> cumulant function -> error transfer matrix
# ---------------------------------------------------------------------------------------------
def decay_amplitudes(control_matrix, spectrum, omega, idx, which='total'):
    """Gamma_{ab,kl} = int dw/2pi Re[R*_{ak} S_{ab} R_{bl}], filter_functions/numeric.py:1194-1337
    with the integrand of _get_integrand (:310-374, 'generalized', control matrix given).
    control_matrix: (A, N, W) for which='total', (G, A, N, W) for 'correlations'."""
    omega = np.asarray(omega, dtype=float)
    idx = np.asarray(idx)
    R = np.asarray(control_matrix)
    S = parse_spectrum(spectrum, omega, idx)
    left, right = R.conj()[..., idx, :, :], R[..., idx, :, :]
    if S.ndim in (1, 2):
        sub = 'g...ko,...o,h...lo->gh...klo' if which == 'correlations' else '...ko,...o,...lo->...klo'
    else:
        sub = 'gako,abo,hblo->ghabklo' if which == 'correlations' else 'ako,abo,blo->abklo'
    integrand = np.einsum(sub, left, S, right).real
    return integrate(integrand, omega)/(2*np.pi)


def decay_amplitudes_shard(control_matrix_block, spectrum_block, omega, w_offset, idx):
    """Contribution of the frequency block [w_offset, w_offset + Wb) to decay_amplitudes(...,
    which='total') over the global grid omega: the trapezoid written as a weighted sum,
    sum_i w_i f_i with w_i = (omega_{i+1} - omega_{i-1})/2 (one-sided at the ends)."""
    omega = np.asarray(omega, dtype=float)
    R = np.asarray(control_matrix_block)[np.asarray(idx)]
    Wb = R.shape[-1]
    wgt = np.zeros(len(omega))
    wgt[:-1] += 0.5*np.diff(omega)
    wgt[1:] += 0.5*np.diff(omega)
    S = parse_spectrum(spectrum_block, np.empty(Wb), np.asarray(idx))*wgt[w_offset:w_offset + Wb]
    if S.ndim in (1, 2):
        # sum_o conj(R_ko) S_o R_lo as one matrix product per operator (full-size grids)
        integrand = (R.conj()*S[..., None, :]) @ R.swapaxes(-1, -2)
    else:
        integrand = np.einsum('ako,abo,blo->abkl', R.conj(), S, R)
    return integrand.real/(2*np.pi)


def four_element_traces(basis):
    """T_ijkl = tr(C_i C_j C_k C_l), filter_functions/basis.py:330-348 (dense; small d only)."""
    C = np.asarray(basis)
    P = np.einsum('iab,jbc->ijac', C, C)
    return np.einsum('ijac,klca->ijkl', P, P)


def cumulant_function_dense(Gamma, basis, single_qubit=False):
    """K_{ij} = -1/2 sum_kl Gamma_kl (T_klji - T_kjli - T_kilj + T_kijl),
    filter_functions/numeric.py:1119-1165, 1190 (first order only).  single_qubit: the simplified
    expression the reference takes for d = 2 Pauli/GGM bases (:1119-1141)."""
    Gamma = np.asarray(Gamma)
    N = Gamma.shape[-1]
    if single_qubit:
        K = np.zeros(Gamma.shape, Gamma.dtype)
        mask = np.zeros((N, N), dtype=bool)
        mask[1:, 1:] = ~np.eye(N - 1, dtype=bool)
        K[..., mask] = Gamma[..., mask]
        for i in range(1, N):
            others = [j for j in range(1, N) if j != i]
            K[..., i, i] = -Gamma[..., others, others].sum(axis=-1)
        return K
    T = four_element_traces(basis)
    K = -(np.einsum('...kl,klji->...ij', Gamma, T).real
          - np.einsum('...kl,kjli->...ij', Gamma, T).real
          - np.einsum('...kl,kilj->...ij', Gamma, T).real
          + np.einsum('...kl,kijl->...ij', Gamma, T).real)
    return K*0.5


def cumulant_function(Gamma, basis):
    """The same contraction without the N^4 trace tensor: with D_k = sum_l Gamma_kl C_l
    the cumulant superoperator is
        K(X) = -1/2 sum_k (C_k D_k X - C_k X D_k - D_k X C_k + X D_k C_k),
    and K_ij = tr(C_i K(C_j)).  O(d^6) instead of O(d^8); what the device path implements."""
    C = np.asarray(basis)
    N, d = C.shape[:2]
    Cf = C.reshape(N, d*d)
    D = np.asarray(Gamma) @ Cf                                   # (..., N, d^2)
    M4 = (Cf.T @ D).reshape(D.shape[:-2] + (d, d, d, d))         # [a,b,c,e] = sum_k C_k[a,b] D_k[c,e]
    G1 = np.einsum('...apbq,pb->...aq', M4, np.eye(d))           # sum_k C_k D_k
    G2 = np.einsum('...pqab,bp->...aq', M4, np.eye(d))           # sum_k D_k C_k
    eye = np.eye(d)
    S4 = (np.einsum('...ap,bq->...abpq', G1, eye)                # [p',q',p,q]: G1 X
          + np.einsum('ap,...qb->...abpq', eye, G2)              # X G2
          - np.einsum('...apqb->...abpq', M4)                    # C_k X D_k
          - np.einsum('...qbap->...abpq', M4))                   # D_k X C_k
    S4 = -0.5*S4
    return np.einsum('iba,...abpq,jpq->...ij', C, S4, C).real


def cumulant_function_matrix_form(Gamma, basis):
    """The cumulant superoperator applied to every basis element with plain matrix products -- for dimensions
    where `cumulant_function`'s (d, d, d, d) intermediates and `cumulant_function_dense`'s N^4 trace tensor are
    out of reach (d > 16: N = d^2 > 256).  Same definition (filter_functions/numeric.py:1119-1165, 1190 with the
    four-element traces written out): with D_k = sum_l Gamma_kl C_l,
        K(X) = -1/2 (G1 X + X G2 - sum_k C_k X D_k - sum_k D_k X C_k),  G1 = sum_k C_k D_k,  G2 = sum_k D_k C_k,
        K_ij = Re tr(C_i K(C_j)).
    Pinned to `cumulant_function` and to the reference's fixtures at small d in tests/test_oracle_golden.py."""
    C = np.asarray(basis)
    N, d = C.shape[:2]
    Gamma = np.asarray(Gamma)
    lead = Gamma.shape[:-2]
    out = np.empty(lead + (N, N))
    for index in np.ndindex(*lead):
        D = np.tensordot(Gamma[index], C, axes=(1, 0))                 # (N, d, d): D_k
        G1 = np.einsum('kab,kbc->ac', C, D)
        G2 = np.einsum('kab,kbc->ac', D, C)
        KX = np.matmul(G1, C) + np.matmul(C, G2)                       # (N, d, d): G1 C_j + C_j G2
        # sum_k C_k X D_k for every X = C_j: (k a b)(j b c)(k c e) -> (j a e), as two batched products
        left = np.tensordot(C, C, axes=(2, 1))                         # [k, a, j, c] = (C_k C_j)[a, c]
        KX -= np.einsum('kajc,kce->jae', left, D, optimize=True)
        left = np.tensordot(D, C, axes=(2, 1))                         # [k, a, j, c] = (D_k C_j)[a, c]
        KX -= np.einsum('kajc,kce->jae', left, C, optimize=True)
        out[index] = -0.5*np.einsum('iab,jba->ij', C, KX, optimize=True).real
    return out


def error_transfer_matrix(K):
    """exp(sum over all but the last two axes of K), filter_functions/numeric.py:2049-2053."""
    from scipy.linalg import expm
    K = np.asarray(K)
    return expm(K.sum(axis=tuple(range(K.ndim - 2))))


# ---------------------------------------------------------------------------------------------
# Second order: nested Magnus integral -> second-order filter function -> frequency shifts
# (SURVEY 8f.3 consumer of the step caches)
# ---------------------------------------------------------------------------------------------
def _frac(x, dt):
    """(e^{i x dt} - 1)/x, and its limit i dt where x == 0 exactly
    (filter_functions/numeric.py:229-235 via util.cexpm1)."""
    x = np.asarray(x, dtype=float)
    out = np.full(x.shape, 1j*dt, dtype=complex)
    nz = x != 0
    out[nz] = cexpm1(x[nz]*dt)/x[nz]
    return out


def second_order_integral(omega, eigvals_g, dt_g):
    """I[o,i,j,m,n], the nested time integral of the second-order Magnus term over one segment,
    filter_functions/numeric.py:170-256:

        b = w + W_mn != 0:             (f(W_ij - w) - f(W_ij + W_mn)) / b,  f(x) = (e^{i x dt}-1)/x
        b == 0, a = W_ij - w != 0:     (f(a) - i dt e^{i a dt}) / a
        b == 0, a == 0:                dt^2/2

    with W_mn = D_m - D_n.  The reference evaluates the b == 0 rows only for w == 0 exactly
    (:241-255; for w != 0 its masked ufuncs leave the buffer untouched there); this restatement
    applies the documented formula (:186-194) to every b == 0 entry, which coincides at w == 0."""
    E = np.asarray(omega, dtype=float)
    dE = np.subtract.outer(eigvals_g, eigvals_g)
    a = np.add.outer(-E, dE)                                    # (W,d,d)  W_ij - w
    b = np.add.outer(E, dE)                                     # (W,d,d)  w + W_mn
    c = np.add.outer(dE, dE)                                    # (d,d,d,d)
    f1 = _frac(a, dt_g)
    f2 = _frac(c, dt_g)
    W, d = len(E), len(eigvals_g)
    bb = np.broadcast_to(b[:, None, None], (W, d, d, d, d))
    aa = np.broadcast_to(a[:, :, :, None, None], (W, d, d, d, d))
    f1b = np.broadcast_to(f1[:, :, :, None, None], (W, d, d, d, d))
    out = np.empty((W, d, d, d, d), dtype=complex)
    gen = bb != 0
    out[gen] = ((f1b - f2[None])[gen])/bb[gen]
    lim = ~gen & (aa != 0)
    out[lim] = (f1b[lim] - 1j*dt_g*cexp(aa[lim]*dt_g))/aa[lim]
    out[~gen & (aa == 0)] = dt_g**2/2
    return out


def second_order_filter_function(eigvals, eigvecs, propagators, omega, basis, n_opers, n_coeffs,
                                 dt):
    """F2[a,b,k,l,o] = sum_g [ conj(G^(g)_{ak}) sum_{g'<g} G^(g')_{bl}
                               + sum_ijmn N^(g)_{ak,ij} I^(g)_{o,ijmn} N^(g)_{bl,mn} ],
    N^(g)_{ak,ij} = Bbar^(g)_{a,ij} Cbar^(g)_{k,ji}, G^(g) the control-matrix step;
    filter_functions/numeric.py:1470-1699 (from scratch, no caches)."""
    dt = np.asarray(dt, dtype=float)
    omega = np.asarray(omega, dtype=float)
    basis = np.asarray(basis)
    G, d = eigvals.shape
    A, N, W = len(n_opers), len(basis), len(omega)
    t = np.concatenate(([0.0], dt.cumsum()))
    QdV, Bbar = _prologue(eigvals, eigvecs, propagators, n_opers, n_coeffs)
    F2 = np.zeros((A, A, N, N, W), dtype=complex)
    cumulative = np.zeros((A, N, W), dtype=complex)
    for g in range(G):
        Cbar = QdV[g].conj().T @ basis @ QdV[g]                              # numeric.py:1644
        NB = (Bbar[:, g, None, :, :]*Cbar.transpose(0, 2, 1)[None]).reshape(A*N, d*d)  # :1632
        I1 = first_order_integral(omega, eigvals[g], dt[g]).reshape(W, d*d)
        step = (NB @ (I1*cexp(omega*t[g])[:, None]).T).reshape(A, N, W)      # numeric.py:1649
        I2 = second_order_integral(omega, eigvals[g], dt[g]).reshape(W, d*d, d*d)
        inc = np.einsum('pi,oim,qm->pqo', NB, I2, NB).reshape(A, N, A, N, W)  # numeric.py:1625
        F2 += inc.transpose(0, 2, 1, 3, 4)
        if g > 0:
            F2 += step.conj()[:, None, :, None]*cumulative[None, :, None]    # numeric.py:1679
        cumulative += step
    return F2


def second_order_from_atomic(F2_atomic, control_matrix_step, propagators_liouville):
    """Concatenation rule of the second-order filter function,
    filter_functions/numeric.py:1702-1818, in its rotated form: the complete steps of pulse g are
    rotated by the Liouville matrix of the preceding propagator on both basis indices (:1809-1812,
    'pk,abpqo,ql->abklo'); the incomplete steps, which the reference re-evaluates with the
    propagated eigenvectors (:1787-1790, :1815-1817), are bilinear in the basis elements and hence
    rotate the same way; and the rank-one term couples pulse g's summand of the control matrix
    with the cumulative earlier ones (:1804-1806).
    F2_atomic (G,A,A,N,N,W); control_matrix_step (G,A,N,W); propagators_liouville (G-1,N,N)."""
    F2_atomic = np.asarray(F2_atomic)
    Rs = np.asarray(control_matrix_step)
    out = F2_atomic[0].copy()
    cum = Rs[0].copy()
    for g in range(1, len(Rs)):
        L = np.asarray(propagators_liouville[g - 1])
        out += np.einsum('pk,abpqo,ql->abklo', L, F2_atomic[g], L)
        out += Rs[g].conj()[:, None, :, None]*cum[None, :, None]
        cum += Rs[g]
    return out


def frequency_shifts(F2, spectrum, omega, idx):
    """Delta_{ab,kl} = int dw/2pi Re[S_ab F2_{ab,kl}], filter_functions/numeric.py:1340-1410 with
    the 'generalized' filter-function branch of _get_integrand (:318-329, :351-354)."""
    omega = np.asarray(omega, dtype=float)
    idx = np.asarray(idx)
    S = parse_spectrum(spectrum, omega, idx)
    if S.ndim in (1, 2):
        integrand = F2[idx, idx]*(S[:, None, None, :] if S.ndim == 2 else S)
    else:
        integrand = F2[idx[:, None], idx]*S[:, :, None, None, :]
    return integrate(integrand.real, omega)/(2*np.pi)


def frequency_shifts_shard(F2_block, spectrum_block, omega, w_offset, idx):
    """Contribution of the frequency block [w_offset, w_offset + Wb) to frequency_shifts over the
    global grid omega (trapezoid as a weighted sum, like decay_amplitudes_shard)."""
    omega = np.asarray(omega, dtype=float)
    idx = np.asarray(idx)
    Wb = F2_block.shape[-1]
    wgt = np.zeros(len(omega))
    wgt[:-1] += 0.5*np.diff(omega)
    wgt[1:] += 0.5*np.diff(omega)
    S = parse_spectrum(spectrum_block, np.empty(Wb), idx)*wgt[w_offset:w_offset + Wb]
    if S.ndim in (1, 2):
        integrand = F2_block[idx, idx]*(S[:, None, None, :] if S.ndim == 2 else S)
    else:
        integrand = F2_block[idx[:, None], idx]*S[:, :, None, None, :]
    return integrand.real.sum(axis=-1)/(2*np.pi)


def cumulant_second_order_dense(Delta, basis, single_qubit=False):
    """The frequency-shift contribution to K: -1/2 sum_kl Delta_kl (T_klji - T_lkji - T_klij +
    T_lkij), filter_functions/numeric.py:1166-1190; single qubit: -(Delta - Delta^T) on the
    traceless block (:1139-1141)."""
    Delta = np.asarray(Delta)
    if single_qubit:
        K = np.zeros(Delta.shape, Delta.dtype)
        K[..., 1:, 1:] = -Delta[..., 1:, 1:] + Delta[..., 1:, 1:].swapaxes(-1, -2)
        return K
    T = four_element_traces(basis)
    K = -(np.einsum('...kl,klji->...ij', Delta, T).real
          - np.einsum('...kl,lkji->...ij', Delta, T).real
          - np.einsum('...kl,klij->...ij', Delta, T).real
          + np.einsum('...kl,lkij->...ij', Delta, T).real)
    return K*0.5


def cumulant_second_order(Delta, basis):
    """The same without the trace tensor: with X = sum_kl (Delta_kl - Delta_lk) C_k C_l the
    contribution is K_ij = -1/2 Re tr(C_i [X, C_j]), a commutator with the effective Hamiltonian
    of the frequency shifts.  What the device path implements."""
    C = np.asarray(basis)
    Delta = np.asarray(Delta)
    Asym = Delta - Delta.swapaxes(-1, -2)
    X = np.einsum('...kl,kab,lbc->...ac', Asym, C, C)
    comm = np.einsum('...ab,jbc->...jac', X, C) - np.einsum('jab,...bc->...jac', C, X)
    return -0.5*np.einsum('iba,...jab->...ij', C, comm).real


# ---------------------------------------------------------------------------------------------
# Gradient of the filter function / infidelity with respect to the control amplitudes
# (filter_functions/gradient.py; consumer of the step caches)
# ---------------------------------------------------------------------------------------------
def _nested_exponential_integral(x, I1, dt):
    """int_0^dt tau e^{i x tau} dtau = (dt e^{i x dt} - I1(x))/(i x), dt^2/2 at x == 0: the
    Omega_pq == 0 case of the derivative integral, filter_functions/gradient.py:84-94."""
    out = np.full(np.shape(x), dt**2/2, dtype=complex)
    nz = x != 0
    out[nz] = (dt*cexp(x[nz]*dt) - I1[nz])/(1j*x[nz])
    return out


def filter_function_derivative(eigvals, eigvecs, propagators, omega, basis, n_opers, n_coeffs,
                               c_opers, dt, n_coeffs_deriv=None):
    """dF_a(w)/du_h(t_s), shape (n_nops, n_dt, n_ctrl, n_omega): what
    PulseSequence.get_filter_function_derivative returns (filter_functions/pulse_sequence.py:977-1054
    = gradient.calculate_filter_function_derivative(:526-556) of
    gradient.calculate_derivative_of_control_matrix_from_scratch(:384-523)).

    The reference builds the derivative of the control matrix, (n_ctrl, W, G, A, d^2), from
      (I)  the derivative of segment s's own contribution (_control_matrix_at_timestep_derivative,
           :200-381, with the nested integral _derivative_integral :69-108), and
      (II) the derivative of every later Liouville propagator (_liouville_derivative :111-197,
           contracted over all pairs of segments at :520),
    and contracts it with conj(R).  Restated here in Hilbert space, with Y_a(w) the interaction
    picture noise operator (R_ak = tr(Y_a C_k)):
      (I)  2 Re tr(Y_a^dag Y'),  Y' = -i e^{i w t_s} T^dag G^T T,  T = V_s^dag Q_s, and
           G_xy = sum_n Bbar_yn Abar_nx J(w; W_yn, W_nx) - sum_q Abar_yq Bbar_qx J(w; W_qx, W_yq),
           J(w; a, b) = int_0^dt dtau e^{i(w+a)tau} int_0^tau dtau' e^{i b tau'};
      (II) a later propagator changes by Q_g -> Q_g E with the SAME generator
           E = -i T^dag (Abar o I1(0)) T for every g > s, so that the sum over later segments
           collapses to -2 Re tr(E [Y_a^dag, Ycum_{s,a}]) with Ycum the steps up to and including s
           (the part proportional to the total Y_a drops out of the real part);
      and the explicit dependence of the noise sensitivities, (n'_ah / n_a) 2 Re tr(Y_a^dag Ystep_s)
      (:376-379)."""
    dt = np.asarray(dt, dtype=float)
    omega = np.asarray(omega, dtype=float)
    G, d = eigvals.shape
    A, H, W = len(n_opers), len(c_opers), len(omega)
    t = np.concatenate(([0.0], dt.cumsum()))
    QdV, Bbar = _prologue(eigvals, eigvecs, propagators, n_opers, n_coeffs)
    _, Abar = _prologue(eigvals, eigvecs, propagators, c_opers, np.ones((H, G)))
    n_coeffs = np.asarray(n_coeffs, dtype=float)
    # Hilbert-space steps Ystep[g,a,w] = e^{i w t_g} T^dag (Bbar o I1) T
    Ystep = np.empty((G, A, W, d, d), dtype=complex)
    I1_all = np.empty((G, W, d, d), dtype=complex)
    for g in range(G):
        T = QdV[g].conj().T
        I1_all[g] = first_order_integral(omega, eigvals[g], dt[g])
        inner = Bbar[:, g, None]*I1_all[g][None]*cexp(omega*t[g])[None, :, None, None]
        Ystep[g] = T.conj().T @ inner @ T
    Ycum = Ystep.cumsum(axis=0)
    Ytot = Ycum[-1]
    Ytot_dag = Ytot.conj().swapaxes(-1, -2)
    out = np.empty((A, G, H, W))
    for s in range(G):
        T = QdV[s].conj().T
        dE = np.subtract.outer(eigvals[s], eigvals[s])
        I0 = first_order_integral(np.zeros(1), eigvals[s], dt[s])[0]
        I1 = I1_all[s]                                        # (W,d,d): I1(w + W_mn)
        x = omega[:, None, None] + dE[None]
        Jd = _nested_exponential_integral(x, I1, dt[s])       # b == 0 case at a = W_mn
        # J1[w,m,n,q] = J(w; W_mn, W_nq),  J2[w,p,q,n] = J(w; W_qn, W_pq)
        J1 = np.empty((W, d, d, d), dtype=complex)
        J2 = np.empty((W, d, d, d), dtype=complex)
        for m in range(d):
            for n in range(d):
                for q in range(d):
                    J1[:, m, n, q] = Jd[:, m, n] if dE[n, q] == 0 else \
                        (I1[:, m, q] - I1[:, m, n])/(1j*dE[n, q])
        for p in range(d):
            for q in range(d):
                for n in range(d):
                    J2[:, p, q, n] = Jd[:, q, n] if dE[p, q] == 0 else \
                        (I1[:, p, n] - I1[:, q, n])/(1j*dE[p, q])
        phase = cexp(omega*t[s])
        Wa = T[None, None] @ Ytot_dag @ T.conj().T[None, None]                    # (A,W,d,d)
        comm = Ytot_dag @ Ycum[s] - Ycum[s] @ Ytot_dag                           # (A,W,d,d)
        for h in range(H):
            E = -1j*(T.conj().T @ (Abar[h, s]*I0) @ T)
            second = -2*np.einsum('xy,awyx->aw', E, comm).real
            for a in range(A):
                Yq = np.einsum('mn,nq,wmnq->wqm', Bbar[a, s], Abar[h, s], J1)
                Zn = np.einsum('pq,qn,wpqn->wnp', Abar[h, s], Bbar[a, s], J2)
                first = 2*(-1j*phase*np.einsum('wxy,wxy->w', Wa[a], Yq - Zn)).real
                out[a, s, h] = first + second[a]
                if n_coeffs_deriv is not None:
                    ratio = np.asarray(n_coeffs_deriv)[a, h, s]/n_coeffs[a, s]
                    out[a, s, h] += ratio*2*np.einsum('wyx,wxy->w', Ytot_dag[a], Ystep[s, a]).real
    return out


def control_matrix_derivative(eigvals, eigvecs, propagators, omega, basis, n_opers, n_coeffs,
                              c_opers, dt, n_coeffs_deriv=None):
    """d R_ak(w) / d u_h(t_s), shape (n_ctrl, n_omega, n_dt, n_nops, n_basis): what
    gradient.calculate_derivative_of_control_matrix_from_scratch returns
    (filter_functions/gradient.py:384-523).  Hilbert-space restatement with the quantities of
    filter_function_derivative above: R_ak = tr(Y_a C_k) and
        dY_a/du_h(t_s) = Y' + [Ytot_a - Ycum_{s,a}, E_hs] + (n'_ahs / n_as) Ystep_{s,a},
    Y' = -i e^{i w t_s} T^dag G^T T (segment s's own contribution, :200-381), the commutator with
    the anti-Hermitian generator E_hs the change of all later propagators (Q_g -> Q_g E for g > s;
    _liouville_derivative :111-197 and the pairwise contraction :520), the last term :376-379."""
    dt = np.asarray(dt, dtype=float)
    omega = np.asarray(omega, dtype=float)
    basis = np.asarray(basis)
    G, d = eigvals.shape
    A, H, W = len(n_opers), len(c_opers), len(omega)
    t = np.concatenate(([0.0], dt.cumsum()))
    QdV, Bbar = _prologue(eigvals, eigvecs, propagators, n_opers, n_coeffs)
    _, Abar = _prologue(eigvals, eigvecs, propagators, c_opers, np.ones((H, G)))
    n_coeffs = np.asarray(n_coeffs, dtype=float)
    Ystep = np.empty((G, A, W, d, d), dtype=complex)
    I1_all = np.empty((G, W, d, d), dtype=complex)
    for g in range(G):
        T = QdV[g].conj().T
        I1_all[g] = first_order_integral(omega, eigvals[g], dt[g])
        inner = Bbar[:, g, None]*I1_all[g][None]*cexp(omega*t[g])[None, :, None, None]
        Ystep[g] = T.conj().T @ inner @ T
    Ycum = Ystep.cumsum(axis=0)
    out = np.empty((H, W, G, A, len(basis)), dtype=complex)
    for s in range(G):
        T = QdV[s].conj().T
        dE = np.subtract.outer(eigvals[s], eigvals[s])
        I0 = first_order_integral(np.zeros(1), eigvals[s], dt[s])[0]
        I1 = I1_all[s]
        Jd = _nested_exponential_integral(omega[:, None, None] + dE[None], I1, dt[s])
        J1 = np.empty((W, d, d, d), dtype=complex)
        J2 = np.empty((W, d, d, d), dtype=complex)
        for m in range(d):
            for n in range(d):
                for q in range(d):
                    J1[:, m, n, q] = Jd[:, m, n] if dE[n, q] == 0 else \
                        (I1[:, m, q] - I1[:, m, n])/(1j*dE[n, q])
                    J2[:, m, n, q] = Jd[:, n, q] if dE[m, n] == 0 else \
                        (I1[:, m, q] - I1[:, n, q])/(1j*dE[m, n])
        phase = cexp(omega*t[s])
        rest = Ycum[-1] - Ycum[s]                                               # (A,W,d,d)
        for h in range(H):
            E = -1j*(T.conj().T @ (Abar[h, s]*I0) @ T)
            later = rest @ E - E @ rest
            for a in range(A):
                Gm = (np.einsum('mn,nq,wmnq->wqm', Bbar[a, s], Abar[h, s], J1)
                      - np.einsum('pq,qn,wpqn->wnp', Abar[h, s], Bbar[a, s], J2))
                dY = -1j*phase[:, None, None]*(T.conj().T @ Gm.swapaxes(-1, -2) @ T) + later[a]
                if n_coeffs_deriv is not None:
                    dY = dY + np.asarray(n_coeffs_deriv)[a, h, s]/n_coeffs[a, s]*Ystep[s, a]
                out[h, :, s, a] = np.einsum('wij,kji->wk', dY, basis)
    return out


def filter_function_derivative_from_control_matrix(ctrlmat, ctrlmat_deriv):
    """2 Re sum_k conj(R_ak) dR_ak: (A, N, W), (H, W, G, A, N) -> (A, G, H, W)
    (gradient.calculate_filter_function_derivative, filter_functions/gradient.py:526-556)."""
    return 2*np.einsum('ako,hotak->atho', np.conj(ctrlmat), ctrlmat_deriv).real


def infidelity_derivative(dF, spectrum, omega, d):
    """int dw/(2 pi d) S dF, filter_functions/gradient.py:667-676."""
    omega = np.asarray(omega, dtype=float)
    S = parse_spectrum(spectrum, omega, np.arange(dF.shape[0]))
    integrand = np.einsum('...o,...tho->...tho', S, dF)
    return integrate(integrand, omega)/(2*np.pi*d)


def infidelity_derivative_shard(dF_block, spectrum_block, omega, w_offset, d):
    """Contribution of the frequency block [w_offset, w_offset + Wb) to infidelity_derivative over
    the global grid omega (trapezoid as a weighted sum, like decay_amplitudes_shard)."""
    omega = np.asarray(omega, dtype=float)
    Wb = dF_block.shape[-1]
    wgt = np.zeros(len(omega))
    wgt[:-1] += 0.5*np.diff(omega)
    wgt[1:] += 0.5*np.diff(omega)
    S = parse_spectrum(spectrum_block, np.empty(Wb), np.arange(dF_block.shape[0]))
    S = S*wgt[w_offset:w_offset + Wb]
    return np.einsum('...o,...tho->...th', S, dF_block)/(2*np.pi*d)
