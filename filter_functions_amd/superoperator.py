"""Superoperator helpers: the Liouville representation of a unitary (on the path) and the small
analysis utilities around it.

``liouville_representation`` mirrors ``filter_functions/superoperator.py:51-84``; the
d^2 x 2d^2 x d^2 real contraction behind it runs on the FP64 matrix cores
(``v_mfma_f64_16x16x4_f64``, csrc/liouville.hip).  ``liouville_to_choi``, ``liouville_is_CP`` and
``liouville_is_cCP`` (superoperator.py:87-266) are host-side diagnostics on single d^2 x d^2
matrices (an index re-arrangement and one Hermitian eigenvalue problem), like the reference's.
"""
import numpy as np

from . import _lib
from ._lib import as_c128, check, ptr

__all__ = ['liouville_representation', 'liouville_to_choi', 'liouville_is_CP', 'liouville_is_cCP']


def liouville_representation(U, basis):
    r"""Liouville representation :math:`\mathcal U_{ij} = \mathrm{tr}(C_i U C_j U^\dagger)` of
    the unitary (or stack of unitaries) *U* with respect to *basis*.

    U: (..., d, d); basis: (n_basis, d, d).  Returns (..., n_basis, n_basis), real if the
    basis is Hermitian (like the reference, which casts iff ``basis.isherm``), else complex.
    """
    U = as_c128(U)
    barr = as_c128(np.asarray(basis))
    if U.ndim < 2 or U.shape[-1] != U.shape[-2]:
        raise ValueError(f'Expected U of shape (..., d, d), not {U.shape}.')
    d = U.shape[-1]
    if barr.ndim != 3 or barr.shape[1:] != (d, d):
        raise ValueError(f'Expected basis of shape (n_basis, {d}, {d}), not {barr.shape}.')
    if not 2 <= d <= _lib.MAX_D:
        raise ValueError(f'Hilbert space dimension d={d} unsupported: need 2 <= d <= {_lib.MAX_D}.')
    hermitian = getattr(basis, 'isherm', None)
    if hermitian is None:
        hermitian = np.allclose(barr, barr.conj().swapaxes(-1, -2),
                                atol=np.finfo(complex).eps*d**3, rtol=0)
    N = len(barr)
    lead = U.shape[:-2]
    batch = int(np.prod(lead)) if lead else 1
    out = np.empty(lead + (N, N), dtype=np.float64 if hermitian else np.complex128)
    if batch:
        check(_lib.load().ffk_liouville(ptr(U), batch, d, ptr(barr), N, int(bool(hermitian)),
                                        ptr(out)))
    return out


def liouville_to_choi(superoperator, basis):
    r"""Choi matrix of a superoperator given in Liouville representation with respect to *basis*
    (reference superoperator.py:87-130):
    :math:`\mathrm{choi}(\mathcal S) = \sum_{ij}\mathcal S_{ij}\,C_j^T\otimes C_i`, shape like
    *superoperator*, (..., d**2, d**2)."""
    S = np.asarray(superoperator)
    C = np.asarray(basis)
    left = np.tensordot(S, C, axes=[-1, 0])                 # (..., i, b, a) = sum_j S_ij C_j[b, a]
    choi = np.einsum('...iba,icd->...acbd', left, C)
    return choi.reshape(S.shape)


def _psd(matrix, atol):
    D, V = np.linalg.eigh(matrix)
    return (D >= -atol).all(axis=-1), (D, V)


def liouville_is_CP(superoperator, basis, return_eig=False, atol=None):
    """Is the superoperator completely positive, i.e. its Choi matrix positive semidefinite
    (reference superoperator.py:133-193)?  Returns a bool (array if broadcast) and, with
    *return_eig*, the eigenvalues and eigenvectors of the Choi matrix."""
    atol = atol or getattr(basis, '_atol', np.finfo(float).eps*np.shape(basis)[-1]**3)
    CP, eig = _psd(liouville_to_choi(superoperator, basis), atol)
    return (CP, eig) if return_eig else CP


def liouville_is_cCP(superoperator, basis, return_eig=False, atol=None):
    r"""Is the superoperator conditionally completely positive, i.e. its Choi matrix projected on
    the complement of the maximally entangled state, :math:`Q\,\mathrm{choi}(\mathcal S)\,Q` with
    :math:`Q = \mathbb I - |\Omega\rangle\langle\Omega|`, positive semidefinite (reference
    superoperator.py:196-266)?"""
    atol = atol or getattr(basis, '_atol', np.finfo(float).eps*np.shape(basis)[-1]**3)
    d2 = np.shape(superoperator)[-1]
    d = int(round(np.sqrt(d2)))
    omega = np.zeros(d2)
    omega[::d + 1] = 1/np.sqrt(d)
    Q = np.eye(d2) - np.multiply.outer(omega, omega)
    cCP, eig = _psd(Q @ liouville_to_choi(superoperator, basis) @ Q, atol)
    return (cCP, eig) if return_eig else cCP
