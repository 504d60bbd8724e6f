"""Superoperator helpers on the path: the Liouville representation of a unitary.

``liouville_representation`` mirrors ``filter_functions/superoperator.py:51-84``; the
d^2 x 2d^2 x d^2 real contraction behind it runs on the FP64 matrix cores
(``v_mfma_f64_16x16x4_f64``, csrc/liouville.hip).
"""
import numpy as np

from . import _lib
from ._lib import as_c128, check, ptr

__all__ = ['liouville_representation']


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
