"""Operator bases: construction and the element-ordering contract of the path.

Mirrors ``filter_functions/basis.py`` for what the hot path needs: the ``Basis``
ndarray subclass with its property checks, the Pauli and generalised Gell-Mann
factories (element ORDER is part of the interface: it fixes the ``k`` axis of the
control matrix and is pinned bit-exactly against the reference by
``tests/test_basis.py``), basis expansion and the two Pauli index maps.
Host-side NumPy on arrays of at most a few hundred KB; no kernel involved.
"""
from functools import cached_property
from itertools import product

from warnings import warn

import numpy as np

from . import util

__all__ = ['Basis', 'expand', 'ggm_expand', 'normalize', 'equivalent_pauli_basis_elements',
           'remap_pauli_basis_elements']


class Basis(np.ndarray):
    """(n, d, d) stack of operator-basis elements (reference basis.py:58-391).

    ``btype`` names the family ('Pauli', 'GGM', 'Custom'), ``labels`` the elements and ``d``
    the Hilbert-space dimension.  ``A == B`` compares with ``allclose``; ``.T``/``.H``
    act element-wise on the last two axes.
    """

    def __new__(cls, basis_array, traceless=None, btype=None, labels=None):
        arr = np.array(basis_array, dtype=complex)  # copies
        if arr.ndim == 2:
            arr = arr[None]
        if arr.ndim != 3 or arr.shape[-1] != arr.shape[-2]:
            raise ValueError('Expected an array of square matrices of shape (n, d, d), '
                             f'not {arr.shape}.')
        if arr.shape[0] > arr.shape[-1]**2:
            raise ValueError('Too many basis elements: at most d**2 are linearly independent.')
        basis = arr.view(cls)
        basis.btype = btype or 'Custom'
        basis.d = arr.shape[-1]
        if labels is not None and len(labels) != len(arr):
            raise ValueError(f'Got {len(labels)} labels for {len(arr)} basis elements.')
        basis.labels = list(labels) if labels is not None else [f'$C_{{{i}}}$'
                                                                for i in range(len(arr))]
        if traceless and not basis.istraceless:
            raise ValueError('The basis elements are not traceless (up to an identity element) '
                             'but a traceless basis was requested!')
        return basis

    def __array_finalize__(self, obj):
        if obj is None:
            return
        self.btype = getattr(obj, 'btype', 'Custom')
        self.labels = getattr(obj, 'labels', [f'$C_{{{i}}}$' for i in range(len(obj))]
                              if np.ndim(obj) else [])
        self.d = getattr(obj, 'd', np.shape(obj)[-1] if np.ndim(obj) else 0)
        self._eps = np.finfo(complex).eps
        self._atol = self._eps*self.d**3
        self._rtol = 0

    def __reduce__(self):
        # ndarray's pickle state plus the attributes of this subclass
        reconstruct, args, state = super().__reduce__()
        return reconstruct, args, state + ({'btype': self.btype, 'labels': self.labels, 'd': self.d},)

    def __setstate__(self, state):
        super().__setstate__(state[:-1])
        self.btype, self.labels, self.d = state[-1]['btype'], state[-1]['labels'], state[-1]['d']
        self._eps = np.finfo(complex).eps
        self._atol = self._eps*self.d**3
        self._rtol = 0

    def __array_wrap__(self, arr, context=None, return_scalar=False):
        # ufunc reductions to 0-d should give scalars, not 0-d Basis objects
        if np.ndim(arr) == 0:
            return arr[()]
        return np.ndarray.__array_wrap__(self, arr, context, False)

    def __eq__(self, other):
        if not hasattr(other, 'shape'):
            return np.equal(self, other)          # scalars and the like: elementwise
        if other.shape != self.shape:
            return False
        return np.allclose(self.view(np.ndarray), np.asarray(other), atol=self._atol,
                           rtol=self._rtol)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None

    def __contains__(self, item):
        item = np.asarray(item)
        return bool(np.isclose(item, self.view(np.ndarray), rtol=self._rtol,
                               atol=self._atol).all(axis=(1, 2)).any())

    def _invalidate_cached_properties(self):
        for attr in ('isherm', 'isnorm', 'isorthogonal', 'isorthonorm', 'istraceless',
                     'iscomplete', 'four_element_traces'):
            self.__dict__.pop(attr, None)

    @cached_property
    def isherm(self):
        return bool(self.H == self)

    @cached_property
    def isnorm(self):
        return bool(normalize(self) == self)

    @cached_property
    def isorthogonal(self):
        if self.ndim == 2 or len(self) == 1:
            return True
        flat = self.view(np.ndarray).reshape(len(self), -1)
        gram = flat.conj() @ flat.T
        off = gram[~np.identity(len(self), dtype=bool)]
        return bool(np.allclose(off, 0, atol=self._eps*(self.d**2)**3, rtol=self._rtol))

    @cached_property
    def isorthonorm(self):
        return self.isorthogonal and self.isnorm

    @cached_property
    def istraceless(self):
        """True if every element is traceless, except possibly one that is ~ identity."""
        arr = self.view(np.ndarray)
        trace = np.einsum('...jj', arr)
        atol = self._eps*self.d**2
        trace = np.where(np.abs(trace) <= atol, 0, trace)
        nonzero = np.atleast_1d(trace).nonzero()[0]
        if nonzero.size == 0:
            return True
        if nonzero.size == 1:
            elem = arr[nonzero[0]] if arr.ndim == 3 else arr
            offdiag = elem[~np.eye(self.d, dtype=bool)]
            return bool((np.diag(elem) == elem[0, 0]).all() and not offdiag.any())
        return False

    @cached_property
    def iscomplete(self):
        flat = self.view(np.ndarray).reshape(len(self), -1)
        return bool(np.linalg.matrix_rank(flat) == self.d**2)

    @cached_property
    def four_element_traces(self):
        """``T_ijkl = tr(C_i C_j C_k C_l)`` (reference basis.py:330-348) as a dense array.  The
        device code never forms this ``N**4`` tensor (see ``numeric.calculate_cumulant_function``);
        the property exists for user code and is limited to small bases."""
        arr = self.view(np.ndarray)
        if arr.ndim != 3:
            raise ValueError('four_element_traces needs a basis of shape (N, d, d).')
        if 16*float(len(arr))**4 > 2**31:
            raise MemoryError(f'four_element_traces of {len(arr)} elements would need '
                              f'{16*float(len(arr))**4/2**30:.0f} GiB as a dense array.')
        pair = np.einsum('iab,jbc->ijac', arr, arr)
        return np.einsum('ijac,klca->ijkl', pair, pair)

    @property
    def H(self):
        return self.T.conj()

    @property
    def T(self):
        return self.swapaxes(-1, -2) if self.ndim >= 2 else self

    def expand(self, M, hermitian=False, traceless=False, tidyup=False):
        """Expansion coefficients of *M* in this basis (reference basis.py:350-371)."""
        if self.btype == 'GGM' and self.iscomplete:
            return ggm_expand(M, traceless, hermitian, tidyup)
        return expand(M, self, self.isnorm, hermitian, tidyup)

    def normalize(self, copy=False):
        if not copy:
            self /= _norm(self)
            self._invalidate_cached_properties()
            return None
        return normalize(self)

    def tidyup(self, eps_scale=None):
        atol = self._atol if eps_scale is None else self._eps*eps_scale
        arr = self.view(np.ndarray)
        arr.real[np.abs(arr.real) <= atol] = 0
        arr.imag[np.abs(arr.imag) <= atol] = 0
        self._invalidate_cached_properties()

    # ---- factories ---------------------------------------------------------------------
    @classmethod
    def pauli(cls, n):
        """n-qubit Pauli basis {I,X,Y,Z}^n / sqrt(2^n) in ``np.indices((4,)*n)`` order, i.e.
        last qubit fastest; labels ``product('IXYZ', repeat=n)`` (reference basis.py:393-426)."""
        d = 2**n
        sigma = np.empty((4**n, d, d), dtype=complex)
        for flat, combo in enumerate(product(range(4), repeat=n)):
            sigma[flat] = util.tensor(*util.paulis[list(combo)])
        sigma /= np.sqrt(2**n)
        return cls(sigma, btype='Pauli', labels=[''.join(t) for t in product('IXYZ', repeat=n)])

    @classmethod
    def ggm(cls, d):
        """Generalised Gell-Mann basis (reference basis.py:428-489): identity/sqrt(d); the
        n_sym = d(d-1)/2 symmetric elements over the strict upper triangle enumerated row-major;
        the n_sym antisymmetric ones (-i at (j,k), +i at (k,j)); the d-1 diagonal ones."""
        n_sym = d*(d - 1)//2
        rows, cols = np.triu_indices(d, k=1)      # row-major (j < k) enumeration
        lam = np.zeros((d*d, d, d), dtype=complex)
        lam[0] = np.eye(d)/np.sqrt(d)
        s = 1/np.sqrt(2)
        pos = np.arange(1, n_sym + 1)
        lam[pos, rows, cols] = s
        lam[pos, cols, rows] = s
        lam[pos + n_sym, rows, cols] = -1j*s
        lam[pos + n_sym, cols, rows] = 1j*s
        for l in range(1, d):
            diag = np.zeros(d, dtype=complex)
            diag[:l] = 1
            diag[l] = -l
            # complex / real in complex arithmetic, like the reference's in-place `/=` on the
            # complex array (basis.py:484-486): -l/sqrt(l(l+1)) can differ by 1 ulp otherwise
            diag /= np.sqrt(l*(l + 1))
            lam[2*n_sym + l, range(d), range(d)] = diag
        return cls(lam, btype='GGM', labels=[rf'$\Lambda_{{{i}}}$' for i in range(d*d)])

    @classmethod
    def from_partial(cls, partial_basis_array, traceless=None, btype=None, labels=None):
        """Complete a set of orthogonal operators to a full orthonormal basis of the d x d matrices
        (reference basis.py:492-620): the given elements come first (after the identity if the
        basis is traceless), the remainder spans their orthogonal complement.  The complement is
        found in the coefficient space of the generalised Gell-Mann basis, so Hermitian input gives
        a Hermitian basis; its choice is not unique (an orthonormal null-space basis)."""
        from scipy.linalg import null_space
        given = normalize(cls(partial_basis_array))
        if labels is None and len(getattr(partial_basis_array, 'labels', ())) == len(given):
            labels = partial_basis_array.labels
        if not given.isherm:
            warn("(Some) elems not hermitian! The resulting basis also won't be.")
        if not given.isorthogonal:
            raise ValueError('The basis elements are not orthogonal!')
        if traceless is None:
            traceless = given.istraceless
        elif traceless and not given.istraceless:
            raise ValueError('The basis elements are not traceless (up to an identity element) '
                             'but a traceless basis was requested!')
        d = given.d
        if labels is not None and len(labels) not in (len(given), d*d):
            raise ValueError(f'Got {len(labels)} labels but expected {len(given)} or {d*d}')
        frame = cls.ggm(d).view(np.ndarray)
        # coordinates of the given elements in the Gell-Mann frame, tr(Lambda_j C_i)
        coords = np.einsum('jab,iba->ij', frame, given.view(np.ndarray))
        if given.isherm:
            coords = coords.real
        coords = _tidy(coords.astype(complex), d*d).real if given.isherm else _tidy(coords, d*d)
        if traceless:
            # the identity direction is fixed as the first element; complete the rest
            frame_rest, coords = frame[1:], coords[:, 1:]
        else:
            frame_rest = frame
        coords = coords[np.any(coords != 0, axis=1)]
        if coords.size:
            coords = np.concatenate((coords, null_space(coords).conj().T))
            elems = np.einsum('ij,jab->iab', coords, frame_rest)
        else:
            elems = frame_rest
        if traceless:
            elems = np.concatenate((frame[:1], elems))
        out = cls(elems, btype=btype or 'From partial')
        out.tidyup()
        if labels is not None and len(labels) == len(given):
            labels = list(labels)
            if traceless:
                is_id = [np.allclose(frame[0], e, rtol=0, atol=given._atol)
                         for e in given.view(np.ndarray)]
                if any(is_id):
                    labels.insert(0, labels.pop(is_id.index(True)))
            labels.extend(f'$C_{{{i}}}$' for i in range(len(labels), len(out)))
        if labels is not None:
            out.labels = list(labels)
        return out


def _norm(b):
    b = np.asarray(b)
    return np.linalg.norm(b, axis=(-1, -2))[..., None, None]


def normalize(b):
    """Frobenius-normalised copy (reference basis.py:629-647)."""
    arr = np.asarray(b)
    out = (arr/_norm(arr)).view(Basis)
    for attr in ('btype', 'labels', 'd'):
        if hasattr(b, attr):
            setattr(out, attr, getattr(b, attr))
    return out


def _tidy(arr, eps_scale=None):
    eps = np.finfo(float).eps*(eps_scale if eps_scale is not None else arr.shape[-1])
    arr = np.array(arr)
    arr.real[np.abs(arr.real) <= eps] = 0
    if np.iscomplexobj(arr):
        arr.imag[np.abs(arr.imag) <= eps] = 0
    return arr


def expand(M, basis, normalized=True, hermitian=False, tidyup=False):
    """c_j = tr(M C_j) / tr(C_j^dag C_j)  (reference basis.py:650-698)."""
    barr = np.asarray(basis)
    herm_basis = getattr(basis, 'isherm', None)
    if herm_basis is None:
        herm_basis = np.allclose(barr, barr.conj().swapaxes(-1, -2))
    real = hermitian and herm_basis
    coeffs = np.tensordot(np.asarray(M), barr, axes=[(-2, -1), (-1, -2)])
    if real:
        coeffs = coeffs.real
    if not normalized:
        norms = np.einsum('bij,bji->b', barr, barr)
        coeffs = coeffs/(norms.real if real else norms)
    return _tidy(coeffs) if tidyup else coeffs


def ggm_expand(M, traceless=False, hermitian=False, tidyup=False):
    """Closed-form expansion in the GGM basis (reference basis.py:701-787)."""
    M = np.asarray(M)
    if M.shape[-1] != M.shape[-2]:
        raise ValueError('M should be square in its last two axes')
    cast = (lambda a: a.real) if hermitian else (lambda a: a)
    square = M.ndim < 3
    if square:
        M = M[None]
    d = M.shape[-1]
    n_sym = d*(d - 1)//2
    rows, cols = np.triu_indices(d, k=1)
    l = np.arange(1, d)
    coeffs = np.zeros(M.shape[:-2] + (d*d,), dtype=float if hermitian else complex)
    if not traceless:
        coeffs[..., 0] = cast(np.trace(M, axis1=-2, axis2=-1))/np.sqrt(d)
    upper, lower = M[..., rows, cols], M[..., cols, rows]
    coeffs[..., 1:n_sym + 1] = cast(upper + lower)/np.sqrt(2)
    coeffs[..., n_sym + 1:2*n_sym + 1] = cast(1j*(upper - lower))/np.sqrt(2)
    diag = np.diagonal(M, axis1=-2, axis2=-1)
    coeffs[..., 2*n_sym + 1:] = cast(np.cumsum(diag[..., :-1], axis=-1) - l*diag[..., 1:])
    coeffs[..., 2*n_sym + 1:] /= np.sqrt(l*(l + 1))
    if square:
        coeffs = coeffs.squeeze()
    return _tidy(coeffs) if tidyup else coeffs


def equivalent_pauli_basis_elements(idx, N):
    """Indices, in the N-qubit Pauli basis, of the elements acting non-trivially only on the
    qubits *idx* (reference basis.py:790-800)."""
    idx = [idx] if isinstance(idx, (int, np.integer)) else list(idx)
    grids = np.ix_(*[range(4) if q in idx else [0] for q in range(N)])
    return np.ravel_multi_index(grids, [4]*N).ravel()


def remap_pauli_basis_elements(order, N):
    """Permutation of the N-qubit Pauli basis under a reordering of the qubits
    (reference basis.py:803-815)."""
    tuples = np.indices((4,)*N).reshape(N, 4**N).T
    return np.array([np.ravel_multi_index([tup[q] for q in order], (4,)*N) for tup in tuples])
