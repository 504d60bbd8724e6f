"""Resident evaluation of one pulse on one frequency grid (``ffk_resident_*``, include/ffk.h).

``PulseSequence.get_filter_function(omega)`` on a pulse with nothing cached, followed by
``ff.infidelity(pulse, S, omega)``, is the north-star call.  Served by the array-in/array-out
entry points it costs five synchronous library calls and moves the control matrix (5x the bytes
of F at BASELINE config 2) across PCIe twice.  Here the whole pass is one call: one H2D copy of
the packed inputs, the fused device pipeline, one D2H copy of eigensystem + F into pinned host
memory that the returned arrays view directly.  The control matrix stays in HBM until someone
reads ``pulse.get_control_matrix`` / ``pulse.frequency_data['control_matrix']``
(:class:`Deferred`), and the infidelity integral runs on the resident F.
"""
import copy
import ctypes
import weakref

import numpy as np

from . import _lib
from ._lib import as_c128, as_f64, check, ptr

__all__ = ['Deferred', 'LazyCache', 'ResidentResult']


class Deferred:
    """A cache entry that is produced on first read (a device-resident array, a by-product nobody
    has asked for yet).  *nbytes* is what the entry will occupy once produced."""
    __slots__ = ('produce', 'nbytes')

    def __init__(self, produce, nbytes=0):
        self.produce = produce
        self.nbytes = nbytes


class LazyCache(dict):
    """``dict`` whose values may be :class:`Deferred`: reading such an entry produces the value,
    stores it in place of the placeholder and returns it.  Membership, length and iteration over
    the keys never trigger production, so ``is_cached`` stays free."""

    def __getitem__(self, key):
        value = dict.__getitem__(self, key)
        if type(value) is Deferred:
            value = value.produce()
            dict.__setitem__(self, key, value)
        return value

    def get(self, key, default=None):
        return self[key] if key in self else default

    def setdefault(self, key, default=None):
        if key not in self:
            dict.__setitem__(self, key, default)
        return self[key]

    def values(self):
        return [self[key] for key in self]

    def items(self):
        return [(key, self[key]) for key in self]

    def peek(self, key):
        """The raw entry (possibly still a :class:`Deferred`)."""
        return dict.__getitem__(self, key)

    def stored_nbytes(self):
        """Bytes held or promised by the entries, without producing any of them."""
        return sum(getattr(dict.__getitem__(self, key), 'nbytes', 0) for key in self)

    def copy(self):
        return LazyCache(dict.items(self))

    __copy__ = copy

    def __deepcopy__(self, memo):
        return LazyCache((key, copy.deepcopy(self[key], memo)) for key in self)

    def __reduce__(self):
        return (LazyCache, (dict(self.items()),))


def _view(address, count, dtype, shape, owner):
    """ndarray over *count* doubles of foreign memory at *address*; *owner* stays alive as long
    as the array (or anything derived from it) does."""
    buf = (ctypes.c_double*count).from_address(address)
    buf._owner = owner
    return np.ctypeslib.as_array(buf).view(dtype).reshape(shape)


class ResidentResult:
    """Owns one ``ffk_resident`` handle: the device-resident control matrix, filter function and
    frequency grid of one pass, and the pinned host block the small results live in."""

    def __init__(self):
        self._lib = _lib.load()
        self._handle = ctypes.c_void_p()
        check(self._lib.ffk_resident_create(ctypes.byref(self._handle)))
        self.shape = None
        self._filter_function = None      # weak: the array's buffer owns this object, not vice versa

    def __del__(self):
        handle, self._handle = getattr(self, '_handle', None), None
        if handle:
            self._lib.ffk_resident_destroy(handle)

    # A copy of a pulse does not own device memory: deep copies and pickles of the owner drop the
    # resident result (their cache entries are ordinary host arrays by then; the integral then takes
    # the array route).  Sharing by reference (copy.copy of a pulse) is fine: one owner, refcounted.
    def __deepcopy__(self, memo):
        return None

    def __reduce__(self):
        return (type(None), ())

    def evaluate(self, hamiltonian, dt, t, omega, basis, n_opers, n_coeffs, c_coeffs=None,
                 spectrum=None, idx=None, d_infidelity=None):
        """One pass; returns (eigvals, eigvecs, propagators, filter_function) as arrays that view
        the handle's pinned memory (no copy).  *hamiltonian* is the summed (G, d, d) array, or --
        with *c_coeffs* (n_cops, G) -- the control operators (n_cops, d, d), summed on the device.
        With *spectrum* (validated, ``util.parse_spectrum``), *idx* and *d_infidelity* the infidelity
        integral rides in the same pass and is returned as a fifth item (controls form only)."""
        H, dt, t, omega = as_c128(hamiltonian), as_f64(dt), as_f64(t), as_f64(omega)
        C, B, s = as_c128(basis), as_c128(n_opers), as_f64(n_coeffs)
        d = H.shape[1]
        G, W, N, A = len(dt), len(omega), len(C), len(B)
        out = [ctypes.c_void_p() for _ in range(4)]
        results = tuple(ctypes.byref(p) for p in out)
        # (plain addresses: every array is bound to a local name until the call has returned, and
        # `.ctypes.data` is half the price of `.ctypes.data_as(c_void_p)` -- ten arguments per call)
        infid = None
        if spectrum is not None:
            c = as_f64(c_coeffs)
            if c.shape != (len(H), G):
                raise ValueError(f'Expected c_coeffs of shape ({len(H)}, {G}), not {c.shape}.')
            idx = np.ascontiguousarray(idx, dtype=np.int32)
            real = not np.iscomplexobj(spectrum)
            S = as_f64(spectrum) if real else as_c128(spectrum)
            n_idx = len(idx)
            infid = np.empty((n_idx, n_idx) if S.ndim == 3 else (n_idx,), dtype=np.float64)
            check(self._lib.ffk_resident_filter_function_infidelity(
                self._handle, H.ctypes.data, len(H), c.ctypes.data, dt.ctypes.data, t.ctypes.data, G, d,
                omega.ctypes.data, W, C.ctypes.data, N, B.ctypes.data, A, s.ctypes.data, S.ctypes.data,
                S.ndim, int(real), idx.ctypes.data, n_idx, int(d_infidelity), *results, infid.ctypes.data))
        elif c_coeffs is None:
            check(self._lib.ffk_resident_filter_function(
                self._handle, H.ctypes.data, dt.ctypes.data, t.ctypes.data, G, d, omega.ctypes.data, W,
                C.ctypes.data, N, B.ctypes.data, A, s.ctypes.data, *results))
        else:
            c = as_f64(c_coeffs)
            if c.shape != (len(H), G):
                raise ValueError(f'Expected c_coeffs of shape ({len(H)}, {G}), not {c.shape}.')
            check(self._lib.ffk_resident_filter_function_from_controls(
                self._handle, H.ctypes.data, len(H), c.ctypes.data, dt.ctypes.data, t.ctypes.data, G, d,
                omega.ctypes.data, W, C.ctypes.data, N, B.ctypes.data, A, s.ctypes.data, *results))
        self.shape = (G, d, W, N, A)
        D = _view(out[0].value, G*d, np.float64, (G, d), self)
        V = _view(out[1].value, 2*G*d*d, np.complex128, (G, d, d), self)
        Q = _view(out[2].value, 2*(G + 1)*d*d, np.complex128, (G + 1, d, d), self)
        F = _view(out[3].value, 2*A*A*W, np.complex128, (A, A, W), self)
        # F is integrated on the DEVICE copy when infidelity() is handed this very array: it must not
        # be edited in place behind the device's back (an edit raises instead of being ignored;
        # `F.copy()` is an ordinary writable array that takes the array route)
        F.flags.writeable = False
        self._filter_function = weakref.ref(F)
        if infid is not None:
            return D, V, Q, F, infid
        return D, V, Q, F

    def adopt(self, shape, filter_function):
        """The handle was filled by another library call (a concatenation that left its control
        matrix and filter function resident): remember the shape and which host array is F."""
        self.shape = tuple(shape)
        self._filter_function = weakref.ref(filter_function)

    @property
    def handle(self):
        """The ``ffk_resident*`` (for calls that read several resident results)."""
        return self._handle

    @property
    def filter_function(self):
        """The host array of the resident F, if it is still alive."""
        return None if self._filter_function is None else self._filter_function()

    def timing(self):
        """Host-clock seconds of the last pass: (packing inputs, enqueueing copies and kernels,
        waiting for the stream)."""
        out = np.zeros(3)
        check(self._lib.ffk_resident_timing(self._handle, ptr(out)))
        return tuple(out)

    def control_matrix(self):
        """The resident control matrix (n_nops, n_basis, n_omega), copied to the host now."""
        G, d, W, N, A = self.shape
        R = np.empty((A, N, W), dtype=np.complex128)
        check(self._lib.ffk_resident_control_matrix(self._handle, ptr(R)))
        return R

    def control_matrix_nbytes(self):
        G, d, W, N, A = self.shape
        return 16*A*N*W

    def infidelity(self, spectrum, idx, d):
        """(1/2 pi d) int dw Re(S F) on the resident F; *spectrum* already validated
        (``util.parse_spectrum``), *idx* the noise-operator indices, *d* the pulse's (possibly
        user-overridden) dimension."""
        G, _, W, N, A = self.shape
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        real = not np.iscomplexobj(spectrum)
        S = as_f64(spectrum) if real else as_c128(spectrum)
        n_idx = len(idx)
        out = np.empty((n_idx, n_idx) if S.ndim == 3 else (n_idx,), dtype=np.float64)
        if W < 2:
            out[...] = 0.0
            return out
        check(self._lib.ffk_resident_infidelity(self._handle, S.ctypes.data, S.ndim, int(real),
                                                idx.ctypes.data, n_idx, int(d), out.ctypes.data))
        return out
