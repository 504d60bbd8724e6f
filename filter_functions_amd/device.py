"""HBM-resident execution of the fused pipeline (``ffk_pipeline_dev``).

``DevicePipeline`` uploads one pulse and one omega block once, owns the output and scratch
buffers, and then launches the whole path -- diagonalise, control matrix, filter function,
(optionally) infidelity -- asynchronously on a stream with no host transfers, allocations or
synchronisation.  Device memory and streams come from PyTorch-ROCm (plumbing); the kernels are
libffk's, addressed through raw device pointers.  Used by bench.py and by the multi-GPU driver.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import check

__all__ = ['DevicePipeline', 'CapturedGraph', 'capture']

_capture_streams = {}


class CapturedGraph:
    """An instantiated hipGraph of ``_dev`` calls (``ffk_graph_*``, include/ffk.h): one runtime
    call replays what was captured.  The buffers the captured calls named must outlive it -- keep
    the owning objects referenced (*keep*)."""

    def __init__(self, handle, keep=None):
        self._lib = _lib.load()
        self._handle = handle
        self._keep = keep
        n = ctypes.c_int(0)
        check(self._lib.ffk_graph_node_count(handle, ctypes.byref(n)))
        self.nodes = n.value

    def launch(self, stream):
        """Replay on *stream* (a raw stream handle: an int or ``None`` for the null stream)."""
        check(self._lib.ffk_graph_launch(self._handle, stream))

    def __del__(self):
        handle, self._handle = getattr(self, '_handle', None), None
        if handle:
            self._lib.ffk_graph_destroy(handle)


def capture(enqueue, stream=None, keep=None):
    """Capture what ``enqueue(stream_handle)`` enqueues -- any sequence of device-pointer calls of
    the library on that stream, and on streams forked from and joined back to it with events --
    into a :class:`CapturedGraph`.  *stream*: raw handle of a created stream; default: a private
    capture stream of the current device."""
    lib = _lib.load()
    if stream is None:
        dev = ctypes.c_int(0)
        check(lib.ffk_get_device(ctypes.byref(dev)))
        stream = _capture_streams.get(dev.value)
        if stream is None:
            made = ctypes.c_void_p()
            check(lib.ffk_stream_create(ctypes.byref(made)))
            stream = _capture_streams[dev.value] = made.value
    check(lib.ffk_graph_capture_begin(stream))
    try:
        enqueue(stream)
    except BaseException:
        lib.ffk_graph_capture_abort(stream)
        raise
    handle = ctypes.c_void_p()
    check(lib.ffk_graph_capture_end(stream, ctypes.byref(handle)))
    return CapturedGraph(handle, keep=keep)


class DevicePipeline:
    """One pulse (c_opers/c_coeffs/n_opers/n_coeffs/dt/basis) on one omega block.

    spectrum: optional, shape ([[n_idx,] n_idx,] W); idx: noise-operator indices (default all).
    """

    def __init__(self, c_opers, c_coeffs, n_opers, n_coeffs, dt, basis, omega, spectrum=None,
                 idx=None, device=None):
        import torch
        self.torch = torch
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else device
        c_opers = np.asarray(c_opers, dtype=complex)
        H = np.einsum('ijk,il->ljk', c_opers, np.asarray(c_coeffs, dtype=float))
        dt = np.asarray(dt, dtype=float)
        t = np.concatenate(([0], dt.cumsum()))          # same expression as the reference
        self.G, self.d = H.shape[0], H.shape[1]
        self.A, self.N, self.W = len(n_opers), len(basis), len(omega)
        up = self._upload
        self.H, self.dt, self.t = up(H, complex), up(dt, float), up(t, float)
        self.omega = up(omega, float)
        self.basis = up(np.asarray(basis), complex)
        self.n_opers, self.n_coeffs = up(n_opers, complex), up(n_coeffs, float)
        self.c_opers = up(c_opers, complex)
        self.s_ndim, self.n_idx = 0, 0
        self.spectrum = self.idx = self.infid = None
        self._launch_args = {}
        if spectrum is not None:
            self.set_spectrum(spectrum, idx)
        kw = dict(device=self.device)
        self.eigvals = torch.empty((self.G, self.d), dtype=torch.float64, **kw)
        self.eigvecs = torch.empty((self.G, self.d, self.d), dtype=torch.complex128, **kw)
        self.propagators = torch.empty((self.G + 1, self.d, self.d), dtype=torch.complex128, **kw)
        self.control_matrix = torch.empty((self.A, self.N, self.W), dtype=torch.complex128, **kw)
        self.filter_function = torch.empty((self.A, self.A, self.W), dtype=torch.complex128, **kw)
        lib = _lib.load()
        self.ws_bytes = lib.ffk_pipeline_workspace_bytes(self.W, self.N, self.A, self.G, self.d,
                                                         max(self.n_idx, self.A), 3)
        if self.ws_bytes == 0:
            raise ValueError('unsupported problem shape')
        self.workspace = torch.empty(self.ws_bytes, dtype=torch.uint8, **kw)

    def _upload(self, arr, dtype):
        return self.torch.from_numpy(np.require(arr, dtype=dtype, requirements=['C', 'W'])).to(self.device)

    def set_spectrum(self, spectrum, idx=None):
        from . import util
        self._launch_args = {}
        idx = np.arange(self.A) if idx is None else np.asarray(idx)
        S = util.parse_spectrum(np.asanyarray(spectrum), np.empty(self.W), idx)
        self.s_ndim, self.n_idx = S.ndim, len(idx)
        self.spectrum = self._upload(S, complex)
        self.idx = self._upload(idx, np.int32)
        self.infid = self.torch.empty((self.n_idx, self.n_idx) if S.ndim == 3 else (self.n_idx,),
                                      dtype=self.torch.float64, device=self.device)

    @staticmethod
    def _p(tensor):
        return None if tensor is None else ctypes.c_void_p(tensor.data_ptr())

    def launch(self, stream=None, with_infidelity=True):
        """Enqueue one pass of the hot path on *stream* (default: torch's current stream)."""
        s = self.torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
        do_inf = with_infidelity and self.spectrum is not None
        args = self._launch_args.get(do_inf)
        if args is None:
            # the buffers of a pipeline never move: their addresses are converted once (set_spectrum,
            # which replaces three of them, drops this cache)
            p = self._p
            args = self._launch_args[do_inf] = (
                _lib.load().ffk_pipeline_dev,
                (p(self.H), p(self.dt), p(self.t), self.G, self.d, p(self.omega), self.W, p(self.basis),
                 self.N, p(self.n_opers), self.A, p(self.n_coeffs),
                 p(self.spectrum) if do_inf else None, self.s_ndim, p(self.idx) if do_inf else None,
                 self.n_idx, p(self.eigvals), p(self.eigvecs), p(self.propagators),
                 p(self.control_matrix), p(self.filter_function), p(self.infid) if do_inf else None,
                 p(self.workspace), self.ws_bytes))
        check(args[0](*args[1], s))

    def graph(self, with_infidelity=True):
        """The pass of :meth:`launch` as a captured hipGraph (cached; ``set_spectrum`` drops it):
        ``pipe.graph().launch(stream)`` enqueues the 6 launches of a pass with one runtime call."""
        do_inf = bool(with_infidelity and self.spectrum is not None)
        graphs = self._launch_args.setdefault('graphs', {})
        g = graphs.get(do_inf)
        if g is None:
            g = graphs[do_inf] = capture(lambda s: self.launch(stream=s, with_infidelity=do_inf))
        return g

    def check_status(self, stream=None):
        """Raise ``numpy.linalg.LinAlgError`` if the eigensolver of the last :meth:`launch` flagged
        a segment (no convergence, NaN/Inf in the Hamiltonian): the device-resident counterpart of
        the exception ``numeric.diagonalize`` raises.  Copies 4 bytes and synchronises the stream.
        Raises ``_lib.FFKKernelFault`` if a flag wait inside the accumulate kernel ran out in any
        launch (direct or through a captured graph) since the last check."""
        torch = self.torch
        s = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
        if getattr(self, '_n_failed', None) is None:
            self._n_failed = torch.zeros(1, dtype=torch.int32, device=self.device)
        check(_lib.load().ffk_eigensolver_status_dev(self._p(self.workspace), self.ws_bytes, self.G,
                                                     self.d, self._p(self._n_failed),
                                                     ctypes.c_void_p(s)))
        check(_lib.load().ffk_stream_synchronize(ctypes.c_void_p(s)))
        failed = int(self._n_failed.cpu().item())
        _lib.check_kernel_fault()
        if failed:
            raise np.linalg.LinAlgError(
                f'Jacobi eigensolver did not converge for {failed} segment(s)')

    def infidelity_from_shards(self, shards, omega, spectrum, idx, out, stream=None):
        """Device trapezoid straight on an all-gather buffer (n_shards, A, A, shard_width);
        *out* is a preallocated float64 tensor.  No allocation, no re-layout."""
        s = self.torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
        p = self._p
        check(_lib.load().ffk_infidelity_sharded_dev(
            p(shards), shards.shape[0], shards.shape[-1], shards.shape[1], p(spectrum),
            spectrum.dim(), p(omega), p(idx), idx.numel(), self.d, p(out), ctypes.c_void_p(s)))
        return out

    def infidelity_from(self, filter_function, omega, spectrum, idx, stream=None):
        """Device trapezoid on an arbitrary (gathered) F: tensors in, tensor out."""
        torch = self.torch
        s = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
        A, W = filter_function.shape[0], filter_function.shape[-1]
        n_idx = idx.numel()
        s_ndim = spectrum.dim()
        out = torch.empty((n_idx, n_idx) if s_ndim == 3 else (n_idx,), dtype=torch.float64,
                          device=self.device)
        lib = _lib.load()
        need = lib.ffk_infidelity_workspace_bytes(W, n_idx, s_ndim)
        if getattr(self, '_iws', None) is None or self._iws.numel() < need:
            self._iws = torch.empty(need, dtype=torch.uint8, device=self.device)
        p = self._p
        check(lib.ffk_infidelity_dev(p(filter_function), A, W, p(spectrum), s_ndim, p(omega),
                                     p(idx), n_idx, self.d, p(out), p(self._iws), need,
                                     ctypes.c_void_p(s)))
        return out

    def decay_amplitudes(self, omega_global=None, w_offset=0, stream=None):
        """Decay amplitudes of the noise operators selected by ``set_spectrum`` from the device
        control matrix of the last ``launch`` (``ffk_decay_amplitudes_shard_dev``): tensor
        ``(n_idx[, n_idx], N, N)``.  With *omega_global* (device tensor, the full grid) and
        *w_offset*, this block's contribution to the integral over the full grid (multi-GPU)."""
        torch = self.torch
        if self.spectrum is None:
            raise ValueError('set_spectrum() first')
        s = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
        omega = self.omega if omega_global is None else omega_global
        shape = (self.n_idx, self.n_idx, self.N, self.N) if self.s_ndim == 3 else \
            (self.n_idx, self.N, self.N)
        out = torch.empty(shape, dtype=torch.float64, device=self.device)
        lib = _lib.load()
        need = lib.ffk_decay_amplitudes_workspace_bytes(1, self.N, self.W, self.n_idx, self.s_ndim)
        if getattr(self, '_dws', None) is None or self._dws.numel() < need:
            self._dws = torch.empty(max(need, 16), dtype=torch.uint8, device=self.device)
        p = self._p
        check(lib.ffk_decay_amplitudes_shard_dev(
            p(self.control_matrix), 1, self.A, self.N, self.W, p(self.spectrum), self.s_ndim,
            p(omega), omega.numel(), int(w_offset), p(self.idx), self.n_idx, p(out), p(self._dws),
            need, ctypes.c_void_p(s)))
        return out

    def cumulant_function(self, decay_amplitudes, single_qubit=False, stream=None):
        """First-order cumulant function of device decay amplitudes ``(..., N, N)``."""
        torch = self.torch
        s = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
        gamma = decay_amplitudes.contiguous()
        batch = gamma.numel()//(self.N*self.N)
        out = torch.empty_like(gamma)
        lib = _lib.load()
        need = 0 if single_qubit else lib.ffk_cumulant_function_workspace_bytes(batch, self.N, self.d)
        if getattr(self, '_cws', None) is None or self._cws.numel() < need:
            self._cws = torch.empty(max(need, 16), dtype=torch.uint8, device=self.device)
        p = self._p
        check(lib.ffk_cumulant_function_dev(p(gamma), batch, self.N, self.d, p(self.basis),
                                            int(single_qubit), p(out), p(self._cws), need,
                                            ctypes.c_void_p(s)))
        return out

    def error_transfer_matrix(self, cumulant_function, stream=None):
        """``exp`` of the device cumulant function ``(..., N, N)`` summed over its leading axes (reference
        numeric.py:2049-2053), computed and left in HBM (``ffk_error_transfer_matrix_dev``: sum, 1-norm and every
        product of the scaling-and-squaring on the device; 16 bytes cross to the host to choose the number of
        squarings)."""
        torch = self.torch
        s = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
        K = cumulant_function.contiguous()
        n = K.shape[-1]
        if K.dim() < 2 or K.shape[-2] != n:
            raise ValueError(f'cumulant_function invalid shape: {tuple(K.shape)}')
        batch = K.numel()//(n*n)
        out = torch.empty((n, n), dtype=torch.float64, device=self.device)
        lib = _lib.load()
        need = lib.ffk_error_transfer_matrix_workspace_bytes(n)
        if getattr(self, '_ews', None) is None or self._ews.numel() < need:
            self._ews = torch.empty(need, dtype=torch.uint8, device=self.device)
        p = self._p
        check(lib.ffk_error_transfer_matrix_dev(p(K), batch, n, p(out), p(self._ews), need, ctypes.c_void_p(s)))
        return out

    def second_order_filter_function(self, stream=None):
        """Second-order filter function ``(A, A, N, N, W)`` of the pulse on this omega block, from
        the eigensystem of the last ``launch`` (``ffk_second_order_filter_function_dev``); stays in
        HBM."""
        torch = self.torch
        s = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
        lib = _lib.load()
        need = lib.ffk_second_order_workspace_bytes(self.W, self.N, self.A, self.G, self.d)
        if getattr(self, '_sows', None) is None or self._sows.numel() < need:
            self._sows = torch.empty(max(need, 16), dtype=torch.uint8, device=self.device)
        if getattr(self, 'filter_function_2', None) is None:
            self.filter_function_2 = torch.empty((self.A, self.A, self.N, self.N, self.W),
                                                 dtype=torch.complex128, device=self.device)
        p = self._p
        check(lib.ffk_second_order_filter_function_dev(
            p(self.eigvals), p(self.eigvecs), p(self.propagators), p(self.omega), self.W,
            p(self.basis), self.N, p(self.n_opers), self.A, p(self.n_coeffs), p(self.dt), p(self.t),
            self.G, self.d, p(self.filter_function_2), p(self._sows), need, ctypes.c_void_p(s)))
        return self.filter_function_2

    def frequency_shifts(self, omega_global=None, w_offset=0, stream=None):
        """Frequency shifts ``(n_idx[, n_idx], N, N)`` from the device second-order filter function
        (computed on demand); with *omega_global* / *w_offset* this block's contribution to the
        integral over the full grid (multi-GPU), like :meth:`decay_amplitudes`."""
        torch = self.torch
        if self.spectrum is None:
            raise ValueError('set_spectrum() first')
        if getattr(self, 'filter_function_2', None) is None:
            self.second_order_filter_function(stream)
        s = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
        omega = self.omega if omega_global is None else omega_global
        shape = (self.n_idx, self.n_idx, self.N, self.N) if self.s_ndim == 3 else \
            (self.n_idx, self.N, self.N)
        out = torch.empty(shape, dtype=torch.float64, device=self.device)
        lib = _lib.load()
        need = lib.ffk_frequency_shifts_workspace_bytes(self.W, self.n_idx, self.s_ndim)
        if getattr(self, '_fsws', None) is None or self._fsws.numel() < need:
            self._fsws = torch.empty(max(need, 16), dtype=torch.uint8, device=self.device)
        p = self._p
        check(lib.ffk_frequency_shifts_shard_dev(
            p(self.filter_function_2), self.A, self.N, self.W, p(self.spectrum), self.s_ndim,
            p(omega), omega.numel(), int(w_offset), p(self.idx), self.n_idx, p(out), p(self._fsws),
            need, ctypes.c_void_p(s)))
        return out

    def add_second_order_cumulant(self, cumulant_function, frequency_shifts, stream=None):
        """Adds the frequency-shift terms to a device cumulant function in place and returns it."""
        torch = self.torch
        s = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
        delta = frequency_shifts.contiguous()
        batch = delta.numel()//(self.N*self.N)
        lib = _lib.load()
        need = lib.ffk_cumulant_function_second_order_workspace_bytes(batch, self.N, self.d)
        if getattr(self, '_c2ws', None) is None or self._c2ws.numel() < need:
            self._c2ws = torch.empty(max(need, 16), dtype=torch.uint8, device=self.device)
        p = self._p
        check(lib.ffk_cumulant_function_second_order_dev(p(delta), batch, self.N, self.d,
                                                         p(self.basis), p(cumulant_function),
                                                         p(self._c2ws), need, ctypes.c_void_p(s)))
        return cumulant_function

    def infidelity_gradient(self, omega_global=None, w_offset=0, n_coeffs_deriv=None, stream=None):
        """Derivative of the infidelity by the amplitude of every control operator in every
        segment, tensor ``(n_nops, n_dt, n_ctrl)`` (``ffk_filter_function_derivative_shard_dev``),
        from the eigensystem of the last ``launch`` and the spectrum of ``set_spectrum`` (shape
        (W,) or (n_nops, W), all noise operators).  With *omega_global* / *w_offset* this block's
        contribution to the integral over the full grid (multi-GPU: sum the per-rank results, e.g.
        with ``parallel.sum_omega_shards``).  The filter-function derivative itself stays in
        ``self.filter_function_derivative`` (n_nops, n_dt, n_ctrl, W)."""
        torch = self.torch
        if self.spectrum is None or self.s_ndim > 2 or self.n_idx != self.A:
            raise ValueError('set_spectrum() with a spectrum of shape (W,) or (n_nops, W) first')
        s = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
        H = self.c_opers.shape[0]
        omega = self.omega if omega_global is None else omega_global
        lib = _lib.load()
        need = lib.ffk_filter_function_derivative_workspace_bytes(self.W, self.A, H, self.G, self.d)
        if need == 0:
            raise ValueError('the gradient kernels support 2 <= d <= 8')
        if getattr(self, '_gws', None) is None or self._gws.numel() < need:
            self._gws = torch.empty(need, dtype=torch.uint8, device=self.device)
        if getattr(self, 'filter_function_derivative', None) is None:
            self.filter_function_derivative = torch.empty((self.A, self.G, H, self.W),
                                                          dtype=torch.float64, device=self.device)
        ratio = None
        if n_coeffs_deriv is not None:
            ratio = (torch.as_tensor(np.asarray(n_coeffs_deriv, dtype=float), device=self.device)
                     / self.n_coeffs[:, None, :]).contiguous()
        out = torch.empty((self.A, self.G, H), dtype=torch.float64, device=self.device)
        p = self._p
        check(lib.ffk_filter_function_derivative_shard_dev(
            p(self.eigvals), p(self.eigvecs), p(self.propagators), p(self.omega), self.W,
            p(self.n_opers), self.A, p(self.n_coeffs), p(self.c_opers), H, p(ratio), p(self.dt),
            p(self.t), self.G, self.d, p(self.spectrum), self.s_ndim, p(omega), omega.numel(),
            int(w_offset), p(self.filter_function_derivative), p(out), p(self._gws), need,
            ctypes.c_void_p(s)))
        return out
