"""Frequency-axis sharding over the GPUs of one node.

Every omega is independent in diagonalize -> control matrix -> filter function; only the
trapezoid of the infidelity couples neighbours.  One process per GPU (``torch.distributed``,
backend ``nccl`` = RCCL over xGMI): each rank evaluates a contiguous omega block with the HIP
pipeline (the tiny omega-independent prologue is recomputed redundantly, cheaper than a
broadcast), then a single all-gather reassembles F(omega) on every rank, after which the
infidelity integral runs on the full grid exactly as in the single-GPU case (no halo, no
all-reduce, bit-identical arithmetic).  torch is plumbing here: device memory, the process
group and the collective; all arithmetic is libffk's.
"""
import os

import numpy as np

__all__ = ['shard_bounds', 'gather_omega_shards', 'sharded_filter_function', 'sum_omega_shards',
           'sharded_error_transfer_matrix', 'ShardedStepRing', 'PeerGather']


def shard_bounds(n_omega, world_size, rank):
    """Contiguous block [w0, w1) of rank *rank*; block sizes differ by at most one."""
    if not 0 <= rank < world_size:
        raise ValueError(f'rank {rank} outside world of size {world_size}')
    base, extra = divmod(n_omega, world_size)
    w0 = rank*base + min(rank, extra)
    return w0, w0 + base + (1 if rank < extra else 0)


def _all_gather(recv, send, group=None):
    """recv[r] <- send of rank r.  nccl (= RCCL): one collective into one buffer.  gloo has no flat
    all-gather, and none at all for device tensors: those (rehearsals of the N > 1 flow with several
    ranks on one GPU, bench.py FFK_BENCH_REHEARSE) are staged through the host."""
    import torch
    import torch.distributed as dist
    if dist.get_backend(group) != 'gloo':
        dist.all_gather_into_tensor(recv, send, group=group)
    elif send.is_cuda:
        parts = [torch.empty(send.shape, dtype=send.dtype) for _ in range(recv.shape[0])]
        dist.all_gather(parts, send.cpu(), group=group)
        recv.copy_(torch.stack(parts))
    else:
        dist.all_gather(list(recv.unbind(0)), send, group=group)


def gather_omega_shards(local, n_omega, group=None):
    """All-gather tensors whose LAST axis is this rank's omega block into the full
    (..., n_omega) tensor, omega fastest, on every rank.

    *local* is a torch tensor (cuda for nccl/RCCL, cpu for gloo) of shape (..., w1 - w0).
    Uneven blocks are padded to the largest one for the collective and trimmed afterwards.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if world == 1 and not os.environ.get('FFK_FORCE_COLLECTIVE'):   # test hook: run the
        return local                                                 # collective on one rank too
    lead = local.shape[:-1]
    widths = [shard_bounds(n_omega, world, r)[1] - shard_bounds(n_omega, world, r)[0]
              for r in range(world)]
    wmax = max(widths)
    send = local
    if local.shape[-1] != wmax:
        send = torch.zeros(lead + (wmax,), dtype=local.dtype, device=local.device)
        send[..., :local.shape[-1]] = local
    # complex dtypes travel as interleaved reals
    flat = torch.view_as_real(send.contiguous()) if send.is_complex() else send.contiguous()
    recv = torch.empty((world,) + flat.shape, dtype=flat.dtype, device=flat.device)
    _all_gather(recv, flat, group)
    if send.is_complex():
        recv = torch.view_as_complex(recv)
    # (world, ..., wmax) -> (..., world, wmax) -> (..., n_omega)
    nd = recv.dim()
    perm = tuple(range(1, nd - 1)) + (0, nd - 1)
    stacked = recv.permute(perm)
    if all(w == wmax for w in widths):
        return stacked.reshape(lead + (world*wmax,)).contiguous()
    parts = [stacked[..., r, :widths[r]] for r in range(world)]
    return torch.cat(parts, dim=-1).contiguous()


def sharded_filter_function(compute_shard, omega, group=None):
    """Evaluate ``compute_shard(omega_block) -> tensor (..., len(omega_block))`` on this rank's
    block of *omega* and return the gathered (..., len(omega)) result on every rank."""
    import torch.distributed as dist

    omega = np.asarray(omega)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    w0, w1 = shard_bounds(len(omega), world, rank)
    local = compute_shard(omega[w0:w1])
    return gather_omega_shards(local, len(omega), group=group)


def sum_omega_shards(local, group=None):
    """Sum per-rank contributions to an integral over omega (decay amplitudes: each rank
    integrates its block with the global trapezoid weights, ``ffk_decay_amplitudes_shard_dev``)
    on every rank.  The partial results are all-gathered and added in rank order, so the sum is
    bit-reproducible and identical on all ranks (a ring all-reduce adds in a rank-dependent
    order); the payload is tiny (n_nops * d^4 doubles)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if world == 1 and not os.environ.get('FFK_FORCE_COLLECTIVE'):
        return local
    flat = local.contiguous()
    recv = torch.empty((world,) + flat.shape, dtype=flat.dtype, device=flat.device)
    _all_gather(recv, flat, group)
    total = recv[0].clone()
    for r in range(1, world):
        total += recv[r]
    return total


def sharded_error_transfer_matrix(pipe, omega_global, w_offset, single_qubit=False, group=None,
                                  second_order=False):
    """Error transfer matrix of a pulse whose frequency axis is sharded over the ranks (BASELINE
    config 5): *pipe* is this rank's ``DevicePipeline`` over its omega block (already launched),
    *omega_global* the full grid as a device tensor, *w_offset* the block's first index.

    Each rank integrates its block of the decay amplitudes with the global trapezoid weights
    (``ffk_decay_amplitudes_shard_dev``), the (n_nops, d^2, d^2) partial results are summed in rank
    order on every rank (:func:`sum_omega_shards`: one small all-gather, bit-reproducible), and
    the cumulant contraction and the matrix exponential -- omega independent, tiny -- run
    redundantly on every rank.  With *second_order* the frequency shifts are treated the same way
    (every frequency of the second-order filter function is independent, so each rank computes its
    block of it; ``ffk_frequency_shifts_shard_dev``; one more small all-gather) and their
    commutator terms are added to the cumulant function.  Returns ``(decay_amplitudes,
    cumulant_function, U)``; the first two are device tensors, ``U`` a NumPy array.
    """
    gamma = sum_omega_shards(pipe.decay_amplitudes(omega_global=omega_global, w_offset=w_offset),
                             group=group)
    K = pipe.cumulant_function(gamma, single_qubit=single_qubit)
    if second_order:
        delta = sum_omega_shards(pipe.frequency_shifts(omega_global=omega_global, w_offset=w_offset),
                                 group=group)
        K = pipe.add_second_order_cumulant(K, delta)
    U = pipe.error_transfer_matrix(K).cpu().numpy()
    return gamma, K, U


class _RawDeviceBuffer:
    """Device memory straight from hipMalloc (libffk ``ffk_malloc``): an IPC handle names a whole
    allocation, so buffers that other ranks map must not be sub-allocations of PyTorch's caching
    allocator.  ``tensor()`` views the memory as a torch tensor (``__cuda_array_interface__``)."""

    def __init__(self, shape, typestr, itemsize, finegrained=False):
        import ctypes
        from . import _lib
        self._lib = _lib.load()
        self.shape = tuple(int(n) for n in shape)
        self.nbytes = int(np.prod(self.shape))*itemsize
        ptr = ctypes.c_void_p()
        allocate = self._lib.ffk_malloc_finegrained if finegrained else self._lib.ffk_malloc
        _lib.check(allocate(ctypes.byref(ptr), max(self.nbytes, 16)))
        self.ptr = ptr.value
        _lib.check(self._lib.ffk_memset(ctypes.c_void_p(self.ptr), 0, max(self.nbytes, 16), None))
        _lib.check(self._lib.ffk_device_synchronize())
        self.__cuda_array_interface__ = {'shape': self.shape, 'typestr': typestr,
                                         'data': (self.ptr, False), 'version': 2}

    def tensor(self, device):
        import torch
        return torch.as_tensor(self, device=device)

    def __del__(self):
        ptr, self.ptr = getattr(self, 'ptr', None), None
        if ptr:
            self._lib.ffk_free(ptr)      # (argtypes declare the pointer: no import at shutdown)


class PeerGather:
    """One-sided all-gather of the ranks' F blocks (``csrc/peer.hip``): every rank pushes its block
    into slot ``rank`` of a buffer set on every rank through IPC-mapped pointers, completion and
    buffer reuse travel as sequence numbers in flag words.  No RCCL kernel, no LDS: the copy shares
    the CUs with the accumulate kernel.

    Protocol of step c (buffer set k = c mod depth), all on the communication stream:
      push:   wait until every peer has consumed step c - depth (ack word >= c - depth + 1), copy
              the block to every rank's set k;
      signal: store c + 1 to this rank's flag word and "c steps consumed" to its ack word on every
              rank (the previous launch -- the copy -- has completed and released its writes);
      wait:   poll the local flag words until every rank has signalled >= c + 1.
    The rank that is furthest behind never waits for anybody, so the scheme cannot deadlock; every
    poll has a timeout (2 s, :meth:`set_timeout_ms`) that sets the sticky ``error`` word instead of
    hanging, and a rank whose word is set signals a poison value from then on, so that its peers
    fail too instead of integrating a slot that was never filled (``csrc/peer.hip``).
    """

    def __init__(self, depth, world, rank, block_shape, device, group=None):
        import ctypes
        import torch
        import torch.distributed as dist
        from . import _lib
        self._lib, self._check = _lib.load(), _lib.check
        self.depth, self.world, self.rank, self.device = depth, world, rank, device
        self.block_bytes = int(np.prod(block_shape))*16
        if self.block_bytes % 16:
            raise ValueError('blocks are complex128')
        # Every rank takes part in every collective of the set-up whatever happens locally: a rank
        # that fails publishes None / ok = False instead of leaving the others waiting.
        self.ok = True
        self._opened = []
        self._fixed = None
        mine = None
        try:
            # the gather buffers are written by peer GPUs and read by local kernels every `depth`
            # steps: fine-grained as well, so that no reader depends on an L2 line of an earlier step
            self._sets = [_RawDeviceBuffer((world,) + tuple(block_shape), '<c16', 16, finegrained=True)
                          for _ in range(depth)]
            # [flags | acks]: polled by running kernels while peers write them -> fine-grained
            self._words = _RawDeviceBuffer((2, world), '<i8', 8, finegrained=True)
            self.gathered = [b.tensor(device) for b in self._sets]
            words = self._words.tensor(device)
            self.flags, self.acks = words[0], words[1]
            self.error = torch.zeros(1, dtype=torch.int32, device=device)
            mine = []
            for buf in self._sets + [self._words]:
                handle = ctypes.create_string_buffer(64)
                self._check(self._lib.ffk_ipc_get_handle(ctypes.c_void_p(buf.ptr), handle))
                mine.append(handle.raw)
        except Exception as err:          # noqa: BLE001 -- reported through self.ok / self.reason
            self.ok, self.reason, mine = False, f'local set-up failed: {err}', None
        everyone = [None]*world
        dist.all_gather_object(everyone, mine, group=group)
        if any(entry is None for entry in everyone):
            self.ok = False
            self.reason = getattr(self, 'reason', 'a peer could not export its buffers')
            return
        try:
            bases = np.zeros((world, depth + 1), dtype=np.int64)
            for p in range(world):
                for j in range(depth + 1):
                    if p == rank:
                        bases[p, j] = (self._sets + [self._words])[j].ptr
                    else:
                        out = ctypes.c_void_p()
                        self._check(self._lib.ffk_ipc_open_handle(everyone[p][j], ctypes.byref(out)))
                        self._opened.append(out.value)
                        bases[p, j] = out.value
            # device tables: where, on rank p, this rank's slot / flag word / ack word lives
            dst = bases[:, :depth].T + rank*self.block_bytes                          # (depth, world)
            self._dst = torch.from_numpy(np.ascontiguousarray(dst)).to(device)
            self._flag_at = torch.from_numpy(bases[:, depth] + 8*rank).to(device)     # (world,)
            self._ack_at = torch.from_numpy(bases[:, depth] + 8*(world + rank)).to(device)
        except Exception as err:          # noqa: BLE001
            self.ok, self.reason = False, f'mapping the peers\' buffers failed: {err}'

    @staticmethod
    def _p(tensor):
        import ctypes
        return ctypes.c_void_p(tensor.data_ptr())

    def step(self, c, block, stream):
        """Push *block* (this rank's F of step *c*) to every rank and wait for everybody's; returns
        the buffer set (world, ...) that holds step c once the stream has passed."""
        k = c % self.depth
        fixed = self._fixed
        if fixed is None:      # device addresses that never change, converted once
            fixed = self._fixed = ([self._dst[j].data_ptr() for j in range(self.depth)],
                                   self.acks.data_ptr(), self._flag_at.data_ptr(),
                                   self._ack_at.data_ptr(), self.flags.data_ptr(), self.error.data_ptr())
        dst, acks, flag_at, ack_at, flags, error = fixed
        self._check(self._lib.ffk_peer_step_dev(block.data_ptr(), self.block_bytes, dst[k], acks,
                                                max(0, c - self.depth + 1), flag_at, ack_at, flags,
                                                self.world, self.rank, c, error, stream))
        return self.gathered[k]

    ERRORS = {1: 'an acknowledgement was not received in time (the copy to that peer was skipped)',
              2: 'a peer\'s signal was not received in time',
              3: 'a peer reported a failure of its own (poisoned sequence word)'}

    def error_code(self):
        """This rank's sticky error word (0 = no poll has failed); synchronises the device."""
        import torch
        torch.cuda.synchronize(self.device)
        return int(self.error.cpu().item())

    def check(self):
        """Raise if a poll of THIS rank failed (reads one word; synchronises the device).  A rank
        whose word is set poisons its signals, so every peer's word is set one step later; for a
        verdict that is the same on all ranks at once use :meth:`ShardedStepRing.check`."""
        code = self.error_code()
        if code:
            raise RuntimeError(f'one-sided all-gather failed on rank {self.rank}: '
                               + self.ERRORS.get(code, f'error word {code}'))

    def set_timeout_ms(self, ms):
        """Timeout of every poll from now on (default 2 s / FFK_PEER_TIMEOUT_MS): it must cover
        legitimate host stalls of a peer (first-launch compilation, profilers, GC pauses)."""
        self._check(self._lib.ffk_peer_set_timeout_ms(float(ms)))

    def close(self):
        import ctypes
        for ptr in self._opened:
            self._lib.ffk_ipc_close_handle(ctypes.c_void_p(ptr))
        self._opened = []


class _CudaStreams:
    """Stream plumbing of :class:`ShardedStepRing` on PyTorch-ROCm streams and events."""

    def __init__(self, compute, comm, events=64):
        import torch
        self.torch, self.compute, self.comm = torch, compute, comm
        # a ring of reusable events: a step needs its two events for `depth` steps at most, and
        # creating one per record cost more host time than recording it
        self._events = [torch.cuda.Event() for _ in range(events)]
        self._next = 0

    def record(self, stream):
        event = self._events[self._next]
        self._next = (self._next + 1) % len(self._events)
        event.record(stream)
        return event

    def wait(self, stream, event):
        stream.wait_event(event)

    def on(self, stream):
        return self.torch.cuda.stream(stream)

    def handle(self, stream):
        return stream.cuda_stream


class ShardedStepRing:
    """The frequency-sharded step of ``bench.py --gpus N`` and of any driver that evaluates many
    independent pulses: rank r computes F on its omega block, one all-gather reassembles F(omega)
    on every rank, the infidelity is integrated over the full grid.

    The collective and the integral of step i run on a communication stream while the following
    steps compute: *pipes* is a ring of ``depth`` buffer sets (one ``DevicePipeline`` each, same
    pulse and omega block) used round robin.  Buffer set k is read by the gather of step c and
    written again by step c + depth; the communication stream is in order, so it is enough that
    the compute stream waits -- every depth/2 steps, a wait costs ~4 us of barrier-packet handling
    -- for the communication work of depth/2 steps ago: for every step j of the following
    half-window, gather(j - depth) is older than that.

    *streams* abstracts record/wait/handle (default: PyTorch-ROCm streams); the CPU/gloo tests
    inject a recording fake to check exactly this ordering logic.
    """

    def __init__(self, pipes, n_omega, omega_full, spectrum_full, compute_stream, comm_stream,
                 world, rank, group=None, streams=None, gather='rccl', use_graph=False):
        import torch
        self.torch = torch
        # use_graph: the pass of a step is replayed from a captured hipGraph (one runtime call
        # instead of 5-6 launches through the binding); on one rank without an exchange the
        # integral is part of the same graph.  step(eager=True) enqueues that step call by call
        # (bench.py's HIP-event instrumentation of the accumulate kernel needs the calls).
        self.use_graph = bool(use_graph) and streams is None
        self._step_graphs = {}
        # FFK_GRAPH_COLLECTIVE=1 (experimental, equal blocks only): the WHOLE sharded step -- pass,
        # hand-over to the communication stream, RCCL all-gather, integral, join -- is captured as one
        # graph per buffer set and replayed with one runtime call.  Rehearsed on one rank only (RCCL
        # refuses several ranks on one device, so a multi-rank capture cannot be tried on a one-GPU
        # box): not the default.
        self.graph_collective = self.use_graph and bool(os.environ.get('FFK_GRAPH_COLLECTIVE'))
        if self.graph_collective:
            import torch.distributed as dist
            self.graph_collective = dist.is_initialized() and dist.get_backend(group) == 'nccl'
        self.pipes = list(pipes)
        self.depth = len(self.pipes)
        if self.depth < 2 or self.depth % 2:
            raise ValueError('the ring needs an even number (>= 2) of buffer sets')
        self.world, self.rank, self.group, self.n_omega = world, rank, group, n_omega
        # compute_stream may be a list: consecutive steps then go to the streams round robin, so that
        # the latency-bound launches of one pass run beside the accumulate kernel of another (as in
        # the N = 1 bench); every step then waits for the release of its own buffer set
        self.compute_streams = list(compute_stream) if isinstance(compute_stream, (list, tuple)) \
            else [compute_stream]
        self.compute_stream, self.comm_stream = self.compute_streams[0], comm_stream
        self.streams = streams if streams is not None else _CudaStreams(self.compute_stream, comm_stream, events=max(64, 8*self.depth))
        first = self.pipes[0]
        device, A = first.filter_function.device, first.A
        w0, w1 = shard_bounds(n_omega, world, rank)
        self.width = w1 - w0
        self.equal_shards = all(
            shard_bounds(n_omega, world, r)[1] - shard_bounds(n_omega, world, r)[0] == self.width
            for r in range(world))
        self.omega_full = torch.as_tensor(np.ascontiguousarray(omega_full, dtype=float)).to(device)
        self.spectrum_full = torch.as_tensor(
            np.ascontiguousarray(spectrum_full, dtype=complex)).to(device)
        self.idx = torch.arange(A, dtype=torch.int32, device=device)
        n_out = (A, A) if self.spectrum_full.dim() == 3 else (A,)
        self.gathered = [torch.empty((world, A, A, self.width), dtype=torch.complex128, device=device)
                         for _ in range(self.depth)]
        self.infid = [torch.empty(n_out, dtype=torch.float64, device=device)
                      for _ in range(self.depth)]
        self.free_events = [None]*self.depth
        self.count = 0
        self.count_offset = 0
        self._gloo = None
        self._real_views = {}
        # gather = 'push': the one-sided all-gather (PeerGather) instead of the collective; needs
        # equal blocks and real streams; 'auto': try it, verify one round, else the collective
        self.peer = None
        # gather = 'none' (one rank): nothing to exchange, the integral reads the pass's own F -- the
        # ring is then just the schedule: passes on the compute streams, integrals on their own stream
        self.local_only = gather == 'none'
        if self.local_only and world != 1:
            raise ValueError("gather='none' is for one rank")
        self._own_shard = [pipe.filter_function.unsqueeze(0) for pipe in self.pipes] if self.local_only else None
        # (one rank: only with the test hook FFK_FORCE_COLLECTIVE, to exercise / time the path)
        if gather in ('push', 'auto') and self.equal_shards and streams is None and \
                (world > 1 or os.environ.get('FFK_FORCE_COLLECTIVE')):
            self.peer = self._try_peer_gather(first, group, required=(gather == 'push'))
        if self.peer is not None:
            self.gathered = self.peer.gathered
        self.gather = 'none' if self.local_only else ('push' if self.peer is not None else 'rccl')

    def _try_peer_gather(self, first, group, required):
        """Set the one-sided gather up and verify one round trip of a known pattern on every rank;
        all ranks agree (minimum over ranks) whether to use it."""
        import torch.distributed as dist
        torch = self.torch
        block = first.filter_function
        peer = PeerGather(self.depth, self.world, self.rank, block.shape, block.device, group=group)
        ok = 0
        if peer.ok:
            try:
                probe = torch.full_like(block, complex(self.rank + 1, -(self.rank + 1)))
                torch.cuda.synchronize(probe.device)
                got = peer.step(0, probe, self.streams.handle(self.comm_stream))
                peer.check()
                expect = torch.arange(1, self.world + 1, dtype=torch.float64, device=got.device)
                good = (got.real.amax(dim=(1, 2, 3)) == expect).all() and \
                    (got.real.amin(dim=(1, 2, 3)) == expect).all() and \
                    (got.imag.amax(dim=(1, 2, 3)) == -expect).all()
                ok = int(bool(good))
            except Exception:             # noqa: BLE001 -- the verdict below is collective
                ok = 0
        on_gpu = dist.get_backend(group) != 'gloo'
        verdict = torch.tensor([ok], dtype=torch.int32, device=block.device if on_gpu else 'cpu')
        dist.all_reduce(verdict, op=dist.ReduceOp.MIN, group=group)
        if int(verdict.item()) != 1:
            peer.close()
            if required:
                raise RuntimeError('one-sided all-gather failed its round-trip check: '
                                   + getattr(peer, 'reason', 'pattern mismatch or timeout'))
            return None
        self.count_offset = 1          # the probe was step 0 of the peer protocol
        return peer

    def check(self):
        """COLLECTIVE: raise on every rank if the one-sided all-gather failed on any rank since the
        ring was built (MAX all-reduce of the sticky error words).  Results of ``step()`` must not
        be consumed without it: a rank whose push was skipped poisons its signals, but the step in
        which that happens may already have been integrated elsewhere.  With the RCCL collective
        (or one rank) there is nothing to check.  Synchronises the device."""
        if self.peer is None:
            return
        import torch.distributed as dist
        torch = self.torch
        torch.cuda.synchronize(self.peer.device)
        on_gpu = dist.get_backend(self.group) != 'gloo'
        word = self.peer.error.to(torch.int32).clone()
        word = word if on_gpu else word.cpu()
        dist.all_reduce(word, op=dist.ReduceOp.MAX, group=self.group)
        code = int(word.item())
        if code:
            raise RuntimeError('one-sided all-gather failed on some rank: '
                               + PeerGather.ERRORS.get(code, f'error word {code}')
                               + '; results since the last successful check are void')

    def _local_step_graph(self, k):
        """One rank, nothing to exchange: pass + integral of buffer set k as ONE graph, in stream
        order (a set always returns to the same compute stream when the number of sets is a
        multiple of the number of streams, so no events are needed either)."""
        g = self._step_graphs.get(k)
        if g is None:
            from .device import capture
            pipe = self.pipes[k]

            def enqueue(s):
                pipe.launch(stream=s, with_infidelity=False)
                pipe.infidelity_from_shards(self._own_shard[k], self.omega_full, self.spectrum_full,
                                            self.idx, self.infid[k], stream=s)
            g = self._step_graphs[k] = capture(enqueue)
        return g

    def _collective_step_graph(self, k, compute):
        """Pass + all-gather + integral of buffer set k as one graph, captured on *compute* (the
        communication stream is forked from and joined back to it with events)."""
        g = self._step_graphs.get(('coll', k))
        if g is None:
            import torch.distributed as dist
            from .device import capture
            st, pipe = self.streams, self.pipes[k]
            send, recv = pipe.filter_function, self.gathered[k]
            views = (self.torch.view_as_real(recv), self.torch.view_as_real(send))

            def enqueue(s):
                pipe.launch(stream=s, with_infidelity=False)
                ready = st.record(compute)
                with st.on(self.comm_stream):
                    st.wait(self.comm_stream, ready)
                    dist.all_gather_into_tensor(views[0], views[1], group=self.group)
                    pipe.infidelity_from_shards(recv, self.omega_full, self.spectrum_full, self.idx,
                                                self.infid[k], stream=st.handle(self.comm_stream))
                    done = st.record(self.comm_stream)
                st.wait(compute, done)
            g = self._step_graphs[('coll', k)] = capture(enqueue, stream=st.handle(compute), keep=views)
        return g

    def step(self, eager=False):
        """Enqueue one sharded step; returns the tensor that will hold its infidelities (valid
        once the communication stream has passed the step AND -- with the one-sided gather -- a
        later :meth:`check` has passed)."""
        import torch.distributed as dist
        st = self.streams
        c = self.count
        self.count += 1
        k = c % self.depth
        pipe = self.pipes[k]
        half = self.depth//2
        compute = self.compute_streams[c % len(self.compute_streams)]
        graphed = self.use_graph and not eager
        if self.local_only and self.use_graph and self.depth % len(self.compute_streams) == 0:
            # (eager or replayed: everything of the step on its compute stream)
            if graphed:
                self._local_step_graph(k).launch(st.handle(compute))
                return self.infid[k]
            s = st.handle(compute)
            pipe.launch(stream=s, with_infidelity=False)
            return pipe.infidelity_from_shards(self._own_shard[k], self.omega_full,
                                               self.spectrum_full, self.idx, self.infid[k], stream=s)
        if (self.graph_collective and self.peer is None and not self.local_only and self.equal_shards
                and self.depth % len(self.compute_streams) == 0):
            # whole step from one graph (or, for an instrumented step, the same calls in the same
            # stream order); a set always returns to the same compute stream and the graph joins the
            # communication stream back: stream order alone protects the buffer set
            if graphed:
                self._collective_step_graph(k, compute).launch(st.handle(compute))
                return self.infid[k]
            pipe.launch(stream=st.handle(compute), with_infidelity=False)
            ready = st.record(compute)
            with st.on(self.comm_stream):
                st.wait(self.comm_stream, ready)
                views = self._real_views.get(k)
                if views is None:
                    views = self._real_views[k] = (self.torch.view_as_real(self.gathered[k]),
                                                   self.torch.view_as_real(pipe.filter_function))
                dist.all_gather_into_tensor(views[0], views[1], group=self.group)
                out = pipe.infidelity_from_shards(self.gathered[k], self.omega_full, self.spectrum_full,
                                                  self.idx, self.infid[k], stream=st.handle(self.comm_stream))
                done = st.record(self.comm_stream)
            st.wait(compute, done)
            return out
        if len(self.compute_streams) == 1:
            if c >= half and c % half == 0:
                st.wait(compute, self.free_events[(c - half) % self.depth])
        elif c >= self.depth:
            st.wait(compute, self.free_events[k])       # the gather of step c - depth read set k
        if graphed:
            pipe.graph(with_infidelity=False).launch(st.handle(compute))
        else:
            pipe.launch(stream=st.handle(compute), with_infidelity=False)
        ready = st.record(compute)
        if self.peer is not None or self.local_only:
            # one-sided: push this rank's block everywhere, poll for everybody's.  Every call names
            # its stream: no "current stream" context to enter (15 us of host time per step)
            st.wait(self.comm_stream, ready)
            comm = st.handle(self.comm_stream)
            recv = self._own_shard[k] if self.local_only else \
                self.peer.step(c + self.count_offset, pipe.filter_function, comm)
            out = pipe.infidelity_from_shards(recv, self.omega_full, self.spectrum_full, self.idx,
                                              self.infid[k], stream=comm)
            self.free_events[k] = st.record(self.comm_stream)
            return out
        with st.on(self.comm_stream):
            st.wait(self.comm_stream, ready)
            if self.equal_shards:
                # one collective into a preallocated buffer; the integral reads the shards in place
                send, recv = pipe.filter_function, self.gathered[k]
                if self._gloo is None:
                    self._gloo = dist.get_backend(self.group) == 'gloo'
                if self._gloo:                                     # gloo has no flat all-gather
                    if send.is_cuda:       # (nor device tensors: complex ones staged as reals)
                        _all_gather(self.torch.view_as_real(recv), self.torch.view_as_real(send),
                                    self.group)
                    else:
                        dist.all_gather(list(recv.unbind(0)), send, group=self.group)
                else:
                    views = self._real_views.get(k)
                    if views is None:      # complex tensors travel as interleaved reals; views cached
                        views = self._real_views[k] = (self.torch.view_as_real(recv),
                                                       self.torch.view_as_real(send))
                    dist.all_gather_into_tensor(views[0], views[1], group=self.group)
                out = pipe.infidelity_from_shards(recv, self.omega_full, self.spectrum_full, self.idx,
                                                  self.infid[k], stream=st.handle(self.comm_stream))
            else:
                full = gather_omega_shards(pipe.filter_function, self.n_omega, group=self.group)
                out = pipe.infidelity_from(full, self.omega_full, self.spectrum_full, self.idx,
                                           stream=st.handle(self.comm_stream))
            self.free_events[k] = st.record(self.comm_stream)
        return out
