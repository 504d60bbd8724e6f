"""CPU, world_size 2 and 8, gloo: the omega-sharding and all-gather logic of
filter_functions_amd.parallel (partition, uneven blocks, complex payloads, layout), with the
oracle standing in for the per-shard compute (the product's compute is the HIP pipeline; these
tests only exercise the distribution logic, which is backend independent)."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip('torch')
import torch.multiprocessing as mp  # noqa: E402

from conftest import ROOT  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_omega, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist

    import ff_oracle as orc
    from filter_functions_amd.parallel import (gather_omega_shards, shard_bounds,
                                               sharded_filter_function)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'rand_d3_ggm.npz'))
    omega = np.linspace(-2.0, 9.0, n_omega)

    def compute_shard(omega_block):
        R = orc.control_matrix_from_scratch(g['eigvals'], g['eigvecs'], g['propagators'],
                                            omega_block, g['basis'], g['n_opers'], g['n_coeffs'],
                                            g['dt'], g['t'])
        return torch.from_numpy(orc.filter_function(R))

    F = sharded_filter_function(compute_shard, omega).numpy()
    # a real-valued payload with a different leading shape through the same collective
    w0, w1 = shard_bounds(n_omega, world, rank)
    local = torch.arange(w0, w1, dtype=torch.float64).repeat(2, 1)*(1.0)
    idx = gather_omega_shards(local, n_omega).numpy()
    # decay amplitudes: per-rank partial integrals with the global trapezoid weights, summed in
    # rank order on every rank (sum_omega_shards)
    from filter_functions_amd.parallel import sum_omega_shards
    e = np.load(os.path.join(ROOT, 'tests', 'golden', 'etm.npz'))
    R, S, om = e['g3_control_matrix'], e['g3_S2'], e['g3_omega']
    b0, b1 = shard_bounds(len(om), world, rank)
    part = orc.decay_amplitudes_shard(R[..., b0:b1], S[..., b0:b1], om, b0, np.arange(len(R)))
    gamma = sum_omega_shards(torch.from_numpy(part)).numpy()
    # frequency shifts: every frequency of the second-order filter function is independent, each
    # rank computes its block of it and integrates with the global weights
    so = np.load(os.path.join(ROOT, 'tests', 'golden', 'second_order.npz'))
    om2, S2 = so['g3_omega'], so['g3_S3']
    c0, c1 = shard_bounds(len(om2), world, rank)
    F2_block = orc.second_order_filter_function(so['g3_eigvals'], so['g3_eigvecs'],
                                                so['g3_propagators'], om2[c0:c1], so['g3_basis'],
                                                so['g3_n_opers'], so['g3_n_coeffs'], so['g3_dt'])
    part = orc.frequency_shifts_shard(F2_block, S2[..., c0:c1], om2, c0, np.arange(len(S2)))
    delta = sum_omega_shards(torch.from_numpy(part)).numpy()
    # infidelity gradient: each rank differentiates on its frequency block, global weights
    gr = np.load(os.path.join(ROOT, 'tests', 'golden', 'gradient.npz'))
    om3, S3 = gr['g3_omega'], gr['g3_S2']
    e0, e1 = shard_bounds(len(om3), world, rank)
    dF_block = orc.filter_function_derivative(gr['g3_eigvals'], gr['g3_eigvecs'],
                                              gr['g3_propagators'], om3[e0:e1], gr['g3_basis'],
                                              gr['g3_n_opers'], gr['g3_n_coeffs'], gr['g3_c_opers'],
                                              gr['g3_dt'])
    part = orc.infidelity_derivative_shard(dF_block, S3[..., e0:e1], om3, e0, 3)
    grad = sum_omega_shards(torch.from_numpy(part)).numpy()
    np.savez(os.path.join(out_dir, f'rank{rank}.npz'), F=F, idx=idx, gamma=gamma, delta=delta,
             grad=grad)
    dist.destroy_process_group()


@pytest.mark.parametrize('n_omega', [10, 13])       # even and uneven blocks
def test_sharded_filter_function_matches_unsharded(tmp_path, n_omega):
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import ff_oracle as orc
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_omega, str(tmp_path)), nprocs=world, join=True)
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'rand_d3_ggm.npz'))
    omega = np.linspace(-2.0, 9.0, n_omega)
    R = orc.control_matrix_from_scratch(g['eigvals'], g['eigvecs'], g['propagators'], omega,
                                        g['basis'], g['n_opers'], g['n_coeffs'], g['dt'], g['t'])
    F_ref = orc.filter_function(R)
    for rank in range(world):
        got = np.load(os.path.join(str(tmp_path), f'rank{rank}.npz'))
        assert got['F'].shape == F_ref.shape
        # the stand-in compute is BLAS-backed, whose blocking (hence rounding) depends on the block
        # width; the layout/gather logic itself is exact (integer payload below)
        assert np.abs(got['F'] - F_ref).max() <= 1e-13*np.abs(F_ref).max()
        assert np.array_equal(got['idx'], np.tile(np.arange(n_omega, dtype=float), (2, 1)))
    e = np.load(os.path.join(ROOT, 'tests', 'golden', 'etm.npz'))
    gammas = [np.load(os.path.join(str(tmp_path), f'rank{r}.npz'))['gamma'] for r in range(world)]
    assert np.array_equal(gammas[0], gammas[1])            # rank-order sum: identical everywhere
    ref = e['g3_decay_amplitudes_S2']
    assert np.abs(gammas[0] - ref).max() <= 1e-13*np.abs(ref).max()
    deltas = [np.load(os.path.join(str(tmp_path), f'rank{r}.npz'))['delta'] for r in range(world)]
    assert np.array_equal(deltas[0], deltas[1])
    ref = np.load(os.path.join(ROOT, 'tests', 'golden', 'second_order.npz'))['g3_frequency_shifts_S3']
    assert np.abs(deltas[0] - ref).max() <= 1e-13*np.abs(ref).max()
    grads = [np.load(os.path.join(str(tmp_path), f'rank{r}.npz'))['grad'] for r in range(world)]
    assert np.array_equal(grads[0], grads[1])
    ref = np.load(os.path.join(ROOT, 'tests', 'golden', 'gradient.npz'))['g3_infidelity_derivative_S2']
    assert np.abs(grads[0] - ref).max() <= 1e-12*np.abs(ref).max()


def test_shard_bounds_partition():
    from filter_functions_amd.parallel import shard_bounds
    for n in (1, 7, 8, 4096, 65536 + 3):
        for world in (1, 2, 3, 8):
            bounds = [shard_bounds(n, world, r) for r in range(world)]
            assert bounds[0][0] == 0 and bounds[-1][1] == n
            assert all(b[1] == c[0] for b, c in zip(bounds, bounds[1:]))
            sizes = [b[1] - b[0] for b in bounds]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


# ---- the buffer ring of the sharded step (bench.py --gpus N) on CPU / gloo --------------------------
class _RecordingStreams:
    """Stands in for the two HIP streams: work executes at issue time (CPU), while the object logs
    what each stream has waited for, so that the buffer-reuse rule of ShardedStepRing can be
    checked: before step c overwrites buffer set c % depth, the compute stream must have waited
    for the communication work of step c - depth."""

    def __init__(self, depth, compute_names=('compute',)):
        self.depth = depth
        names = tuple(compute_names) + ('comm',)
        self.position = {name: 0 for name in names}    # ops recorded so far per stream
        self.waited = {name: {} for name in names}     # stream -> {other stream: position}
        self.done_position = []                         # comm position after gather + integral of step i
        self.launches = 0
        self.waits_on_compute = 0

    def record(self, stream):
        self.position[stream] += 1
        if stream == 'comm':
            self.done_position.append(self.position[stream])
        return (stream, self.position[stream])

    def wait(self, stream, event):
        other, pos = event
        self.waited[stream][other] = max(self.waited[stream].get(other, 0), pos)
        if stream != 'comm':
            self.waits_on_compute += 1

    def on(self, stream):
        import contextlib
        return contextlib.nullcontext()

    def handle(self, stream):
        return stream

    def check_launch(self, stream='compute'):
        c = self.launches
        self.launches += 1
        if c >= self.depth:
            assert self.waited[stream].get('comm', 0) >= self.done_position[c - self.depth], \
                f'step {c} reuses buffer set {c % self.depth} before the gather of step {c - self.depth}'


class _FakePipe:
    """DevicePipeline stand-in on CPU tensors: F[a, b, w] = (1 + a + 2 b + i a b) omega_w (step + 1)."""

    def __init__(self, omega_block, A, d, streams, step_counter):
        self.A, self.d = A, d
        self.omega_block = torch.from_numpy(omega_block)
        self.filter_function = torch.zeros((A, A, len(omega_block)), dtype=torch.complex128)
        self.streams, self.step_counter = streams, step_counter

    @staticmethod
    def model(omega, A, step):
        a = torch.arange(A, dtype=torch.float64)
        coeff = (1 + a[:, None] + 2*a[None, :]) + 1j*(a[:, None]*a[None, :])
        return coeff[:, :, None]*omega[None, None, :]*(step + 1)

    def launch(self, stream, with_infidelity):
        assert stream.startswith('compute') and not with_infidelity
        self.streams.check_launch(stream)
        self.filter_function.copy_(self.model(self.omega_block, self.A, self.step_counter[0]))
        self.step_counter[0] += 1

    @staticmethod
    def integrate(F, omega, S, d):
        f = (F.diagonal(dim1=0, dim2=1).T*S).real            # (A, W)
        return torch.trapezoid(f, omega, dim=-1)/(2*np.pi*d)

    def infidelity_from_shards(self, shards, omega, S, idx, out, stream):
        assert stream == 'comm'
        world, A, _, width = shards.shape
        F = shards.permute(1, 2, 0, 3).reshape(A, A, world*width)
        out.copy_(self.integrate(F, omega, S, self.d))
        return out

    def infidelity_from(self, F_full, omega, S, idx, stream):
        assert stream == 'comm'
        return self.integrate(F_full, omega, S, self.d)


def _ring_worker(rank, world, port, n_omega, depth, n_steps, out_dir, n_compute=1):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist

    from filter_functions_amd.parallel import ShardedStepRing, shard_bounds
    dist.init_process_group('gloo', rank=rank, world_size=world)
    A, d = 3, 4
    omega = np.geomspace(0.1, 50.0, n_omega)
    S = 1e-3/omega
    w0, w1 = shard_bounds(n_omega, world, rank)
    names = ['compute'] if n_compute == 1 else [f'compute{j}' for j in range(n_compute)]
    streams = _RecordingStreams(depth, names)
    counter = [0]
    pipes = [_FakePipe(omega[w0:w1], A, d, streams, counter) for _ in range(depth)]
    ring = ShardedStepRing(pipes, n_omega, omega, S, names if n_compute > 1 else 'compute', 'comm',
                           world, rank, streams=streams)
    results = [ring.step().clone().numpy() for _ in range(n_steps)]
    np.savez(os.path.join(out_dir, f'ring{rank}.npz'), results=np.array(results),
             waits=streams.waits_on_compute)
    dist.destroy_process_group()


@pytest.mark.parametrize('n_omega,depth', [(12, 4), (13, 4), (16, 2), (40, 8)])
def test_sharded_step_ring(tmp_path, n_omega, depth):
    """World size 2: every step's infidelities equal the unsharded integral, for equal and unequal
    omega blocks, over more steps than buffer sets; buffer sets are never reused before the
    communication work that reads them, and the compute stream waits only twice per ring."""
    world, n_steps = 2, 3*depth + 1
    mp.spawn(_ring_worker, args=(world, _free_port(), n_omega, depth, n_steps, str(tmp_path)),
             nprocs=world, join=True)
    omega = torch.from_numpy(np.geomspace(0.1, 50.0, n_omega))
    S = 1e-3/omega
    ref = np.array([_FakePipe.integrate(_FakePipe.model(omega, 3, step), omega, S, 4).numpy()
                    for step in range(n_steps)])
    for rank in range(world):
        got = np.load(os.path.join(str(tmp_path), f'ring{rank}.npz'))
        assert got['results'].shape == ref.shape
        assert np.abs(got['results'] - ref).max() <= 1e-14*np.abs(ref).max()
        assert int(got['waits']) == (n_steps - 1)//(depth//2)


def test_sharded_step_ring_with_several_compute_streams(tmp_path):
    """Steps distributed round robin over three compute streams (passes in flight): every step's
    stream waits for the release of its own buffer set, results unchanged."""
    world, depth, n_steps, n_omega = 2, 6, 20, 24
    mp.spawn(_ring_worker, args=(world, _free_port(), n_omega, depth, n_steps, str(tmp_path), 3),
             nprocs=world, join=True)
    omega = torch.from_numpy(np.geomspace(0.1, 50.0, n_omega))
    S = 1e-3/omega
    ref = np.array([_FakePipe.integrate(_FakePipe.model(omega, 3, step), omega, S, 4).numpy()
                    for step in range(n_steps)])
    for rank in range(world):
        got = np.load(os.path.join(str(tmp_path), f'ring{rank}.npz'))
        assert np.abs(got['results'] - ref).max() <= 1e-14*np.abs(ref).max()
        assert int(got['waits']) == n_steps - depth


@pytest.mark.parametrize('n_compute', [1, 2])
def test_sharded_step_ring_world_size_8(tmp_path, n_compute):
    """The shape `bench.py --gpus 8` runs: eight ranks, eight buffer sets, one or two
    compute streams, an omega grid that does not divide by eight (blocks of 9 and 8 frequencies): every
    step's infidelities equal the unsharded integral on every rank, buffer sets are not reused early."""
    world, depth, n_steps, n_omega = 8, 8, 19, 67
    mp.spawn(_ring_worker, args=(world, _free_port(), n_omega, depth, n_steps, str(tmp_path), n_compute),
             nprocs=world, join=True)
    omega = torch.from_numpy(np.geomspace(0.1, 50.0, n_omega))
    S = 1e-3/omega
    ref = np.array([_FakePipe.integrate(_FakePipe.model(omega, 3, step), omega, S, 4).numpy()
                    for step in range(n_steps)])
    for rank in range(world):
        got = np.load(os.path.join(str(tmp_path), f'ring{rank}.npz'))
        assert got['results'].shape == ref.shape
        assert np.abs(got['results'] - ref).max() <= 1e-14*np.abs(ref).max()
        assert int(got['waits']) == ((n_steps - 1)//(depth//2) if n_compute == 1 else n_steps - depth)


def test_sharded_filter_function_world_size_8(tmp_path):
    """Eight ranks, 67 frequencies: partition, all-gather of F, rank-ordered sums of the partial decay
    amplitudes, frequency shifts and gradients -- identical on every rank, equal to the unsharded values."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import ff_oracle as orc
    world, n_omega = 8, 67
    mp.spawn(_worker, args=(world, _free_port(), n_omega, str(tmp_path)), nprocs=world, join=True)
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'rand_d3_ggm.npz'))
    omega = np.linspace(-2.0, 9.0, n_omega)
    R = orc.control_matrix_from_scratch(g['eigvals'], g['eigvecs'], g['propagators'], omega,
                                        g['basis'], g['n_opers'], g['n_coeffs'], g['dt'], g['t'])
    F_ref = orc.filter_function(R)
    got = [np.load(os.path.join(str(tmp_path), f'rank{r}.npz')) for r in range(world)]
    for r in range(world):
        assert np.abs(got[r]['F'] - F_ref).max() <= 1e-13*np.abs(F_ref).max()
        assert np.array_equal(got[r]['idx'], np.tile(np.arange(n_omega, dtype=float), (2, 1)))
        for key in ('gamma', 'delta', 'grad'):
            assert np.array_equal(got[r][key], got[0][key]), key
    ref = np.load(os.path.join(ROOT, 'tests', 'golden', 'etm.npz'))['g3_decay_amplitudes_S2']
    assert np.abs(got[0]['gamma'] - ref).max() <= 1e-13*np.abs(ref).max()
