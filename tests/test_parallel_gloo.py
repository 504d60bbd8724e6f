"""CPU, world_size 2, gloo: the omega-sharding and all-gather logic of
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
