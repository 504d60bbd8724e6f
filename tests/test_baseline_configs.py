"""BASELINE.json configurations 3, 4 and 5 at their full sizes on the GPU (through the C ABI).

Each configuration is checked three ways: (i) against the reference's own outputs on a sub-grid
of the frequency axis (fixtures `tests/golden/{cfg3_subgrid,cfg4_subgrid,qft}.npz`, written by
oracle/make_golden.py from the imported reference; every frequency is independent, so the
full-size run must reproduce them at those frequencies), (ii) against the CPU oracle on another
subsample / on the integrals over the full grid, (iii) through size-independent properties
(exact Hermiticity of F, vanishing identity column, sharded == unsharded, trace preservation
of the error transfer matrix).  Inputs come from workloads.py.
"""
import numpy as np
import pytest

import ff_oracle as orc
import filter_functions_amd as ff
import workloads as wl
from conftest import load_golden, rel_err
from filter_functions_amd import numeric, util

pytestmark = pytest.mark.gpu
TOL = 1e-10          # acceptance bar (north_star)


def _hermitian_in_operators(F):
    return np.array_equal(F, F.conj().swapaxes(0, 1)) and np.all(np.diagonal(F).imag == 0)


# ---- config 5: examples/qft.py, d = 16, full error-transfer-matrix path --------------------------
def test_config5_qft_fixture_parity():
    g = load_golden('qft')
    omega = g['omega']
    qft = wl.qft_pulse(ff)
    U = wl.bit_reversal() @ qft.total_propagator
    assert util.oper_equiv(U, wl.qft_matrix(), eps=1e-13)[0]        # the example's own check
    assert rel_err(qft.total_propagator, g['total_propagator']) < 1e-13
    R = qft.get_control_matrix(omega)
    F = qft.get_filter_function(omega)
    assert R.shape == (18, 256, 64) and F.shape == (18, 18, 64)
    for k, row in enumerate(g['rows']):
        assert rel_err(R[row], g['control_matrix_rows'][k]) < 1e-12
    assert rel_err(F, g['filter_function']) < 1e-12
    assert _hermitian_in_operators(F)
    assert rel_err(ff.infidelity(qft, g['S2'], omega), g['infidelity_S2']) < 1e-12
    ids = [str(i) for i in g['decay_identifiers']]
    idx = [list(qft.n_oper_identifiers).index(i) for i in ids]
    gamma = numeric.calculate_decay_amplitudes(qft, g['S2'][idx], omega, n_oper_identifiers=ids)
    assert rel_err(gamma, g['decay_amplitudes_S2_sub']) < 1e-12
    # the same pulse assembled through the concatenation rule (every gate's control matrix cached
    # first; gates lack most of the 18 noise operators, whose rows are evaluated on the gate's own
    # control Hamiltonian, reference pulse_sequence.py:1789-1815)
    by_rule = wl.qft_pulse(ff, omega=omega)
    assert by_rule.is_cached('control_matrix')
    assert rel_err(by_rule.get_control_matrix(omega), R) < 1e-12
    assert rel_err(by_rule.get_filter_function(omega), F) < 1e-12


def test_config5_full_size_error_transfer_matrix():
    """16384 omega: control matrix -> decay amplitudes -> cumulant function -> exp, device
    resident, unsharded and as 8 logical frequency shards (the 8-GPU partition on one device)."""
    import torch
    from filter_functions_amd.device import DevicePipeline
    from filter_functions_amd.parallel import shard_bounds
    g = load_golden('qft')
    W = wl.CONFIG5['W']
    omega = np.logspace(-2, 2, W)
    qft = wl.qft_pulse(ff)
    A, d, N = 18, 16, 256
    S = np.outer(1e-6*(np.arange(A) + 1), 1/omega)
    args = (qft.c_opers, qft.c_coeffs, qft.n_opers, qft.n_coeffs, qft.dt, qft.basis)
    pipe = DevicePipeline(*args, omega, spectrum=S)
    pipe.launch()
    torch.cuda.synchronize()
    R = pipe.control_matrix.cpu().numpy()
    F = pipe.filter_function.cpu().numpy()
    # (i) the reference's outputs at the fixture's frequencies
    at = g['omega_index']
    assert np.array_equal(omega[at], g['omega'])
    assert rel_err(R[g['rows']][:, :, at], g['control_matrix_rows']) < 1e-12
    assert rel_err(F[:, :, at], g['filter_function']) < 1e-12
    # (ii) the oracle on another subsample
    rng = np.random.default_rng(5)
    sub = np.sort(rng.choice(W, 96, replace=False))
    D, V, Q = orc.diagonalize(orc.hamiltonian(qft.c_opers, qft.c_coeffs), qft.dt)
    basis = np.asarray(qft.basis)
    R_ref = orc.control_matrix_from_scratch(D, V, Q, omega[sub], basis, qft.n_opers, qft.n_coeffs,
                                            qft.dt)
    for a in range(A):
        assert rel_err(R[a][:, sub], R_ref[a]) < 1e-12
    # (iii) properties
    assert _hermitian_in_operators(F)
    assert np.abs(R[:, 0]).max() < 1e-13*np.abs(R).max()             # traceless noise operators
    infid = pipe.infid.cpu().numpy()
    assert rel_err(infid, orc.infidelity_from_filter_function(F, S, omega, np.arange(A), d)) < 1e-12
    # decay amplitudes over the full grid against the oracle's weighted-sum form on the device's R
    gamma = pipe.decay_amplitudes()
    gamma_ref = orc.decay_amplitudes_shard(R, S, omega, 0, np.arange(A))
    assert rel_err(gamma.cpu().numpy(), gamma_ref) < 1e-12
    # ... and through the reference-shaped host API on a coarser grid (trapezoid of the integrand)
    K = pipe.cumulant_function(gamma)
    K_ref = orc.cumulant_function(gamma_ref, basis)
    assert rel_err(K.cpu().numpy(), K_ref) < 1e-12
    K_total = K.sum(dim=0).cpu().numpy()
    U = ff.error_transfer_matrix(cumulant_function=K_total[None])
    U_ref = orc.error_transfer_matrix(K_ref)
    assert np.abs(U - U_ref).max() < TOL*np.abs(U_ref - np.eye(N)).max()
    # the same without leaving HBM (what bench.py's config-5 entry times)
    # (sums the 18 operators in order; torch's reduction above pairs them: equal to rounding, not to the bit)
    assert np.abs(pipe.error_transfer_matrix(K).cpu().numpy() - U).max() < 1e-15
    # trace preservation: first row of the transfer matrix is e_0 (identity element first)
    assert np.abs(U[0] - np.eye(N)[0]).max() < 1e-14
    # entanglement infidelity to first order = sum of the operators' infidelities
    assert np.isclose(1 - np.trace(U)/d**2, infid.sum(), rtol=1e-3)
    assert np.allclose(-np.trace(K_total)/d**2, infid.sum(), rtol=1e-10)
    # the 8-GPU partition as logical shards: global trapezoid weights, rank-ordered sum
    omega_dev = torch.from_numpy(omega).cuda()
    total = None
    for rank in range(8):
        w0, w1 = shard_bounds(W, 8, rank)
        part = DevicePipeline(*args, omega[w0:w1], spectrum=S[:, w0:w1])
        part.launch(with_infidelity=False)
        # (the number of segment chunks depends on the block width: same values, re-associated sum)
        assert rel_err(part.filter_function.cpu().numpy(), F[:, :, w0:w1]) < 1e-13
        contribution = part.decay_amplitudes(omega_global=omega_dev, w_offset=w0)
        total = contribution if total is None else total + contribution
    assert rel_err(total.cpu().numpy(), gamma_ref) < 1e-12
    K8 = part.cumulant_function(total).sum(dim=0).cpu().numpy()
    assert np.abs(ff.error_transfer_matrix(cumulant_function=K8[None]) - U_ref).max() \
        < TOL*np.abs(U_ref - np.eye(N)).max()


# ---- config 4: d = 8, 512 segments, 9 noise operators, 65536 omega in 8 shards -------------------
def test_config4_full_grid_sharded_and_unsharded():
    import torch
    from filter_functions_amd.device import DevicePipeline
    from filter_functions_amd.parallel import shard_bounds
    g = load_golden('cfg4_subgrid')
    cfg = wl.CONFIG4
    d, A, W, n = cfg['d'], cfg['A'], cfg['W'], cfg['n_shards']
    c_opers, c_coeffs, n_opers, n_coeffs, dt = wl.random_pulse_inputs(**cfg)
    omega = wl.random_pulse_omega(dt, W)
    S = 1e-3/omega
    basis = ff.Basis.pauli(3)
    pulse = ff.PulseSequence(list(zip(c_opers, c_coeffs)), list(zip(n_opers, n_coeffs)), dt, basis)
    assert np.array_equal(pulse.n_opers, n_opers)                   # 'B_0'..'B_8' sort in place
    args = (pulse.c_opers, pulse.c_coeffs, pulse.n_opers, pulse.n_coeffs, dt, basis)
    # the 8-GPU partition as 8 logical shards (one rank's shard = 8192 omega = the bench workload)
    dev = lambda a, ty: torch.from_numpy(np.ascontiguousarray(a, dtype=ty)).cuda()
    shards = torch.empty((n, A, A, W//n), dtype=torch.complex128, device='cuda')
    R = np.empty((A, d*d, W), dtype=complex)
    for rank in range(n):
        w0, w1 = shard_bounds(W, n, rank)
        part = DevicePipeline(*args, omega[w0:w1])
        part.launch()
        shards[rank] = part.filter_function
        R[:, :, w0:w1] = part.control_matrix.cpu().numpy()
    infid = torch.empty(A, dtype=torch.float64, device='cuda')
    part.infidelity_from_shards(shards, dev(omega, float), dev(S, complex),
                                torch.arange(A, dtype=torch.int32, device='cuda'), infid)
    torch.cuda.synchronize()
    F = shards.permute(1, 2, 0, 3).reshape(A, A, W).cpu().numpy()
    # (i) reference outputs at the fixture's 12 frequencies (all 512 segments)
    at = g['omega_index']
    assert np.array_equal(omega[at], g['omega'])
    assert np.abs(part.eigvals.cpu().numpy() - g['eigvals']).max() < 1e-12
    for a in range(A):
        assert rel_err(R[a][:, at], g['control_matrix'][a]) < TOL
    assert rel_err(F[:, :, at], g['filter_function']) < TOL
    # (ii) the oracle on a 128-omega subsample across all shards
    sub = np.sort(np.random.default_rng(4).choice(W, 128, replace=False))
    D, V, Q = orc.diagonalize(orc.hamiltonian(pulse.c_opers, pulse.c_coeffs), dt)
    R_ref = orc.control_matrix_from_scratch(D, V, Q, omega[sub], np.asarray(basis), pulse.n_opers,
                                            pulse.n_coeffs, dt)
    for a in range(A):
        assert rel_err(R[a][:, sub], R_ref[a]) < TOL
    assert rel_err(F[:, :, sub], orc.filter_function(R_ref)) < TOL
    infid_ref = orc.infidelity_from_filter_function(F, S, omega, np.arange(A), d)
    assert rel_err(infid.cpu().numpy(), infid_ref) < 1e-12
    # (iii) properties, and the unsharded pass over the whole grid
    assert _hermitian_in_operators(F)
    assert np.abs(R[:, 0]).max() < 1e-12*np.abs(R).max()
    whole = DevicePipeline(*args, omega, spectrum=S)
    whole.launch()
    torch.cuda.synchronize()
    # (the number of segment chunks depends on the block width: same values, re-associated sum)
    assert rel_err(whole.filter_function.cpu().numpy(), F) < 1e-13
    assert rel_err(whole.control_matrix.cpu().numpy(), R) < 1e-13
    assert rel_err(whole.infid.cpu().numpy(), infid.cpu().numpy()) < 1e-13
    assert _hermitian_in_operators(whole.filter_function.cpu().numpy())
    # the user-facing call on one rank's shard
    w0, w1 = shard_bounds(W, n, 3)
    assert rel_err(pulse.get_filter_function(omega[w0:w1]), F[:, :, w0:w1]) < 1e-13


# ---- config 3: 1000-gate randomized-benchmarking sequence by concatenation, 8192 omega -----------
def test_config3_thousand_gate_sequence_by_concatenation():
    g = load_golden('cfg3_subgrid')
    cfg = wl.CONFIG3
    omega = wl.rb_omega(cfg['W'], cfg['T'])
    _, cliffords = wl.rb_cliffords(ff, omega, cfg['T'])
    assert all(c.is_cached('control_matrix') for c in cliffords)
    at = g['omega_index']
    table = np.array([c.get_control_matrix(omega) for c in cliffords])
    assert rel_err(table[..., at], g['clifford_control_matrices']) < 1e-12
    draw = wl.rb_draw(cfg['n_gates'], cfg['seed'])
    assert np.array_equal(draw, g['draw'])
    seq = [cliffords[k] for k in draw]
    total = ff.concatenate(seq)
    assert len(total) == sum(len(wl.CLIFFORD_WORDS[k]) for k in draw)
    R = total.get_control_matrix(omega)
    F = total.get_filter_function(omega)
    assert R.shape == (1, 4, cfg['W'])
    assert rel_err(total.total_propagator, g['total_propagator']) < 1e-11
    # (i) reference outputs at 24 of the 8192 frequencies
    assert rel_err(R[..., at], g['control_matrix']) < TOL
    assert rel_err(F[..., at], g['filter_function']) < TOL
    # (ii) the oracle's concatenation rule on the materialised arrays, full size
    phases = np.array([p.get_total_phases(omega) for p in seq[:-1]]).cumprod(axis=0)
    L = util.adot(np.array([p.total_propagator_liouville for p in seq[:-1]]))
    R_ref = orc.control_matrix_from_atomic(phases, table[draw], L)
    assert rel_err(R, R_ref) < 1e-12
    S = wl.rb_spectrum(omega)
    infid = ff.infidelity(total, S, omega)
    assert rel_err(infid, orc.infidelity_from_filter_function(orc.filter_function(R_ref), S, omega,
                                                              np.arange(1), 2)) < 1e-12
    # (iii) the same sequence evaluated from scratch, segment by segment (3.3k segments)
    scratch = ff.concatenate_without_filter_function(seq)
    assert rel_err(scratch.get_filter_function(omega), F) < TOL
    # the plain (non-indexed) rule on the device as well
    R_plain = numeric.calculate_control_matrix_from_atomic(phases, table[draw], L)
    # (concatenate() takes the Liouville representation of the cumulative propagators, L above is the
    # cumulative product of the pulses' Liouville propagators: 1000 products associated differently)
    assert rel_err(R_plain, R) < 1e-11


def test_config3_optimized_gates_full_size():
    """Config 3 as the example runs it with its OPTIMISED gate set (examples/randomized_benchmarking.py:112-151;
    VERDICT r5 "what's missing" 2): the X/2 and Y/2 atoms are 100-segment pulses (data of examples/data/X2ID.mat,
    Y2ID.mat, held in tests/golden/rb_optimized_gates.npz), so their control matrices come from the d = 2
    from-scratch kernel, the 24 Cliffords (100 to 700 segments) from the concatenation rule, and the 1000-gate
    sequence (332 200 segments of bookkeeping) from the rule kernel -- all 8192 frequencies, against the reference's
    outputs at 16 of them, the oracle's rule on the materialised arrays, and the oracle from scratch on a Clifford."""
    g = load_golden('rb_optimized_gates')
    cfg = wl.CONFIG3
    omega = wl.rb_omega(cfg['W'], cfg['T'])
    gates = {name: (g[f'{name}_eps'], g[f'{name}_t'], g[f'{name}_B']) for name in ('X2', 'Y2')}
    atoms, cliffords = wl.rb_cliffords_optimized(ff, omega, gates)
    at = g['omega_index']
    for k, letter in enumerate('xy'):
        assert len(atoms[letter]) == 100
        assert rel_err(atoms[letter].get_control_matrix(omega)[..., at], g['atom_control_matrices'][k]) < TOL
    assert all(c.is_cached('control_matrix') for c in cliffords)
    assert np.array_equal([len(c) for c in cliffords], g['clifford_segments'])
    table = np.array([c.get_control_matrix(omega) for c in cliffords])
    assert rel_err(table[..., at], g['clifford_control_matrices']) < TOL
    # one Clifford (700 segments) from scratch on the device and by the oracle
    long = cliffords[16]
    fresh = ff.concatenate_without_filter_function([atoms[c] for c in wl.CLIFFORD_WORDS[16]])
    assert len(fresh) == 700
    assert rel_err(fresh.get_control_matrix(omega), table[16]) < TOL
    sub = np.linspace(0, cfg['W'] - 1, 40).astype(int)
    D, V, Q = orc.diagonalize(orc.hamiltonian(long.c_opers, long.c_coeffs), long.dt)
    R_orc = orc.control_matrix_from_scratch(D, V, Q, omega[sub], np.asarray(long.basis), long.n_opers, long.n_coeffs,
                                            long.dt)
    assert rel_err(table[16][..., sub], R_orc) < TOL
    draw = wl.rb_draw(cfg['n_gates'], cfg['seed'])
    seq = [cliffords[k] for k in draw]
    total = ff.concatenate(seq)
    assert len(total) == g['n_segments'] and abs(total.tau - g['tau']) < 1e-9*g['tau']
    F = total.get_filter_function(omega)
    assert rel_err(total.total_propagator, g['total_propagator']) < 1e-10
    assert rel_err(F[..., at], g['filter_function']) < 1e-9
    S = wl.rb_spectrum(omega)
    # (the reference's infidelity on the 16-frequency sub-grid integrates ITS cached filter function -- the one of the
    # concatenation rule; integrating ours there with the oracle's trapezoid)
    assert rel_err(orc.infidelity_from_filter_function(F[..., at], S[at], omega[at], np.arange(1), 2), g['infidelity']) < 1e-9
    # the oracle's rule on the materialised arrays, full size
    phases = np.array([p.get_total_phases(omega) for p in seq[:-1]]).cumprod(axis=0)
    L = util.adot(np.array([p.total_propagator_liouville for p in seq[:-1]]))
    R_ref = orc.control_matrix_from_atomic(phases, table[draw], L)
    assert rel_err(total.get_control_matrix(omega), R_ref) < 1e-10


@pytest.mark.slow
def test_published_example_periodic_driving():
    """The reference's timed example (doc/source/examples/periodic_driving.ipynb) at full size:
    10 000 periods by concatenate_periodic, by ff.concatenate, and the 200 002 segments written out
    and evaluated from scratch, against each other and (on a frequency sub-grid) the oracle."""
    from itertools import repeat
    cfg = wl.PERIODIC_DRIVING
    atomic, wait, full, omega = wl.periodic_driving(ff)
    atomic.cache_filter_function(omega)
    periodic = ff.concatenate_periodic(atomic, cfg['n_periods'])
    standard = ff.concatenate(repeat(atomic, cfg['n_periods']))
    assert len(periodic) == len(standard) == len(full) == cfg['n_periods']*cfg['n_per_period']
    assert rel_err(periodic.get_filter_function(omega), standard.get_filter_function(omega)) < 1e-10
    echo = ff.concatenate((wait, periodic, wait))
    written_out = ff.concatenate((wait, full, wait), calc_filter_function=False)
    assert len(written_out) == 200002 and not written_out.is_cached('filter_function')
    F = written_out.get_filter_function(omega)
    assert rel_err(echo.get_filter_function(omega), F) < 1e-9
    sub = np.linspace(0, len(omega) - 1, 5).astype(int)
    D, V, Q = orc.diagonalize(orc.hamiltonian(written_out.c_opers, written_out.c_coeffs), written_out.dt)
    assert rel_err(written_out.propagators, Q) < 1e-10
    R = orc.control_matrix_from_scratch(D, V, Q, omega[sub], np.asarray(written_out.basis),
                                        written_out.n_opers, written_out.n_coeffs, written_out.dt)
    assert rel_err(written_out.get_control_matrix(omega)[..., sub], R) < 1e-9
    assert rel_err(F[..., sub], orc.filter_function(R)) < 1e-9
    # and the reference's own outputs on the same inputs (oracle/make_golden.py periodic_driving)
    g = load_golden('periodic_driving')
    assert np.array_equal(omega, g['omega'])
    assert rel_err(atomic.get_filter_function(omega), g['atomic_filter_function']) < 1e-11
    assert rel_err(periodic.get_control_matrix(omega), g['periodic_control_matrix']) < 1e-8
    assert rel_err(periodic.get_filter_function(omega), g['periodic_filter_function']) < 1e-8
    assert rel_err(periodic.total_propagator, g['periodic_total_propagator']) < 1e-9
    assert rel_err(echo.get_filter_function(omega), g['echo_filter_function']) < 1e-8
    assert rel_err(F[..., g['omega_index']], g['written_out_filter_function']) < 1e-8
    assert rel_err(written_out.total_propagator, g['written_out_total_propagator']) < 1e-9
