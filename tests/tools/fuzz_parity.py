"""Randomised parity sweep against the oracle (not part of the test suite: run on the GPU box when
kernels change).  Usage: python tests/tools/fuzz_parity.py [n_cases] [seed] [long | d=<n>]
("d=4": every case at that dimension -- the sweep of one accumulate kernel)
("long": segment counts up to a few thousand -- the unfused front end, the chunked scans, the
long-sequence kernel choice at d = 2 -- on small frequency grids and d <= 8, so that the oracle keeps up)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import ff_oracle as orc  # noqa: E402
import filter_functions_amd as ff  # noqa: E402
from filter_functions_amd import numeric  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
long_sequences = len(sys.argv) > 3 and sys.argv[3] == 'long'
fixed_d = int(sys.argv[3][2:]) if len(sys.argv) > 3 and sys.argv[3].startswith('d=') else None
rng = np.random.default_rng(seed)


def rel(got, ref):
    s = np.abs(ref).max()
    return np.abs(got - ref).max()/(s if s > 0 else 1.0)


worst = {}
t0 = time.time()
for case in range(n_cases):
    d = int(rng.integers(2, 17))
    G = int(rng.integers(1, 40))
    A = int(rng.integers(1, 6))
    W = int(rng.choice([1, 3, 31, 64, 65, 127, 200, 513]))
    if long_sequences:
        d = int(rng.choice([2, 2, 2, 3, 4, 4, 5, 8]))
        G = int(rng.choice([63, 64, 65, 257, 1023, 1024, 1025, 2500, 4097]))
        W = int(rng.choice([1, 3, 64, 65]))
    if fixed_d is not None:
        d = fixed_d
        G = int(rng.choice([1, 2, 7, 8, 9, 16, 17, 33, 64, 100, 257]))
    n_cops = int(rng.integers(1, 4))
    btype = 'Pauli' if d in (2, 4, 8, 16) and rng.random() < 0.5 else 'GGM'

    def herm(n):
        M = rng.standard_normal((n, d, d)) + 1j*rng.standard_normal((n, d, d))
        return M + M.conj().transpose(0, 2, 1)
    c_opers, n_opers = herm(n_cops), herm(A)
    c_coeffs = rng.standard_normal((n_cops, G))*rng.choice([0.1, 1.0, 5.0])
    if G > 2 and rng.random() < 0.3:
        c_coeffs[:, int(rng.integers(0, G))] = 0.0          # idle segment: degenerate spectrum
    n_coeffs = rng.random((A, G)) + 0.1
    dt = rng.random(G)*rng.choice([0.1, 1.0, 3.0]) + 0.05
    if long_sequences:
        dt /= G/10          # total duration of order ten: frequencies stay comparable
    omega = np.sort(rng.random(W))*rng.choice([5.0, 50.0]) - rng.choice([0.0, 2.0])
    if W > 2 and rng.random() < 0.5:
        omega[int(rng.integers(0, W))] = 0.0
        omega = np.sort(omega)
    basis = ff.Basis.pauli(int(np.log2(d))) if btype == 'Pauli' else ff.Basis.ggm(d)
    pulse = ff.PulseSequence(list(zip(c_opers, c_coeffs)), list(zip(n_opers, n_coeffs)), dt, basis)
    tag = f'case {case}: d={d} G={G} A={A} W={W} {btype}'
    H = orc.hamiltonian(pulse.c_opers, pulse.c_coeffs)
    D, V, Q = orc.diagonalize(H, dt)
    R_ref = orc.control_matrix_from_scratch(D, V, Q, omega, np.asarray(basis), pulse.n_opers,
                                            pulse.n_coeffs, dt)
    if case % 2:
        # array route: control matrix first, filter function from it
        errs = {'R': rel(pulse.get_control_matrix(omega), R_ref),
                'F': rel(pulse.get_filter_function(omega), orc.filter_function(R_ref))}
    else:
        # resident route (round 2): one library call, control matrix fetched from HBM afterwards,
        # infidelity below integrated on the resident F
        errs = {'F': rel(pulse.get_filter_function(omega), orc.filter_function(R_ref))}
        assert W == 0 or pulse._resident is not None
        errs['R'] = rel(pulse.get_control_matrix(omega), R_ref)
    if W > 1:
        S = 1/(1 + omega**2)
        idx = np.arange(A)
        errs['Gamma'] = rel(numeric.calculate_decay_amplitudes(pulse, S, omega),
                            orc.decay_amplitudes(R_ref, S, omega, idx))
        ref = orc.infidelity_from_filter_function(orc.filter_function(R_ref), S, omega, idx, d)
        errs['infid'] = rel(ff.infidelity(pulse, S, omega), ref)
    if d <= 6 and A*d*d <= 80 and G <= 12 and W <= 130:
        F2_ref = orc.second_order_filter_function(D, V, Q, omega, np.asarray(basis), pulse.n_opers,
                                                  pulse.n_coeffs, dt)
        errs['F2'] = rel(pulse.get_filter_function(omega, order=2), F2_ref)
    for k, v in errs.items():
        if v > worst.get(k, (0, ''))[0]:
            worst[k] = (v, tag)
        if not v < (1e-9 if long_sequences else 1e-10):
            print('FAIL', tag, k, v, flush=True)
print(f'{n_cases} cases in {time.time() - t0:.0f} s; worst relative errors:')
for k, (v, tag) in worst.items():
    print(f'  {k:6s} {v:.2e}   {tag}')
