"""Randomised sweep of the concatenation routes against the from-scratch evaluation of the same
sequence and the oracle (not part of the test suite: run on the GPU box when the concatenation
kernels or their host bookkeeping change).

Every case draws T distinct pulses (shared or partly different noise operators, equal or different
segment counts) and a sequence of G positions over them, and checks
  * ff.concatenate(seq) [the one-call device route when pulses repeat and all carry every operator,
    else the plain rule] against concatenate_without_filter_function(seq) evaluated from scratch
    and against the oracle on the concatenated Hamiltonian,
  * pulse correlations (which='correlations') summing to the total,
  * ff.concatenate_periodic against ff.concatenate([pulse]*n).

Usage: python tests/tools/fuzz_concatenate.py [n_cases] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import ff_oracle as orc  # noqa: E402
import filter_functions_amd as ff  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)


def rel(got, ref):
    s = np.abs(ref).max()
    return np.abs(got - ref).max()/(s if s > 0 else 1.0)


worst = {}
t0 = time.time()
for case in range(n_cases):
    d = int(rng.choice([2, 2, 3, 4, 4, 5, 8]))
    T = int(rng.integers(1, 6))
    G = int(rng.choice([2, 3, 7, 16, 17, 40, 129, 300, 1001])) if d <= 4 else int(rng.integers(2, 20))
    A = int(rng.integers(1, 4))
    W = int(rng.choice([1, 3, 64, 65, 200]))
    btype = 'Pauli' if d in (2, 4, 8) and rng.random() < 0.5 else 'GGM'
    basis = ff.Basis.pauli(int(np.log2(d))) if btype == 'Pauli' else ff.Basis.ggm(d)
    same_length = rng.random() < 0.5
    partial = A > 1 and rng.random() < 0.3          # some pulses lack the last noise operator

    def herm(n):
        M = rng.standard_normal((n, d, d)) + 1j*rng.standard_normal((n, d, d))
        return M + M.conj().transpose(0, 2, 1)
    c_opers, n_opers = herm(2), herm(A)
    n_dt0 = int(rng.integers(1, 5))
    pulses = []
    for k in range(T):
        n_dt = n_dt0 if same_length else int(rng.integers(1, 5))
        own = A - 1 if (partial and k % 2) else A
        pulses.append(ff.PulseSequence(
            [[c_opers[i], rng.standard_normal(n_dt), f'c{i}'] for i in range(2)],
            [[n_opers[a], np.full(n_dt, 0.5 + a), f'n{a}'] for a in range(own)],
            rng.random(n_dt) + 0.1, basis))
    index = rng.integers(0, T, G)
    seq = [pulses[k] for k in index]
    omega = np.sort(rng.random(W))*rng.choice([5.0, 50.0]) - rng.choice([0.0, 2.0])
    tag = f'case {case}: d={d} T={T} G={G} A={A} W={W} {btype} same_length={same_length} partial={partial}'
    for p in pulses:
        p.cache_filter_function(omega)
    total = ff.concatenate(seq, calc_filter_function=True, omega=omega)
    scratch = ff.concatenate_without_filter_function(seq)
    H = orc.hamiltonian(scratch.c_opers, scratch.c_coeffs)
    D, V, Q = orc.diagonalize(H, scratch.dt)
    R_ref = orc.control_matrix_from_scratch(D, V, Q, omega, np.asarray(basis), scratch.n_opers,
                                            scratch.n_coeffs, scratch.dt)
    # Conditioning: the rule sums G terms of magnitude |R_atomic| whose phase errors grow like
    # g*eps; off resonance the sum itself may be much smaller than its terms (one frequency sampled
    # where 300 periods interfere destructively: 9e-10 relative to the sum, 2e-13 relative to the
    # terms).  Deviations from the from-scratch evaluation are therefore measured against the size
    # of the summed terms, G * max |R_atomic| (a NumPy restatement of the rule shows the same
    # deviations, profiles/r02_m_*).
    terms = G*max(np.abs(p.get_control_matrix(omega)).max() for p in pulses)

    def against_terms(got, ref, power=1):
        return np.abs(got - ref).max()/terms**power
    errs = {'R_vs_oracle': against_terms(total.get_control_matrix(omega), R_ref),
            'F_vs_oracle': against_terms(total.get_filter_function(omega), orc.filter_function(R_ref), 2),
            'R_vs_scratch': against_terms(total.get_control_matrix(omega), scratch.get_control_matrix(omega)),
            'U_total': rel(total.total_propagator, Q[-1])}
    if G <= 40:
        pc = ff.concatenate(seq, calc_pulse_correlation_FF=True, omega=omega)
        errs['correlations_sum'] = rel(pc.get_pulse_correlation_filter_function().sum(axis=(0, 1)),
                                       orc.filter_function(R_ref))
    if case % 3 == 0:
        reps = int(rng.choice([1, 2, 5, 64, 1000]))
        per = ff.concatenate_periodic(pulses[0], reps)
        rep = ff.concatenate([pulses[0]]*reps, calc_filter_function=True, omega=omega) if reps > 1 else pulses[0]
        size = reps*np.abs(pulses[0].get_control_matrix(omega)).max()
        errs['periodic_vs_repeated'] = (np.abs(per.get_control_matrix(omega) - rep.get_control_matrix(omega)).max()
                                        / size * min(1.0, G/reps))      # (judged with the tolerance of reps terms)
    # every evaluation route rounds its phases omega*t to eps*|omega t| radians: routes differ by that
    tolerance = max(1e-11, 8*np.finfo(float).eps*np.abs(omega).max()*scratch.tau)
    bad = {k: v for k, v in errs.items() if not v < tolerance}
    if bad:
        print('FAIL', tag, bad, flush=True)
        sys.exit(1)
    for k, v in errs.items():
        if v > worst.get(k, (0, ''))[0]:
            worst[k] = (v, tag)
    if case % 25 == 0:
        print(f'{case} cases, {time.time() - t0:.0f} s', flush=True)
print(f'{n_cases} cases passed in {time.time() - t0:.0f} s; worst relative errors:')
for k, (v, tag) in worst.items():
    print(f'  {k:22s} {v:.2e}   {tag}')
