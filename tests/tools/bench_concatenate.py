"""BASELINE config 3: filter function of a 1000-gate randomized-benchmarking sequence built by
concatenation (examples/randomized_benchmarking.py:95-151, naive gates, d=2, 1 noise operator,
8192 omega).  Times ff.concatenate(...) on the GPU (whole call, host bookkeeping included) and
the CPU oracle's concatenation rule on the same atomic control matrices.

    python tests/tools/bench_concatenate.py [--gates 1000] [--omega 8192]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))

import ff_oracle as orc  # noqa: E402
import filter_functions_amd as ff  # noqa: E402
from filter_functions_amd import numeric, util  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gates', type=int, default=1000)
    ap.add_argument('--omega', type=int, default=8192)
    ap.add_argument('--reps', type=int, default=5)
    args = ap.parse_args()
    X, Y = util.paulis[1], util.paulis[2]
    T = 20.0
    omega = 2*np.pi*np.geomspace(1e-2/(7*151*T), 1e2/T, args.omega)
    X2 = ff.PulseSequence([[X/2, [np.pi/2/T], 'X']], [[X/2, [1], 'X']], [T])
    Y2 = ff.PulseSequence([[Y/2, [np.pi/2/T], 'Y']], [[X/2, [1], 'X']], [T])
    for p in (X2, Y2):
        p.cache_control_matrix(omega)
    t0 = time.perf_counter()
    cliffords = np.array([
        Y2 @ Y2 @ Y2 @ Y2, X2 @ X2, Y2 @ Y2, Y2 @ Y2 @ X2 @ X2, X2 @ Y2, X2 @ Y2 @ Y2 @ Y2,
        X2 @ X2 @ X2 @ Y2, X2 @ X2 @ X2 @ Y2 @ Y2 @ Y2, Y2 @ X2, Y2 @ X2 @ X2 @ X2,
        Y2 @ Y2 @ Y2 @ X2, Y2 @ Y2 @ Y2 @ X2 @ X2 @ X2, X2, X2 @ X2 @ X2, Y2, Y2 @ Y2 @ Y2,
        X2 @ Y2 @ Y2 @ Y2 @ X2 @ X2 @ X2, X2 @ X2 @ X2 @ Y2 @ Y2 @ Y2 @ X2, X2 @ X2 @ Y2,
        X2 @ X2 @ Y2 @ Y2 @ Y2, Y2 @ Y2 @ X2, Y2 @ Y2 @ X2 @ X2 @ X2, X2 @ Y2 @ X2,
        X2 @ Y2 @ Y2 @ Y2 @ X2], dtype=object)
    print(f'Clifford group by concatenation: {time.perf_counter() - t0:.3f} s')
    rng = np.random.default_rng(0)
    draw = rng.integers(0, 24, args.gates)
    seq = list(cliffords[draw])
    n_seg = sum(len(p) for p in seq)

    times = []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        total = ff.concatenate(seq)
        F = total.get_filter_function(omega)
        times.append(time.perf_counter() - t0)
    t_gpu = min(times)
    print(f'GPU  ff.concatenate of {args.gates} gates ({n_seg} segments), {args.omega} omega: '
          f'{t_gpu*1e3:.1f} ms (min of {args.reps}; whole Python call)')

    # kernel-only: the indexed rule on pre-staged tables
    table = np.array([c.get_control_matrix(omega) for c in cliffords])
    tp = np.array([c.get_total_phases(omega) for c in cliffords])
    L = util.adot(np.array([p.total_propagator_liouville for p in seq[:-1]]))
    t0 = time.perf_counter()
    R = numeric.calculate_control_matrix_from_atomic_indexed(tp, table, draw.astype(np.int32), L)
    t_call = time.perf_counter() - t0
    print(f'GPU  indexed rule alone (tables H2D + kernel + R D2H): {t_call*1e3:.2f} ms')

    # CPU oracle on the materialised arrays (what the reference's numeric core does)
    t0 = time.perf_counter()
    phases = np.array([p.get_total_phases(omega) for p in seq[:-1]]).cumprod(axis=0)
    R_atomic = np.array([p.get_control_matrix(omega) for p in seq])
    t_prep = time.perf_counter() - t0
    t0 = time.perf_counter()
    R_ref = orc.control_matrix_from_atomic(phases, R_atomic, L)
    t_cpu = time.perf_counter() - t0
    err = np.abs(R - R_ref).max()/np.abs(R_ref).max()
    print(f'CPU  oracle rule: {t_cpu:.2f} s (+ {t_prep:.2f} s materialising phases and atomic '
          f'control matrices), max rel err GPU vs CPU {err:.1e}')
    elements = args.gates*args.omega*1*4
    print(f'elements (gates*omega*nops*d^2) = {elements:.3g}: GPU {elements/t_gpu:.3g} el/s (whole call), '
          f'CPU {elements/(t_cpu + t_prep):.3g} el/s')
    assert err < 1e-12 and F.shape == (1, 1, args.omega)


if __name__ == '__main__':
    main()
