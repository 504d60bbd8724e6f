"""Where the time of the user-facing call goes (BASELINE config 2, host arrays in and out):
PulseSequence(...) -> pulse.get_filter_function(omega) -> ff.infidelity(pulse, S, omega).

    python tools/time_api_call.py [--reps 200]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import filter_functions_amd as ff  # noqa: E402
import workloads as wl  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=200)
    ap.add_argument('--profile', action='store_true', help='cProfile the call instead of timing it')
    args = ap.parse_args()
    cfg = wl.CONFIG2
    c_opers, c_coeffs, n_opers, n_coeffs, dt = wl.random_pulse_inputs(**cfg)
    omega = wl.random_pulse_omega(dt, cfg['W'])
    S = 1e-3/omega
    basis = ff.Basis.pauli(2)
    H_c, H_n = list(zip(c_opers, c_coeffs)), list(zip(n_opers, n_coeffs))
    if args.profile:
        import cProfile
        import pstats

        def call():
            pulse = ff.PulseSequence(H_c, H_n, dt, basis)
            pulse.get_filter_function(omega)
            ff.infidelity(pulse, S, omega)
        for _ in range(5):
            call()
        prof = cProfile.Profile()
        prof.enable()
        for _ in range(args.reps):
            call()
        prof.disable()
        pstats.Stats(prof).sort_stats('tottime').print_stats(22)
        return
    rows = []
    for i in range(args.reps + 5):
        t0 = time.perf_counter()
        pulse = ff.PulseSequence(H_c, H_n, dt, basis)
        t1 = time.perf_counter()
        pulse.get_filter_function(omega)
        t2 = time.perf_counter()
        ff.infidelity(pulse, S, omega)
        t3 = time.perf_counter()
        stage, enqueue, wait = pulse._resident.timing()
        if i >= 5:
            rows.append((t1 - t0, t2 - t1, t3 - t2, stage, enqueue, wait))
    med = np.median(np.array(rows), axis=0)*1e3
    print(f'construct PulseSequence          {med[0]:.3f} ms')
    print(f'get_filter_function (resident)   {med[1]:.3f} ms   of which inside libffk: pack inputs '
          f'{med[3]:.3f}, enqueue H2D + 6 kernels + D2H {med[4]:.3f}, wait for the stream {med[5]:.3f}; '
          f'Python around it {med[1] - med[3] - med[4] - med[5]:.3f}')
    print(f'infidelity (resident F)          {med[2]:.3f} ms')
    print(f'get_filter_function + infidelity {med[1] + med[2]:.3f} ms (median of {args.reps})')


if __name__ == '__main__':
    main()
