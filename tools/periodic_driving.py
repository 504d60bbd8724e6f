"""The reference's own timed example (doc/source/examples/periodic_driving.ipynb: 0.0286 s periodic,
0.9008 s standard concatenation, 38.38 s brute force, hardware unstated) on the GPU, all three routes,
with their mutual agreement (the check against the oracle and the reference's own outputs is
tests/test_baseline_configs.py::test_published_example_periodic_driving).

    python tools/periodic_driving.py
"""
import argparse
import functools
import os
import sys
import time
from itertools import repeat

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import filter_functions_amd as ff  # noqa: E402
import workloads as wl  # noqa: E402


print = functools.partial(print, flush=True)


def timed(fn, reps=3):
    best, out = None, None
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best, out


def main():
    argparse.ArgumentParser(description=__doc__).parse_args()
    cfg = wl.PERIODIC_DRIVING
    atomic, wait, full, omega = wl.periodic_driving(ff)
    pub = cfg['published_s']

    def atomic_ff():
        atomic.cleanup('all')
        atomic.cache_filter_function(omega)
        return atomic
    t_atomic, _ = timed(atomic_ff)
    t_periodic, not_periodic = timed(lambda: ff.concatenate_periodic(atomic, cfg['n_periods']))
    t_standard, not_standard = timed(lambda: ff.concatenate(repeat(atomic, cfg['n_periods'])), reps=2)
    t_echo, echo = timed(lambda: ff.concatenate((wait, not_periodic, wait)))

    def brute():
        echo_full = ff.concatenate((wait, full, wait), calc_filter_function=False)
        return echo_full, echo_full.get_filter_function(omega)
    t_brute, (echo_full, F_brute) = timed(brute, reps=2)
    rel = lambda a, b: float(np.abs(a - b).max()/np.abs(b).max())
    F_echo = echo.get_filter_function(omega)
    print(f'segments of the written-out sequence: {len(echo_full)}, frequencies: {len(omega)}')
    for name, t, key in (('atomic filter function', t_atomic, 'atomic_filter_function'),
                         ('concatenate_periodic (10000 periods)', t_periodic, 'concatenate_periodic'),
                         ('concatenate (10000 pulse objects)', t_standard, 'concatenate_standard'),
                         ('echo concatenation', t_echo, 'echo_concatenation'),
                         ('brute force (200002 segments from scratch)', t_brute, 'brute_force')):
        print(f'{name:45s} {t*1e3:10.2f} ms   reference notebook {pub[key]*1e3:10.1f} ms   x{pub[key]/t:8.0f}')
    print('periodic vs standard concatenation :', rel(not_periodic.get_filter_function(omega),
                                                      not_standard.get_filter_function(omega)))
    print('concatenated echo vs brute force   :', rel(F_echo, F_brute))


if __name__ == '__main__':
    main()
