"""cProfile of the two slow routes of the published periodic_driving example (bench.py
bench_published_example): the echo concatenation and the from-scratch evaluation."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import filter_functions_amd as ff  # noqa: E402
import workloads as wl  # noqa: E402

cfg = wl.PERIODIC_DRIVING
atomic, wait, full, omega = wl.periodic_driving(ff)
atomic.cache_filter_function(omega)
not_periodic = ff.concatenate_periodic(atomic, cfg['n_periods'])


def echo():
    return ff.concatenate((wait, not_periodic, wait))


def brute():
    written_out = ff.concatenate((wait, full, wait), calc_filter_function=False)
    return written_out.get_filter_function(omega)


for name, fn in (('echo', echo), ('brute', brute)):
    for _ in range(2):
        t0 = time.perf_counter()
        fn()
        print(f'{name}: {(time.perf_counter() - t0)*1e3:.2f} ms')
    pr = cProfile.Profile()
    pr.enable()
    fn()
    pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(8)
