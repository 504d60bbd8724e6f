"""Kernel breakdown of a long single-qubit sequence evaluated from scratch (the written-out
200 002-segment sequence of the reference's periodic_driving example, d=2, 2 noise ops, 500 omega):
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -- python3 tools/profile_long_sequence.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import filter_functions_amd as ff  # noqa: E402
import workloads as wl  # noqa: E402

atomic, wait, full, omega = wl.periodic_driving(ff)
written_out = ff.concatenate((wait, full, wait), calc_filter_function=False)
for i in range(6):
    written_out.cleanup('all')
    t0 = time.perf_counter()
    written_out.get_filter_function(omega)
    t1 = time.perf_counter()
    timing = written_out._resident.timing() if getattr(written_out, '_resident', None) else None
    print(f'pass {i}: {1e3*(t1 - t0):.2f} ms', 'stage/enqueue/wait ms:',
          None if timing is None else [round(1e3*x, 3) for x in timing], flush=True)

if '--cprofile' in sys.argv:
    import cProfile
    import pstats
    prof = cProfile.Profile()
    prof.enable()
    for _ in range(5):
        written_out.cleanup('all')
        written_out.get_filter_function(omega)
    prof.disable()
    pstats.Stats(prof).sort_stats('tottime').print_stats(18)
    prof = cProfile.Profile()
    prof.enable()
    for _ in range(5):
        ff.concatenate((wait, full, wait), calc_filter_function=False)
    prof.disable()
    pstats.Stats(prof).sort_stats('tottime').print_stats(12)
