"""cProfile of BASELINE config 3's whole call (ff.concatenate of 1000 Clifford gates + filter
function, 8192 omega): where the host time goes.   python tools/profile_concatenate.py"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import filter_functions_amd as ff  # noqa: E402
import workloads as wl  # noqa: E402

cfg = wl.CONFIG3
omega = wl.rb_omega(cfg['W'], cfg['T'])
_, cliffords = wl.rb_cliffords(ff, omega, cfg['T'])
draw = wl.rb_draw(cfg['n_gates'], cfg['seed'])
seq = [cliffords[k] for k in draw]


def call():
    total = ff.concatenate(seq)
    return total.get_filter_function(omega)


for _ in range(3):
    call()
ts = []
for _ in range(10):
    t0 = time.perf_counter()
    call()
    ts.append(time.perf_counter() - t0)
print(f'whole call: min {min(ts)*1e3:.2f} ms, median {sorted(ts)[5]*1e3:.2f} ms')
prof = cProfile.Profile()
prof.enable()
for _ in range(10):
    call()
prof.disable()
pstats.Stats(prof).sort_stats('tottime').print_stats(25)
