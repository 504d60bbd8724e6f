"""BASELINE config 3 (1000-gate randomized-benchmarking sequence by concatenation, 8192 omega): the
whole Python call and its parts.    python tools/time_config3.py [--profile]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import filter_functions_amd as ff  # noqa: E402
import workloads as wl  # noqa: E402
from filter_functions_amd import pulse_sequence as ps  # noqa: E402

cfg = wl.CONFIG3
omega = wl.rb_omega(cfg['W'], cfg['T'])
_, cliffords = wl.rb_cliffords(ff, omega, cfg['T'])
draw = wl.rb_draw(cfg['n_gates'], cfg['seed'])
seq = [cliffords[k] for k in draw]


def whole():
    total = ff.concatenate(seq)
    return total.get_filter_function(omega)


for _ in range(5):
    F = whole()
ts, host = [], []
for _ in range(30):
    t0 = time.perf_counter()
    whole()
    ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    ps._concatenate_distinct(*ps._validated_sequence(seq))
    host.append(time.perf_counter() - t0)
print(f'whole call: min {min(ts)*1e3:.3f} ms, median {np.median(ts)*1e3:.3f} ms; '
      f'host bookkeeping (validate + merge tables) alone: min {min(host)*1e3:.3f} ms')
# the device call alone
from filter_functions_amd import numeric  # noqa: E402
pulses, distinct, first, index = ps._validated_sequence(seq)
residents = [p._resident for p in distinct]
taus = [p.tau for p in distinct]
dev = []
for _ in range(30):
    t0 = time.perf_counter()
    numeric.concatenate_sequence_resident(residents, taus, index, distinct[0].basis, which='total',
                                          return_filter_function=True)
    dev.append(time.perf_counter() - t0)
print(f'ffk_concatenate_sequence_resident (R and F to the host): min {min(dev)*1e3:.3f} ms')
if '--profile' in sys.argv:
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        whole()
    pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(18)
