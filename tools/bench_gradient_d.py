"""Time gradient.infidelity_derivative for a given dimension.  Usage: bench_gradient_d.py d G A H W"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import filter_functions_amd as ff  # noqa: E402
from filter_functions_amd import gradient  # noqa: E402

d, G, A, H, W = (int(x) for x in sys.argv[1:6])
rng = np.random.default_rng(0)


def herm(n):
    M = rng.standard_normal((n, d, d)) + 1j*rng.standard_normal((n, d, d))
    return M + M.conj().transpose(0, 2, 1)


pulse = ff.PulseSequence(list(zip(herm(H), rng.standard_normal((H, G)))),
                         list(zip(herm(A), rng.random((A, G)) + 0.1)), rng.random(G) + 0.2,
                         ff.Basis.ggm(d))
omega = np.geomspace(1e-2, 50, W)
S = 1e-3/omega
pulse.diagonalize()
gradient.infidelity_derivative(pulse, S, omega)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    g = gradient.infidelity_derivative(pulse, S, omega)
    best = min(best, time.perf_counter() - t0)
print(f'd={d} G={G} A={A} H={H} W={W}: {best*1e3:.2f} ms  {g.shape}')
