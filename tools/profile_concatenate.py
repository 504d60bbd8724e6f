import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import filter_functions_amd as ff
from filter_functions_amd import util
X, Y = util.paulis[1], util.paulis[2]
T = 20.0
omega = 2*np.pi*np.geomspace(1e-2/(7*151*T), 1e2/T, 8192)
X2 = ff.PulseSequence([[X/2, [np.pi/2/T], 'X']], [[X/2, [1], 'X']], [T])
Y2 = ff.PulseSequence([[Y/2, [np.pi/2/T], 'Y']], [[X/2, [1], 'X']], [T])
for p in (X2, Y2): p.cache_control_matrix(omega)
cl = np.array([Y2 @ Y2 @ Y2 @ Y2, X2 @ X2, Y2 @ Y2, Y2 @ Y2 @ X2 @ X2, X2 @ Y2, X2 @ Y2 @ Y2 @ Y2, X2 @ X2 @ X2 @ Y2, X2 @ X2 @ X2 @ Y2 @ Y2 @ Y2, Y2 @ X2, Y2 @ X2 @ X2 @ X2, Y2 @ Y2 @ Y2 @ X2, Y2 @ Y2 @ Y2 @ X2 @ X2 @ X2, X2, X2 @ X2 @ X2, Y2, Y2 @ Y2 @ Y2, X2 @ Y2 @ Y2 @ Y2 @ X2 @ X2 @ X2, X2 @ X2 @ X2 @ Y2 @ Y2 @ Y2 @ X2, X2 @ X2 @ Y2, X2 @ X2 @ Y2 @ Y2 @ Y2, Y2 @ Y2 @ X2, Y2 @ Y2 @ X2 @ X2 @ X2, X2 @ Y2 @ X2, X2 @ Y2 @ Y2 @ Y2 @ X2], dtype=object)
rng = np.random.default_rng(0)
seq = list(cl[rng.integers(0, 24, 1000)])
ff.concatenate(seq).get_filter_function(omega)
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    total = ff.concatenate(seq); F = total.get_filter_function(omega)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(18); print(s.getvalue()[:3500])
