import sys, time, numpy as np
sys.path.insert(0, '.')
import filter_functions_amd as ff
rng = np.random.default_rng(0)
for d, batch in ((2, 1), (2, 100), (2, 1000), (2, 10000), (4, 1000)):
    U = np.linalg.qr(rng.standard_normal((batch, d, d)) + 1j*rng.standard_normal((batch, d, d)))[0]
    basis = ff.Basis.pauli(int(np.log2(d)))
    ff.liouville_representation(U, basis)
    t0 = time.perf_counter()
    for _ in range(20): ff.liouville_representation(U, basis)
    print(d, batch, (time.perf_counter() - t0)/20*1e3, 'ms')
