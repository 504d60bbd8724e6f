"""Time the second-order filter function at the BASELINE config-2 shape (d=4, 256 segments, 3 noise
operators, 4096 frequencies: F2 is (3,3,16,16,4096) c128 = 151 MB) and check a frequency subsample
against the oracle.  Usage: python tests/tools/bench_second_order.py [W] [reps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import ff_oracle as orc  # noqa: E402
import filter_functions_amd as ff  # noqa: E402
from filter_functions_amd import numeric  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
d, G, A = 4, 256, 3
rng = np.random.default_rng(42)


def herm(n):
    M = rng.standard_normal((n, d, d)) + 1j*rng.standard_normal((n, d, d))
    M = (M + M.conj().transpose(0, 2, 1))/2
    return M - np.trace(M, axis1=1, axis2=2)[:, None, None]*np.eye(d)/d


c_opers, n_opers = herm(3), herm(A)
c_coeffs = rng.standard_normal((3, G))
n_coeffs = rng.random((A, G))
dt = 1 - rng.random(G)
omega = np.geomspace(1e-2/dt.sum(), 1e2/dt.min(), W)
pulse = ff.PulseSequence(list(zip(c_opers, c_coeffs)), list(zip(n_opers, n_coeffs)), dt,
                         ff.Basis.pauli(2))
pulse.diagonalize()
args = (pulse.eigvals, pulse.eigvecs, pulse.propagators, omega, pulse.basis, pulse.n_opers,
        pulse.n_coeffs, pulse.dt)
F2 = numeric.calculate_second_order_filter_function_from_scratch(*args)
best = 1e9
for _ in range(reps):
    t0 = time.perf_counter()
    F2 = numeric.calculate_second_order_filter_function_from_scratch(*args)
    best = min(best, time.perf_counter() - t0)
flops = 8.0*(A*16)**2*d*d*G*W
print(f'W={W}: {best*1e3:.1f} ms wall incl. PCIe ({F2.nbytes/1e6:.0f} MB out), '
      f'{flops/best/1e12:.2f} TFLOP/s (second contraction only)')
sel = np.linspace(0, W - 1, 24).astype(int)
t0 = time.perf_counter()
ref = orc.second_order_filter_function(pulse.eigvals, pulse.eigvecs, pulse.propagators, omega[sel],
                                       np.asarray(pulse.basis), pulse.n_opers, pulse.n_coeffs, dt)
t_cpu = time.perf_counter() - t0
err = np.abs(F2[..., sel] - ref).max()/np.abs(ref).max()
print(f'oracle on {len(sel)} frequencies: {t_cpu:.2f} s ({t_cpu/len(sel)*W:.0f} s for all); '
      f'rel err {err:.2e}')
