"""The runtime-d path (csrc/generic.hip) on a five-qubit register: d = 32, Pauli basis (1024 elements).
    python tools/time_large_d.py [--d 32 --G 100 --A 10 --W 1000]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import filter_functions_amd as ff  # noqa: E402
from filter_functions_amd import numeric  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--d', type=int, default=32)
    ap.add_argument('--G', type=int, default=100)
    ap.add_argument('--A', type=int, default=10)
    ap.add_argument('--W', type=int, default=1000)
    args = ap.parse_args()
    d, G, A, W = args.d, args.G, args.A, args.W
    rng = np.random.default_rng(5)

    def herm(n):
        M = rng.standard_normal((n, d, d)) + 1j*rng.standard_normal((n, d, d))
        M = (M + M.conj().transpose(0, 2, 1))/2
        return M - np.trace(M, axis1=1, axis2=2)[:, None, None]*np.eye(d)/d
    c_opers, n_opers = herm(3), herm(A)
    c_coeffs, n_coeffs = rng.standard_normal((3, G)), rng.random((A, G))
    dt = 1 - rng.random(G)
    omega = np.geomspace(1e-2/dt.sum(), 1e2/dt.min(), W)
    basis = ff.Basis.pauli(int(np.log2(d))) if d & (d - 1) == 0 else ff.Basis.ggm(d)
    H = np.einsum('ijk,il->ljk', c_opers, c_coeffs)

    def best(fn, reps=3):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            out = fn()
            ts.append(time.perf_counter() - t0)
        return min(ts), out
    t_diag, (D, V, Q) = best(lambda: numeric.diagonalize(H, dt))
    t_R, R = best(lambda: numeric.calculate_control_matrix_from_scratch(D, V, Q, omega, basis, n_opers, n_coeffs, dt))
    t_F, F = best(lambda: numeric.calculate_filter_function(R))
    t_L, L = best(lambda: ff.liouville_representation(Q[-1], basis))
    flops = 16.0*d**3*A*G*W
    print(f'd={d} G={G} A={A} W={W} N={len(basis)}')
    print(f'diagonalize {t_diag*1e3:9.2f} ms   control matrix {t_R*1e3:9.2f} ms ({flops/t_R/1e12:.2f} TFLOP/s on the '
          f'two products alone, host arrays in and out)   filter function {t_F*1e3:8.2f} ms   Liouville {t_L*1e3:8.2f} ms')


if __name__ == '__main__':
    main()
