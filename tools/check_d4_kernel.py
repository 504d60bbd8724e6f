"""Parity of the d = 4 accumulate kernel (ctrl_pq.hip) on random pulses of many shapes -- one, two and three
operators per block, part-filled frequency tiles, one segment, two-sided grids through zero: control matrix
against the oracle and against the symmetric kernel (ffk_set_accumulate_variant 2).

    python tools/check_d4_kernel.py
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    sys.path.insert(0, p)

import ff_oracle as orc  # noqa: E402
import filter_functions_amd as ff  # noqa: E402
from filter_functions_amd import _lib, numeric  # noqa: E402


def case(rng, G, A, W, two_sided=False):
    d = 4

    def herm(n):
        M = rng.standard_normal((n, d, d)) + 1j*rng.standard_normal((n, d, d))
        M = (M + M.conj().transpose(0, 2, 1))/2
        return M - np.trace(M, axis1=1, axis2=2)[:, None, None]*np.eye(d)/d
    c_opers, n_opers = herm(2), herm(A)
    c_coeffs, n_coeffs = rng.standard_normal((2, G)), rng.random((A, G))
    dt = 1 - rng.random(G)
    omega = np.geomspace(1e-2/dt.sum(), 1e2/dt.min(), W)
    if two_sided:
        omega = np.concatenate([-omega[::-1][:W//2], [0.0, 1e-10], omega])[:W]
    return c_opers, c_coeffs, n_opers, n_coeffs, dt, omega


def main():
    lib = _lib.load()
    rng = np.random.default_rng(5)
    basis = ff.Basis.pauli(2)
    worst = 0.0
    for G, A, W, ts in [(1, 1, 7, False), (3, 2, 64, False), (5, 3, 100, True), (9, 3, 65, False),
                        (17, 4, 300, True), (40, 7, 129, False), (256, 3, 4096, False), (64, 3, 1000, True),
                        (33, 1, 512, False), (100, 2, 700, False), (256, 6, 512, False)]:
        c_opers, c_coeffs, n_opers, n_coeffs, dt, omega = case(rng, G, A, W, ts)
        pulse = ff.PulseSequence(list(zip(c_opers, c_coeffs)), list(zip(n_opers, n_coeffs)), dt, basis)
        pulse.diagonalize()
        args = (pulse.eigvals, pulse.eigvecs, pulse.propagators, omega, basis, pulse.n_opers, pulse.n_coeffs,
                pulse.dt)
        out = {}
        for variant in (2, 0):
            _lib.check(lib.ffk_set_accumulate_variant(variant))
            out[variant] = numeric.calculate_control_matrix_from_scratch(*args)
        _lib.check(lib.ffk_set_accumulate_variant(0))
        H = orc.hamiltonian(pulse.c_opers, pulse.c_coeffs)
        D, V, Q = orc.diagonalize(H, dt)
        ref = orc.control_matrix_from_scratch(D, V, Q, omega, np.asarray(basis), pulse.n_opers, pulse.n_coeffs,
                                              dt) if G*W <= 300000 else out[2]
        sc = np.abs(ref).max(axis=(1, 2), keepdims=True)
        e0 = (np.abs(out[2] - ref)/sc).max()
        e6 = (np.abs(out[0] - ref)/sc).max()
        e06 = (np.abs(out[0] - out[2])/sc).max()
        worst = max(worst, e6)
        print(f'G={G:4d} A={A} W={W:5d} two-sided={ts!s:5}  symmetric vs oracle {e0:.2e}  pq vs oracle {e6:.2e}  '
              f'pq vs symmetric {e06:.2e}', flush=True)
    print('worst pq error', worst)
    assert worst < 1e-11


if __name__ == '__main__':
    main()
