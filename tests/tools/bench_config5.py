"""BASELINE config 5 shaped workload: 4 qubits (d = 16), 13 segments, 18 control and 18 noise
operators, GGM basis, 16384 omega -- the full error-transfer-matrix path
(control matrix -> decay amplitudes -> cumulant function -> exp), device resident.

    python tests/tools/bench_config5.py [--W 16384] [--cpu-sample 128]

The pulse is synthetic (seeded random amplitudes on single-qubit X/Y/Z and nearest-neighbour
ZZ / XX terms); the CPU leg runs the NumPy oracle on a bounded sample of the frequencies.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))

import torch  # noqa: E402

import filter_functions_amd as ff  # noqa: E402
from filter_functions_amd import util  # noqa: E402
from filter_functions_amd.device import DevicePipeline  # noqa: E402


def four_qubit_operators():
    P = util.paulis
    ops = []
    for q in range(4):
        for k in (1, 2, 3):
            ops.append(util.tensor(*[P[k] if i == q else P[0] for i in range(4)]))
    for k in (3, 1):
        for q in range(3):
            ops.append(util.tensor(*[P[k] if i in (q, q + 1) else P[0] for i in range(4)]))
    return np.array(ops)/4.0          # normalised like a 4-qubit Pauli basis element


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--W', type=int, default=16384)
    ap.add_argument('--G', type=int, default=13)
    ap.add_argument('--cpu-sample', type=int, default=128)
    ap.add_argument('--reps', type=int, default=5)
    args = ap.parse_args()
    rng = np.random.default_rng(5)
    ops = four_qubit_operators()
    A = len(ops)
    G, W, d = args.G, args.W, 16
    c_coeffs = rng.standard_normal((A, G))
    n_coeffs = np.ones((A, G))
    dt = 1 - 0.5*rng.random(G)
    basis = ff.Basis.ggm(d)
    omega = np.geomspace(1e-2/dt.sum(), 1e2, W)
    S = np.outer(1e-6*(np.arange(A) + 1), 1/omega)
    pipe = DevicePipeline(ops, c_coeffs, ops, n_coeffs, dt, basis, omega, spectrum=S)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.reps):
            t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)), out

    t_R, _ = timed(lambda: pipe.launch(with_infidelity=False))
    t_G, gamma = timed(pipe.decay_amplitudes)
    t_K, K = timed(lambda: pipe.cumulant_function(gamma))

    def host_expm():
        # the summed cumulant (one d^2 x d^2 real matrix) through ff.error_transfer_matrix:
        # scaling and squaring on the matrix cores (ffk_expm_real) instead of scipy.linalg.expm
        return ff.error_transfer_matrix(cumulant_function=K.sum(dim=0).cpu().numpy()[None])
    t_U, U = timed(host_expm)
    total = t_R + t_G + t_K + t_U
    print(f'config 5 shape: d={d}, G={G}, A={A}, N={d*d}, W={W}')
    print(f'GPU control matrix + filter function : {t_R*1e3:9.3f} ms')
    print(f'GPU decay amplitudes (MFMA f64 GEMM)  : {t_G*1e3:9.3f} ms')
    print(f'GPU cumulant function                 : {t_K*1e3:9.3f} ms')
    print(f'exp of the {d*d}x{d*d} cumulant (GPU, host in/out): {t_U*1e3:9.3f} ms')
    print(f'total error transfer matrix           : {total*1e3:9.3f} ms   '
          f'({G*W*A*d*d/total:.3e} filter-function elements/s)')
    print(f'entanglement infidelity 1 - tr(U)/d^2 = {1 - np.trace(U)/d**2:.6e}')

    # CPU oracle on a bounded frequency sample (strided subset, same pulse)
    import ff_oracle as orc
    n = args.cpu_sample
    sub = np.linspace(0, W - 1, n).astype(int)
    om_s, S_s = omega[sub], S[:, sub]
    H = np.einsum('ijk,il->ljk', ops, c_coeffs)
    t0 = time.perf_counter()
    D, V, Q = orc.diagonalize(H, dt)
    t = np.concatenate(([0], dt.cumsum()))
    R = orc.control_matrix_from_scratch(D, V, Q, om_s, np.asarray(basis), ops, n_coeffs, dt, t)
    t_cR = time.perf_counter() - t0
    t0 = time.perf_counter()
    g_cpu = orc.decay_amplitudes(R, S_s, om_s, np.arange(A))
    t_cG = time.perf_counter() - t0
    t0 = time.perf_counter()
    K_cpu = orc.cumulant_function(g_cpu, np.asarray(basis))
    t_cK = time.perf_counter() - t0
    print(f'CPU oracle on {n} of {W} omega: control matrix {t_cR:.2f} s, decay amplitudes '
          f'{t_cG:.2f} s, cumulant {t_cK:.2f} s  -> extrapolated to {W} omega: '
          f'{(t_cR + t_cG)*W/n + t_cK:.1f} s ({os.cpu_count()} logical CPUs)')
    # parity of the device path on the same sample
    pipe_s = DevicePipeline(ops, c_coeffs, ops, n_coeffs, dt, basis, om_s, spectrum=S_s)
    pipe_s.launch(with_infidelity=False)
    torch.cuda.synchronize()
    R_gpu = pipe_s.control_matrix.cpu().numpy()
    g_gpu = pipe_s.decay_amplitudes()
    K_gpu = pipe_s.cumulant_function(g_gpu).cpu().numpy()
    rel = lambda a, b: np.abs(a - b).max()/np.abs(b).max()
    print(f'parity on the sample: R {rel(R_gpu, R):.1e}, Gamma {rel(g_gpu.cpu().numpy(), g_cpu):.1e}, '
          f'K {rel(K_gpu, K_cpu):.1e}')


if __name__ == '__main__':
    main()
