"""Config 5 alone (examples/qft.py: d = 16, 13 segments, 18 noise operators, 16384 omega): control
matrix + F + infidelity -> decay amplitudes -> cumulant function, device resident.  Wall time per
pass; run it under `rocprofv3 --kernel-trace --stats` for the per-kernel split.

    python tools/time_config5.py [--reps 20]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import filter_functions_amd as ff  # noqa: E402
import workloads as wl  # noqa: E402
from filter_functions_amd.device import DevicePipeline  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    args = ap.parse_args()
    W = wl.CONFIG5['W']
    omega = np.logspace(-2, 2, W)
    qft = wl.qft_pulse(ff)
    A = len(qft.n_opers)
    S = np.outer(1e-6*(np.arange(A) + 1), 1/omega)
    device = torch.device('cuda:0')
    pipe = DevicePipeline(qft.c_opers, qft.c_coeffs, qft.n_opers, qft.n_coeffs, qft.dt, qft.basis,
                          omega, spectrum=S, device=device)
    ts = torch.cuda.Stream(device=device)
    stream = ts.cuda_stream

    def one_pass():
        with torch.cuda.stream(ts):
            pipe.launch(stream=stream)
            gamma = pipe.decay_amplitudes(stream=stream)
            K = pipe.cumulant_function(gamma, stream=stream)
            return K.sum(dim=0)

    for _ in range(3):
        one_pass()
    torch.cuda.synchronize()
    times = []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        one_pass()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0)*1e3)
    print(f'config 5 pass (no exp, no copy to the host): median {np.median(times):.3f} ms, '
          f'min {np.min(times):.3f} ms over {args.reps} passes')

    # the tail: sum over the operators -> exp -> host, the round-6 way (all in HBM) and the earlier one (sum to the
    # host, ffk_expm_real from and to host memory)
    def tail_device(K):
        return pipe.error_transfer_matrix(K, stream=stream).cpu().numpy()

    def tail_host(K):
        return ff.error_transfer_matrix(cumulant_function=K.sum(dim=0).cpu().numpy()[None])

    with torch.cuda.stream(ts):
        pipe.launch(stream=stream)
        K = pipe.cumulant_function(pipe.decay_amplitudes(stream=stream), stream=stream)
        for name, tail in (('in HBM', tail_device), ('through the host', tail_host)):
            for _ in range(3):
                U = tail(K)
            torch.cuda.synchronize()
            times = []
            for _ in range(args.reps):
                t0 = time.perf_counter()
                U = tail(K)
                times.append((time.perf_counter() - t0)*1e3)
            print(f'sum over operators + exp + copy to the host, {name}: median {np.median(times):.3f} ms, '
                  f'min {np.min(times):.3f} ms; trace {np.trace(U):.12f}')


if __name__ == '__main__':
    main()
