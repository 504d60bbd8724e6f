"""Where a wavefront of the d = 12 / 16 accumulate kernel (ctrl_mfma.hip) spends its cycles: phase sums
from a -DFFK_MFMA_CLOCK build.

    make -C filter_functions_amd/csrc VARIANT=mclock VSRCS="ctrl_mfma.hip" VFLAGS="-DFFK_MFMA_CLOCK"
    FFK_LIBRARY=build/libffk_mclock.so python tools/trace_mfma_phases.py [--reps 5]
"""
import argparse
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import filter_functions_amd as ff  # noqa: E402
import workloads as wl  # noqa: E402
from filter_functions_amd import _lib  # noqa: E402
from filter_functions_amd.device import DevicePipeline  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=5)
    args = ap.parse_args()
    lib = _lib.load()
    fn = getattr(lib, 'ffk_debug_mfma_phases', None)
    if fn is None:
        sys.exit('this library was not built with -DFFK_MFMA_CLOCK')
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    W = wl.CONFIG5['W']
    omega = np.logspace(-2, 2, W)
    qft = wl.qft_pulse(ff)
    A = len(qft.n_opers)
    S = np.outer(1e-6*(np.arange(A) + 1), 1/omega)
    pipe = DevicePipeline(qft.c_opers, qft.c_coeffs, qft.n_opers, qft.n_coeffs, qft.dt, qft.basis,
                          omega, spectrum=S, device=torch.device('cuda:0'))
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        pipe.launch(stream=stream)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong*8)()
    fn(out, 1)
    for _ in range(args.reps):
        pipe.launch(stream=stream)
    torch.cuda.synchronize()
    fn(out, 0)
    v = np.array(list(out), dtype=float)
    steps = v[5]
    names = ['wait: barrier that frees the tile', 'staging loads issued + generation', 'park staged operands',
             'wait: barrier that publishes the tile', 'contraction']
    total = v[:5].sum()
    print(f'{int(steps)} wavefront-segment steps over {args.reps} passes; shader-clock cycles per step:')
    for n, c in zip(names, v[:5]):
        print(f'  {n:40s} {c/steps:10.0f}  ({100*c/total:5.1f} %)')
    print(f'  {"sum":40s} {total/steps:10.0f}')
    print(f'  of the second line, issuing the staging loads: {v[6]/steps:.0f}')


if __name__ == '__main__':
    main()
