"""Sweep the segment-chunk count of ffk::ctrl_accumulate on one workload and report the kernel
time (HIP events around the accumulate launch) and the whole control-matrix time.

    python tools/tune_accumulate.py [--d 4 --G 256 --A 3 --W 4096] [--chunks 8 12 16 ...]
"""
import argparse
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import filter_functions_amd as ff  # noqa: E402
from filter_functions_amd import _lib  # noqa: E402
from filter_functions_amd.device import DevicePipeline  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--d', type=int, default=4)
    ap.add_argument('--G', type=int, default=256)
    ap.add_argument('--A', type=int, default=3)
    ap.add_argument('--W', type=int, default=4096)
    ap.add_argument('--reps', type=int, default=30)
    ap.add_argument('--variant', type=int, default=0, help='0 default, 1 one-wave kernel (d <= 4), 2 no in-block segment split')
    ap.add_argument('--chunks', type=int, nargs='*', default=[0, 4, 8, 11, 12, 16, 22, 32, 43, 64])
    args = ap.parse_args()
    d, G, A, W = args.d, args.G, args.A, args.W
    rng = np.random.default_rng(42)

    def herm(n):
        M = rng.standard_normal((n, d, d)) + 1j*rng.standard_normal((n, d, d))
        M = (M + M.conj().transpose(0, 2, 1))/2
        return M - np.trace(M, axis1=1, axis2=2)[:, None, None]*np.eye(d)/d
    c_opers, n_opers = herm(3), herm(A)
    c_coeffs, n_coeffs = rng.standard_normal((3, G)), rng.random((A, G))
    dt = 1 - rng.random(G)
    omega = np.geomspace(1e-2/dt.sum(), 1e2/dt.min(), W)
    basis = ff.Basis.pauli(int(np.log2(d))) if d in (2, 4, 8, 16) else ff.Basis.ggm(d)
    lib = _lib.load()
    _lib.check(lib.ffk_set_accumulate_variant(args.variant))
    e0, e1, t0, t1 = (ctypes.c_void_p() for _ in range(4))
    for e in (e0, e1, t0, t1):
        _lib.check(lib.ffk_event_create(ctypes.byref(e)))
    stream = torch.cuda.current_stream().cuda_stream
    ms = ctypes.c_float()
    print(f'd={d} G={G} A={A} W={W}')
    for chunks in args.chunks:
        _lib.check(lib.ffk_set_segment_chunks(chunks))
        pipe = DevicePipeline(c_opers, c_coeffs, n_opers, n_coeffs, dt, basis, omega,
                              spectrum=1e-3/omega)
        for _ in range(5):
            pipe.launch(stream=stream)
        torch.cuda.synchronize()
        acc, tot = [], []
        for _ in range(args.reps):
            _lib.check(lib.ffk_set_accumulate_events(e0, e1))
            _lib.check(lib.ffk_event_record(t0, ctypes.c_void_p(stream)))
            pipe.launch(stream=stream)
            _lib.check(lib.ffk_event_record(t1, ctypes.c_void_p(stream)))
            torch.cuda.synchronize()
            _lib.check(lib.ffk_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
            acc.append(ms.value)
            _lib.check(lib.ffk_event_elapsed_ms(t0, t1, ctypes.byref(ms)))
            tot.append(ms.value)
        st = _lib.stats()
        _lib.check(lib.ffk_set_accumulate_events(None, None))
        print(f'chunks={st["chunks"]:4d} (req {chunks:3d}) grid=({st["grid_x"]},{st["grid_y"]},{st["grid_z"]}) '
              f'block={st["block"]} accumulate: med {np.median(acc)*1e3:8.1f} us min {np.min(acc)*1e3:8.1f} us  '
              f'pipeline: med {np.median(tot)*1e3:8.1f} us  '
              f'-> {st["accumulate_flops"]/np.median(acc)/1e9:6.2f} TFLOP/s')
        del pipe
    lib.ffk_set_segment_chunks(0)


if __name__ == '__main__':
    main()
