"""Time the decay-amplitude GEMM and the cumulant contraction on device-resident data.

    python tools/bench_etm.py [--d 16 --A 18 --W 16384] [--reps 10]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--d', type=int, default=16)
    ap.add_argument('--A', type=int, default=18)
    ap.add_argument('--W', type=int, default=16384)
    ap.add_argument('--reps', type=int, default=10)
    args = ap.parse_args()
    d, A, W = args.d, args.A, args.W
    N = d*d
    lib = _lib.load()
    dev = torch.device('cuda:0')
    gen = torch.Generator(device=dev).manual_seed(1)
    R = torch.randn(A, N, W, 2, dtype=torch.float64, device=dev, generator=gen)
    S = torch.zeros(A, W, 2, dtype=torch.float64, device=dev)
    S[..., 0] = torch.rand(A, W, dtype=torch.float64, device=dev, generator=gen)
    omega = torch.linspace(0.01, 100.0, W, dtype=torch.float64, device=dev)
    idx = torch.arange(A, dtype=torch.int32, device=dev)
    gamma = torch.empty(A, N, N, dtype=torch.float64, device=dev)
    wsb = lib.ffk_decay_amplitudes_workspace_bytes(1, N, W, A, 2)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    vp = ctypes.c_void_p

    def run_gamma():
        _lib.check(lib.ffk_decay_amplitudes_dev(vp(R.data_ptr()), 1, A, N, W, vp(S.data_ptr()), 2,
                                                vp(omega.data_ptr()), vp(idx.data_ptr()), A,
                                                vp(gamma.data_ptr()), vp(ws.data_ptr()), wsb, None))

    basis = ff.Basis.ggm(d) if d & (d - 1) else ff.Basis.pauli(int(np.log2(d)))
    C = torch.from_numpy(np.ascontiguousarray(np.asarray(basis)).view(np.float64)).to(dev)
    K = torch.empty_like(gamma)
    wsb2 = lib.ffk_cumulant_function_workspace_bytes(A, N, d)
    ws2 = torch.empty(max(wsb2, 16), dtype=torch.uint8, device=dev)

    def run_cumulant():
        _lib.check(lib.ffk_cumulant_function_dev(vp(gamma.data_ptr()), A, N, d, vp(C.data_ptr()), 0,
                                                 vp(K.data_ptr()), vp(ws2.data_ptr()), wsb2, None))

    for name, fn, flops in (('decay amplitudes', run_gamma, 2.0*A*N*N*2*W),
                            ('cumulant function', run_cumulant, 8.0*A*4*float(N)**3)):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t = float(np.median(ts))
        print(f'{name:18s} d={d} A={A} W={W}: {t:8.3f} ms  {flops/t/1e9:8.2f} TFLOP/s')
        if name.startswith('decay') and hasattr(lib, 'ffk_debug_dg_clock'):
            # tuning build -DFFK_DG_CLOCK: shader-clock and 100 MHz ticks summed over the GEMM's wavefronts
            c = (ctypes.c_ulonglong*3)()
            lib.ffk_debug_dg_clock(c)
            print(f'  in-kernel clock of the GEMM wavefronts: {c[0]/max(c[1], 1)*100:.0f} MHz '
                  f'({c[2]} wavefronts, {c[1]/max(c[2], 1)/100:.1f} us each)')
            if hasattr(lib, 'ffk_debug_dg_trace'):
                # per-block start / end / hardware id of the LAST launch: who ran where and when
                nblk = 8192
                tr = (ctypes.c_ulonglong*(3*nblk))()
                lib.ffk_debug_dg_trace(tr, nblk)
                tr = np.array(tr, dtype=np.uint64).reshape(nblk, 3)
                tr = tr[tr[:, 1] > 0]
                t0 = tr[:, 0].min()
                start = (tr[:, 0] - t0)/100.0
                end = (tr[:, 1] - t0)/100.0
                hw = tr[:, 2] & 0xffffffff
                xcc = (tr[:, 2] >> 32) & 0xf
                cu = (hw >> 8) & 0xf
                sh = (hw >> 12) & 0x1
                se = (hw >> 13) & 0x7
                simd = (hw >> 4) & 0x3
                print(f'  trace: {len(tr)} working blocks, span {end.max():.0f} us')
                edges = np.linspace(0, end.max(), 21)
                for lo, hi in zip(edges[:-1], edges[1:]):
                    mid = 0.5*(lo + hi)
                    live = (start <= mid) & (end > mid)
                    per_xcc = [int((live & (xcc == x)).sum()) for x in range(8)]
                    print(f'    t = {mid:7.1f} us: {int(live.sum()):5d} wavefronts live, per XCD {per_xcc}')
                slot = xcc*4096 + se*512 + sh*256 + cu*4 + simd
                print(f'  distinct (XCD, SE, SH, CU, SIMD) slots used: {len(np.unique(slot))}; '
                      f'distinct (XCD, SE, SH, CU): {len(np.unique(slot >> 2))}')
                first = start < 5.0
                print(f'  blocks started in the first 5 us: {int(first.sum())}')
    # spot check against float64 torch on one operator
    wgt = torch.zeros(W, dtype=torch.float64, device=dev)
    wgt[:-1] += 0.5*(omega[1:] - omega[:-1])
    wgt[1:] += 0.5*(omega[1:] - omega[:-1])
    for a in sorted({0, A//2, A - 1}):
        Rc = torch.view_as_complex(R[a])
        ref = torch.real((Rc.conj()*(S[a, :, 0]*wgt/(2*np.pi))) @ Rc.T)
        print(f'operator {a}: max rel err vs torch:', float((gamma[a] - ref).abs().max()/ref.abs().max()))


if __name__ == '__main__':
    main()
