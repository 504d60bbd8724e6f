"""K5 evidence: superoperator.liouville_representation on the FP64 matrix cores (liouville.hip),
device resident and batched over the segments of a pulse as `concatenate` and the gradient call it.

    python tools/time_liouville.py [--batch 512] [--d 4 8 16] [--reps 20]

Prints, per dimension, the time of the whole launch sequence (memset, operand build, basis
conjugation, GEMM) from HIP events, the real FP64 flops of the GEMM
(batch * N * d^2 * N * 2, N = d^2: Hermitian Pauli basis, half of the plain trace's rows) and the rate.
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split and under
`rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE` for the utilisation.
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
    ap.add_argument('--batch', type=int, default=512)
    ap.add_argument('--d', type=int, nargs='*', default=[4, 8, 16])
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--dense-basis', action='store_true',
                    help='rotate the basis by a random unitary: Hermitian, orthonormal, no zero entries (the GEMM form)')
    args = ap.parse_args()
    lib = _lib.load()
    rng = np.random.default_rng(0)
    stream = torch.cuda.current_stream().cuda_stream
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    for e in (e0, e1):
        _lib.check(lib.ffk_event_create(ctypes.byref(e)))
    ms = ctypes.c_float()
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    for d in args.d:
        N, B = d*d, args.batch
        basis = np.asarray(ff.Basis.pauli(int(np.log2(d))))
        if args.dense_basis:
            V = np.linalg.qr(rng.standard_normal((d, d)) + 1j*rng.standard_normal((d, d)))[0]
            basis = V @ basis @ V.conj().T
        U = np.linalg.qr(rng.standard_normal((B, d, d)) + 1j*rng.standard_normal((B, d, d)))[0]
        Ud = torch.from_numpy(U).cuda()
        Cd = torch.from_numpy(np.ascontiguousarray(np.asarray(basis))).cuda()
        out = torch.empty((B, N, N), dtype=torch.float64, device='cuda')
        need = lib.ffk_liouville_workspace_bytes(B, d, N)
        ws = torch.empty(need, dtype=torch.uint8, device='cuda')

        def run():
            _lib.check(lib.ffk_liouville_dev(p(Ud), B, d, p(Cd), N, 1, p(out), p(ws), need,
                                             ctypes.c_void_p(stream)))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        times = []
        for _ in range(args.reps):
            _lib.check(lib.ffk_event_record(e0, ctypes.c_void_p(stream)))
            run()
            _lib.check(lib.ffk_event_record(e1, ctypes.c_void_p(stream)))
            torch.cuda.synchronize()
            _lib.check(lib.ffk_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
            times.append(ms.value)
        t = float(np.median(times))
        # executed by the GEMM since round 4: K = d^2 rows for a Hermitian basis (the pair (a, b), (b, a)
        # of a Hermitian operand contributes twice its stored half); 2 d^2 rows is the plain trace
        flops = B*N*(d*d)*N*2.0
        flops_plain = B*N*(2*d*d)*N*2.0
        # parity on the first elements against the defining trace
        L = out[:2].cpu().numpy()
        C = np.asarray(basis)
        ref = np.einsum('bka,ikl,blm,jma->bij', U[:2].conj(), C, U[:2], C).real
        err = np.abs(L - ref).max()
        # d = 12, 16 with a basis of short operand columns (Pauli, GGM): since round 6 no GEMM runs -- the conjugation
        # kernel contracts with the non-zeros itself; the rate below is then what a GEMM would have had to sustain
        fused = d in (12, 16) and not args.dense_basis and not os.environ.get('FFK_LIOUVILLE_GEMM')
        what = 'fused sparse contraction; a GEMM of' if fused else 'GEMM'
        print(f'd={d:2d} N={N:3d} batch={B}: {t*1e3:9.1f} us per call, {what} {flops/1e9:8.3f} GFLOP '
              f'{"would need" if fused else "executed ->"} {flops/(t*1e-3)/1e12:6.2f} TFLOP/s over the whole call '
              f'({flops_plain/(t*1e-3)/1e12:6.2f} at the plain trace\'s 2 d^2 rows); max abs err {err:.1e}')


if __name__ == '__main__':
    main()
