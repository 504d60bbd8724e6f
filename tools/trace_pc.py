"""Per-wavefront timeline of the d = 4 accumulate kernel (tuning build -DFFK_PC_CLOCK).

    make -C filter_functions_amd/csrc -j8 VARIANT=pcclock VFLAGS=-DFFK_PC_CLOCK
    FFK_LIBRARY=build/libffk_pcclock.so python tools/trace_pc.py [--warm 2000]

Every wavefront stamps s_memtime after each barrier (step top) and when its work of the step is
done (in front of the barrier).  Printed: the in-kernel clock, where the wavefronts sit (SIMD census
by role), and per role the share of a step spent working vs waiting at the barrier, the prologue and
the epilogue.  The instrumentation itself costs time (one s_memtime + s_waitcnt per stamp): read
shares, not absolute microseconds.
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
    ap.add_argument('--G', type=int, default=256)
    ap.add_argument('--A', type=int, default=3)
    ap.add_argument('--W', type=int, default=4096)
    ap.add_argument('--warm', type=int, default=3000)
    ap.add_argument('--waves', type=int, default=16, help='wavefronts per block')
    ap.add_argument('--dump', default=None)
    args = ap.parse_args()
    d, G, A, W = 4, args.G, args.A, args.W
    rng = np.random.default_rng(42)

    def herm(n):
        M = rng.standard_normal((n, d, d)) + 1j*rng.standard_normal((n, d, d))
        M = (M + M.conj().transpose(0, 2, 1))/2
        return M - np.trace(M, axis1=1, axis2=2)[:, None, None]*np.eye(d)/d
    c_opers, n_opers = herm(3), herm(A)
    c_coeffs, n_coeffs = rng.standard_normal((3, G)), rng.random((A, G))
    dt = 1 - rng.random(G)
    omega = np.geomspace(1e-2/dt.sum(), 1e2/dt.min(), W)
    lib = _lib.load()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    stream = torch.cuda.current_stream().cuda_stream
    pipe = DevicePipeline(c_opers, c_coeffs, n_opers, n_coeffs, dt, ff.Basis.pauli(2), omega,
                          spectrum=1e-3/omega)
    for _ in range(args.warm):
        pipe.launch(stream=stream)
    torch.cuda.synchronize()
    st = _lib.stats()
    n_blocks = st['grid_x']*st['grid_y']*st['grid_z']
    nw = st['block']//64
    L = raw.ffk_debug_pc_trace_len()
    n_slots = min(n_blocks, 1024)*16
    buf = np.zeros((n_slots, L), dtype=np.uint64)
    rc = raw.ffk_debug_pc_trace(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(n_slots))
    assert rc == 0
    tr = buf.reshape(-1, 16, L)[:n_blocks, :nw].astype(np.int64)
    if args.dump:
        np.save(args.dump, tr)
    steps = (L - 8)//2
    hw = tr[..., 0]
    simd = (hw >> 4) & 3
    cu = (hw >> 8) & 15
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 7
    xcc = (hw >> 32) & 15
    role = tr[..., 1]
    t0, r0 = tr[..., 2], tr[..., 3]
    loop0 = tr[..., 4]
    tops = tr[..., 5:5 + 2*steps:2]
    dones = tr[..., 6:6 + 2*steps:2]
    loop1 = tr[..., 5 + 2*steps]
    t1, r1 = tr[..., 6 + 2*steps], tr[..., 7 + 2*steps]
    n_steps = int((tops[0, 0] != 0).sum())
    tops, dones = tops[..., :n_steps], dones[..., :n_steps]
    clock = (t1 - t0)/np.maximum(r1 - r0, 1)*100.0
    print(f'grid {st["grid_x"]}x{st["grid_y"]}x{st["grid_z"]} block {st["block"]}: {n_blocks} blocks, '
          f'{nw} waves each, {n_steps} steps per wave')
    print(f'in-kernel clock (median over waves): {np.median(clock):.0f} MHz  '
          f'[{np.percentile(clock, 5):.0f} .. {np.percentile(clock, 95):.0f}]')
    life = (t1 - t0)
    print(f'wave lifetime: median {np.median(life):.0f} cycles = {np.median(life)/np.median(clock):.1f} us; '
          f'block start spread (100 MHz ticks): {(r0.min(axis=1).max() - r0.min())/100.0:.2f} us, '
          f'kernel span {(r1.max() - r0.min())/100.0:.1f} us')
    # SIMD census
    print('SIMD census per block (producer SIMD ids of wave 0,5,10,15 / consumers per SIMD):')
    uniq = {}
    for b in range(n_blocks):
        key = tuple(np.bincount(simd[b][role[b] == 0], minlength=4)) + tuple(np.bincount(simd[b][role[b] > 0], minlength=4))
        uniq[key] = uniq.get(key, 0) + 1
    for k, v in sorted(uniq.items(), key=lambda kv: -kv[1])[:6]:
        print(f'   producers/SIMD {k[:4]} consumers/SIMD {k[4:]}: {v} blocks')
    print(f'   wave -> SIMD of block 0: {simd[0].tolist()}')
    cukey = xcc*1000 + se*100 + sh*10 + cu
    per_cu = np.array([len(np.unique(cukey[b])) for b in range(n_blocks)])
    print(f'   distinct CUs per block: {np.unique(per_cu).tolist()}; blocks per CU: '
          f'{np.unique(np.unique(cukey[:, 0], return_counts=True)[1]).tolist()}')
    step_len = np.diff(tops, axis=-1)                       # top to top
    work = dones - tops
    wait = tops[..., 1:] - dones[..., :-1]
    for name, mask in (('producer', role == 0), ('consumer', role > 0)):
        w_, s_, b_ = work[mask], step_len[mask], wait[mask]
        print(f'{name}: work per step median {np.median(w_):.0f} cycles (p10 {np.percentile(w_, 10):.0f}, '
              f'p90 {np.percentile(w_, 90):.0f}); barrier wait median {np.median(b_):.0f} '
              f'(p10 {np.percentile(b_, 10):.0f}, p90 {np.percentile(b_, 90):.0f}); step {np.median(s_):.0f}')
        print(f'   prologue (entry -> first top) {np.median((loop0 - t0)[mask]):.0f} cycles, loop '
              f'{np.median((loop1 - loop0)[mask]):.0f}, epilogue {np.median((t1 - loop1)[mask]):.0f}')
    # who arrives last at the barrier?
    last = np.argmax(dones, axis=1)                         # (blocks, steps): wave index
    last_role = np.take_along_axis(role, last.reshape(n_blocks, -1)[:, :1]*0 + last.reshape(n_blocks, -1), axis=1) \
        if False else np.array([[role[b, last[b, s]] for s in range(n_steps)] for b in range(n_blocks)])
    print(f'last arriver at the barrier is a producer in {np.mean(last_role == 0)*100:.0f} % of the steps')
    first = np.min(dones, axis=1)
    lastt = np.max(dones, axis=1)
    print(f'arrival skew at the barrier (last - first done): median {np.median(lastt - first):.0f} cycles; '
          f'release latency (min next top - last done): median {np.median(tops[..., 1:].min(axis=1) - lastt[:, :-1]):.0f}')
    # per-step profile of one block
    b = n_blocks//2
    print(f'block {b}, per step: step length | producer work (4) | consumer work min..max')
    for s in range(n_steps - 1):
        pw = work[b][role[b] == 0][:, s]
        cw = work[b][role[b] > 0][:, s]
        print(f'   {s:2d}: {int(np.median(step_len[b][:, s])):6d} | {pw.tolist()} | {cw.min()}..{cw.max()}')


if __name__ == '__main__':
    main()
