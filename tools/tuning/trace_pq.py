"""Timeline of one block of the d = 4 matrix-core accumulate kernel (tuning build -DFFK_PQ_CLOCK).

    make -C filter_functions_amd/csrc VARIANT=pqclock VSRCS=ctrl_pq.hip VFLAGS=-DFFK_PQ_CLOCK
    FFK_LIBRARY=build/libffk_pqclock.so python tools/trace_pq.py

Lane 0 of every wavefront of block (0, 0, 0) stamps s_memtime four times per tile: producers at the top
of the iteration, when the tile's numbers are computed, when the slot is free, when the tile is published;
consumers at the loop top, when the tile's flag is seen, when its arithmetic is done, when the slot is
handed back.  Printed per wavefront: median cycles of each phase and of a whole tile.
"""
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


def analyse_waves(w):
    """per CU: life of a block (first wavefront in .. last wavefront out) and the idle time between consecutive blocks"""
    # per launch (a slot of 4096 records): first wavefront in, last wavefront out
    spans = []
    for k in range(w.shape[0]//4096):
        x = w[4096*k:4096*(k + 1)]
        x = x[x[:, 1] > 0]
        if x.shape[0] >= 3000:
            spans.append((int(x[:, 0].min()), int(x[:, 1].max()), x.shape[0]))
    spans.sort()
    if len(spans) > 2:
        st0 = np.array([a for a, _, _ in spans]); en0 = np.array([b for _, b, _ in spans])
        print(f'{len(spans)} launches: first wavefront in -> last out median {np.median(en0 - st0)/100:.2f} us; '
              f'launch to launch {np.median(np.diff(st0))/100:.2f} us (min {np.diff(st0).min()/100:.2f}, max {np.diff(st0).max()/100:.2f})')
    w = w[w[:, 1] > 0]
    start, end = w[:, 0].astype(np.int64), w[:, 1].astype(np.int64)
    hw, xcc = (w[:, 2] & np.uint64(0xffffffff)).astype(np.int64), (w[:, 2] >> np.uint64(32)).astype(np.int64)
    cu = xcc*1024 + ((hw >> 13) & 7)*16 + ((hw >> 8) & 15)
    gaps, lives, first_to_first = [], [], []
    for c in np.unique(cu):
        m = cu == c
        order = np.argsort(start[m])
        st_, en_ = start[m][order], end[m][order]
        # clusters of wavefronts that arrived within 2 us of each other = one block
        cuts = np.nonzero(np.diff(st_) > 200)[0] + 1
        b_start = np.array([x.min() for x in np.split(st_, cuts)])
        b_end = np.array([x.max() for x in np.split(en_, cuts)])
        n_w = np.array([x.size for x in np.split(st_, cuts)])
        ok = n_w == 12
        for i in range(len(b_start) - 1):
            if ok[i] and ok[i + 1]:
                gaps.append(b_start[i + 1] - b_end[i])
                first_to_first.append(b_start[i + 1] - b_start[i])
        lives.extend((b_end - b_start)[ok])
    if 'detail' in sys.argv:
        c = np.unique(cu)[5]
        m = cu == c
        order = np.argsort(start[m])
        t0 = start[m].min()
        blk = (w[m][:, 3] & np.uint64(0xffffffff)).astype(np.int64)[order]
        wv = (w[m][:, 3] >> np.uint64(32)).astype(np.int64)[order]
        for a, b, bl, ww in list(zip(start[m][order] - t0, end[m][order] - t0, blk, wv))[:60]:
            print(f'   start {a/100:8.2f}  end {b/100:8.2f}  block {bl:4d} wave {ww:2d}')
    q = lambda a: f'median {np.median(a)/100:.2f}  p10 {np.percentile(a, 10)/100:.2f}  p90 {np.percentile(a, 90)/100:.2f} us  (n = {len(a)})'
    print(f'{len(np.unique(cu))} CUs, {w.shape[0]} wavefront records')
    print('life of a block on its CU            ', q(np.array(lives)))
    print('CU idle between consecutive blocks   ', q(np.array(gaps)))
    print('block start to next block start      ', q(np.array(first_to_first)))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == 'waves':
        analyse_waves(np.load(sys.argv[2]))       # FFK_BENCH_DUMP_PQ_WAVES=<file> python bench.py ... on a -DFFK_PQ_CLOCK build
        return
    c_opers, c_coeffs, n_opers, n_coeffs, dt = wl.random_pulse_inputs(**wl.CONFIG2)
    omega = wl.random_pulse_omega(dt, 4096)
    _lib.load()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    stream = torch.cuda.current_stream().cuda_stream
    pipe = DevicePipeline(c_opers, c_coeffs, n_opers, n_coeffs, dt, ff.Basis.pauli(2), omega, spectrum=1e-3/omega)
    if 'cu' in sys.argv[1:]:
        # the bench's schedule in small: two passes in flight on two streams, graph replay; then, per CU, the time
        # between the last wavefront of one launch's block leaving and the first wavefront of the next launch's arriving
        pipe2 = DevicePipeline(c_opers, c_coeffs, n_opers, n_coeffs, dt, ff.Basis.pauli(2), omega, spectrum=1e-3/omega)
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        g1, g2 = pipe.graph(), pipe2.graph()
        for _ in range(300):
            g1.launch(s1.cuda_stream)
            g2.launch(s2.cuda_stream)
        torch.cuda.synchronize()
        ring = (ctypes.c_ulonglong*(4*65536))()
        head = ctypes.c_uint(0)
        assert raw.ffk_debug_pq_waves(ring, ctypes.byref(head)) == 0
        analyse_waves(np.array(ring, dtype=np.uint64).reshape(-1, 4))
        return
    for _ in range(2000):
        pipe.launch(stream=stream)
    torch.cuda.synchronize()
    n = raw.ffk_debug_pq_trace_words()
    buf = (ctypes.c_ulonglong*n)()
    assert raw.ffk_debug_pq_trace(buf) == 0
    tr = np.array(buf, dtype=np.uint64).reshape(16, -1)
    st = _lib.stats()
    nw = st['block']//64
    t0 = int(tr[:nw, 0].min())
    print(f'block of {nw} wavefronts, grid ({st["grid_x"]}, {st["grid_y"]}, {st["grid_z"]})')
    if 'blocks' in sys.argv[1:]:
        # when does each block of the LAST launch start and end (100 MHz ticks), and where does it run?
        nb = st['grid_x']*st['grid_y']*st['grid_z']
        bb = (ctypes.c_ulonglong*(3*nb))()
        assert raw.ffk_debug_pq_blocks(bb, nb) == 0
        b = np.array(bb, dtype=np.uint64).reshape(nb, 3)
        start, end = b[:, 0].astype(np.int64), b[:, 1].astype(np.int64)
        first = start.min()
        xcc, hw = (b[:, 2] >> np.uint64(32)).astype(np.int64), (b[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
        cu, se = (hw >> 8) & 15, (hw >> 13) & 7
        places = {(int(x), int(s_), int(c)) for x, s_, c in zip(xcc, se, cu)}
        q = lambda a: f'min {a.min()/100:.2f} median {np.median(a)/100:.2f} max {a.max()/100:.2f} us'
        print(f'{nb} blocks on {len(places)} distinct (XCD, SE, CU) places')
        print(f'start after the first block\'s start: {q(start - first)}')
        print(f'duration of a block:                  {q(end - start)}')
        print(f'end after the first block\'s start:   {q(end - first)}')
        late = np.argsort(start)[-8:]
        print('latest starters (block, start us, duration us, XCD/SE/CU):',
              [(int(i), round((start[i] - first)/100, 2), round((end[i] - start[i])/100, 2), (int(xcc[i]), int(se[i]), int(cu[i]))) for i in late])
        return
    if 'loop' in sys.argv[1:]:
        # whole-loop build (GEN_PQ_LOOP_CLOCK=1 block, -DFFK_PQ_CLOCK): the consumers stamp loop entry and exit only
        for w in range(nw):
            if w < 4:
                stamps = tr[w, 2:].reshape(-1, 12).astype(np.int64)[:, :4]
                used = np.nonzero(stamps[:, 3])[0]
                s = stamps[used] - t0
                print(f'wave {w:2d} producer: tiles {used.size:3d}  first top {s[0, 0]:7d}  first tile published {s[0, 3]:7d}  '
                      f'last published {s[-1, 3]:7d}  waiting for a free slot, median {int(np.median(s[:, 2] - s[:, 1]))}')
            else:
                a, b, n_it = int(tr[w, 2]) - t0, int(tr[w, 3]) - t0, int(tr[w, 4])
                print(f'wave {w:2d} consumer: loop entered at {a:7d}, left at {b:7d}: {n_it} tiles, {(b - a)/max(n_it, 1):7.1f} cycles per tile')
        clock = (int(tr[:nw, 3].max()) - t0)
        print(f'block: last consumer leaves its loop {clock} cycles after the first wavefront started')
        ghz = [(int(tr[w, 5]) - int(tr[w, 0]))/((int(tr[w, 6]) - int(tr[w, 1]))*10.0) for w in range(4, nw)]
        print(f'clock over the consumers\' lifetime (s_memtime / s_memrealtime): {min(ghz):.3f} .. {max(ghz):.3f} GHz')
        return
    for w in range(nw):
        stamps = tr[w, 2:].reshape(-1, 12).astype(np.int64)[:, :4]
        used = np.nonzero(stamps[:, 3])[0]
        if used.size == 0:
            continue
        s = stamps[used] - t0
        role = 'producer' if w < 4 else 'consumer'
        ph = np.diff(s, axis=1)
        gap = s[1:, 0] - s[:-1, 3]
        per = np.diff(s[:, 0])
        print(f'wave {w:2d} {role}: tiles {used.size:3d}  first top {s[0, 0]:7d}  last end {s[-1, 3]:7d}  '
              f'phases median {np.median(ph, axis=0).astype(int)}  between tiles {int(np.median(gap)) if gap.size else 0}  '
              f'tile period median {int(np.median(per)) if per.size else 0}')
    # inside the generated block (GEN_PQ_STAMPS=1 build): top, operands in, set 0 vector done, set 0 matrix done,
    # set 1 operands in, set 1 vector done, set 1 matrix done, hand-over issued
    names = ['entry wait', 'set 0 vector', 'set 0 matrix', 'wait set 1', 'set 1 vector (+ requests)', 'set 1 matrix',
             'hand-over']
    for w in range(4, nw):
        inner = tr[w, 2:].reshape(-1, 12).astype(np.int64)[:, 4:]
        used = np.nonzero(inner[:, 7])[0]
        if used.size < 4:
            continue
        ph = np.diff(inner[used], axis=1)
        nxt = inner[used[1:], 0] - inner[used[:-1], 7]
        print(f'wave {w:2d} inside the block, median cycles: ' +
              ', '.join(f'{n} {int(v)}' for n, v in zip(names, np.median(ph, axis=0))) +
              f'; end of block -> next block top {int(np.median(nxt))}')
    print('per-tile stamps of wave 4 (consumer), first 12 tiles:')
    s = tr[4, 2:].reshape(-1, 12).astype(np.int64)[:, :4]
    for it in np.nonzero(s[:, 3])[0][:12]:
        print(f'  tile {it:3d}:', (s[it] - t0).tolist())
    print('per-tile stamps of wave 0 (producer), first 8 tiles:')
    s = tr[0, 2:].reshape(-1, 12).astype(np.int64)[:, :4]
    for it in np.nonzero(s[:, 3])[0][:8]:
        print(f'  tile {it:3d}:', (s[it] - t0).tolist())


if __name__ == '__main__':
    main()
