"""Phase timeline of the fused Liouville conjugation's blocks (build: make VARIANT=lvtrace VSRCS=liouville.hip
VFLAGS=-DFFK_LV_TRACE; run with FFK_LIBRARY=build/libffk_lvtrace.so)."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import filter_functions_amd as ff  # noqa: E402
from filter_functions_amd import _lib  # noqa: E402

lib = _lib.load()
d, B = 16, 512
N = d*d
rng = np.random.default_rng(0)
U = np.linalg.qr(rng.standard_normal((B, d, d)) + 1j*rng.standard_normal((B, d, d)))[0]
Ud = torch.from_numpy(U).cuda()
Cd = torch.from_numpy(np.ascontiguousarray(np.asarray(ff.Basis.pauli(4)))).cuda()
out = torch.empty((B, N, N), dtype=torch.float64, device='cuda')
need = lib.ffk_liouville_workspace_bytes(B, d, N)
ws = torch.empty(need, dtype=torch.uint8, device='cuda')
p = lambda t: ctypes.c_void_p(t.data_ptr())
for _ in range(3):
    _lib.check(lib.ffk_liouville_dev(p(Ud), B, d, p(Cd), N, 1, p(out), p(ws), need, None))
    torch.cuda.synchronize()
nblk = 8192
tr = (ctypes.c_ulonglong*(6*nblk))()
lib.ffk_debug_lv_trace(tr, nblk)
tr = np.array(tr, dtype=np.uint64).reshape(nblk, 6)
if not os.environ.get('FFK_LIOUVILLE_FUSED', '').startswith('b'):
    # the persistent form: a block per batch element (or share): start, U staged, first group's matrix instructions done,
    # first barrier, conjugating wavefronts done, (the gather's end in place of the hardware id)
    tr = tr[tr[:, 4] > 0]
    t0 = tr[:, 0].min()
    st = (tr[:, :5].astype(np.int64) - int(t0))/100.0
    gend = (tr[:, 5].astype(np.int64) - int(t0))/100.0
    print('blocks', len(tr), 'span %.1f us' % max(st[:, 4].max(), gend.max()))
    for name, col in (('head (U, first operands, barrier)', st[:, 1] - st[:, 0]),
                      ('first group: matrix instructions', st[:, 2] - st[:, 1]),
                      ('first group: tile + barrier', st[:, 3] - st[:, 2]),
                      ('all groups, conjugating wavefronts', st[:, 4] - st[:, 1]),
                      ('gather behind the conjugation', gend - st[:, 4])):
        print(f'  {name:36s}: median {np.median(col):7.2f} us, 10 % {np.percentile(col, 10):7.2f}, 90 % {np.percentile(col, 90):7.2f}')
    order = np.argsort(st[:, 0])
    print('first and last blocks to start (start, head done, mfma 1 done, barrier 1, conj done, gather done):')
    for b in list(order[:4]) + list(order[-4:]):
        print('   ', ' '.join('%8.2f' % v for v in st[b]), '%8.2f' % gend[b])
    sys.exit(0)
t0 = tr[:, 0].min()
st = (tr[:, :5] - t0)/100.0          # us
hw = tr[:, 5] & 0xffffffff
xcc = (tr[:, 5] >> 32) & 0xf
cu = xcc*4096 + ((hw >> 13) & 7)*512 + ((hw >> 12) & 1)*256 + ((hw >> 8) & 0xf)*4
print('blocks', nblk, 'span %.1f us' % st[:, 4].max())
ph = np.diff(st, axis=1)
for name, col in zip(('U staged + barrier', 'matrix instructions', 'tile + list request + barrier', 'gather + stores'), ph.T):
    print(f'  {name:32s}: median {np.median(col):6.2f} us, 10 % {np.percentile(col, 10):6.2f}, 90 % {np.percentile(col, 90):6.2f}')
print('  whole block                     : median %.2f us' % np.median(st[:, 4] - st[:, 0]))
print('distinct CUs', len(np.unique(cu)))
one = np.argsort(st[:, 0])
c0 = cu[one[0]]
mine = one[cu[one] == c0][:12]
print('the first blocks of one CU (start, mfma begin, mfma end, barrier, end):')
for b in mine:
    print('   block %5d' % b, ' '.join('%7.2f' % v for v in st[b]))
mid = 0.5*st[:, 4].max()
for t in np.linspace(0, st[:, 4].max(), 11)[1:-1]:
    live = (st[:, 0] <= t) & (st[:, 4] > t)
    in_mfma = (st[:, 1] <= t) & (st[:, 2] > t)
    print(f'  t = {t:6.1f} us: {int(live.sum()):4d} blocks live, {int(in_mfma.sum()):4d} in their matrix-instruction phase')
