"""Generates filter_functions_amd/csrc/ctrl_pc_consumer.inc: the consumer wavefront's work on one
segment of the d = 4 accumulate kernel (ctrl_pc.hip) as ONE inline-asm block with an explicit
instruction order.

    python tools/gen_pc_consumer.py [--ring 5] > filter_functions_amd/csrc/ctrl_pc_consumer.inc

Why generated assembly: the step is 448 v_fma/v_mul_f64 fed by 64 + 5 LDS reads whose data has no
reuse (one ds_read_b128 per two multiply-adds), and the kernel must stay within 112 VGPRs (four of
its wavefronts per SIMD AND room for a wavefront of another pass's small kernels beside them), of
which the accumulators take 64.  hipcc either sinks every read down to its first use (read,
s_waitcnt lgkmcnt(0), two FMAs, ...: a consumer took 6-9 k cycles for 1.8 k cycles of issue) or,
when pushed with scheduling barriers, spills the accumulators (profiles/r04_b_*).  Here the folded
operands stream through a ring of RING registers-quads, each read issued RING - 1 elements ahead
of its use, and every s_waitcnt carries the exact count of younger reads that may still fly.

Mathematics per segment:  for every row m and column j
    zz   = sum_n q[m][n] W[m][n][j]                    (q real, per lane; W complex, uniform)
    z    = psi zz                                      (psi complex, per lane)
    Y[i][j] += conj(T[m][i]) z        i = 0..3         (T complex, uniform: SGPR operands)

Tile planes (64 doubles each, one per lane): 0..12 the distinct q, 13 psi.re, 14 psi.im.
Registers: v[0 : 4 RING - 1] the ring (one complex W per quad), then q of the current row (8), zz
(4), z (4), psi (4); Y[i][j] = v[Y0 + 4 (4 i + j) : +3] (re, im); T[m][i] = s[36 + 16 m + 4 i : +3].
"""
import argparse

SLOT = {}          # entry e = m*4+n -> tile plane (diagonal entries share plane 0)
_k = 0
for _e in range(16):
    if _e != 0 and _e // 4 == _e % 4:
        SLOT[_e] = 0
        continue
    SLOT[_e] = _k
    _k += 1
PSI_RE, PSI_IM = 13, 14
T0 = 36


def v(r):
    return f'v[{r}:{r + 1}]'


def s(r):
    return f's[{r}:{r + 1}]'


class Stream:
    def __init__(self):
        self.lines = []
        self.fifo = []           # tags of LDS reads in flight, oldest first

    def emit(self, text):
        self.lines.append(text)

    def lds(self, text, tag):
        self.lines.append(text)
        self.fifo.append(tag)

    def need(self, *tags):
        """wait until the reads tagged `tags` have returned (LDS returns in order)"""
        last = max((i for i, t in enumerate(self.fifo) if t in tags), default=None)
        if last is None:
            return
        younger = len(self.fifo) - 1 - last
        self.lines.append(f's_waitcnt lgkmcnt({younger})')
        self.fifo = self.fifo[last + 1:]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ring', type=int, default=5)
    args = ap.parse_args()
    R = args.ring
    Q = 4*R
    ZZ, Z, PSI = Q + 8, Q + 12, Q + 16
    NTMP = Q + 20
    Y0 = NTMP
    st = Stream()

    def read_q(m):
        sl = [SLOT[m*4 + n] for n in range(4)]
        st.lds(f'ds_read2st64_b64 v[{Q}:{Q + 3}], %[vq] offset0:{sl[0]} offset1:{sl[1]}', f'q{m}')
        st.lds(f'ds_read2st64_b64 v[{Q + 4}:{Q + 7}], %[vq] offset0:{sl[2]} offset1:{sl[3]}', f'q{m}')

    def read_w(idx):
        kk, n = divmod(idx, 4)
        m, j = divmod(kk, 4)
        base = 4*(idx % R)
        off = ((m*4 + n)*4 + j)*16
        st.lds(f'ds_read_b128 v[{base}:{base + 3}], %[vw] offset:{off}', f'w{idx}')

    st.lds(f'ds_read2st64_b64 v[{PSI}:{PSI + 3}], %[vq] offset0:{PSI_RE} offset1:{PSI_IM}', 'psi')
    read_q(0)
    for idx in range(R - 1):
        read_w(idx)
    for kk in range(16):
        m, j = divmod(kk, 4)
        for n in range(4):
            idx = kk*4 + n
            if idx + R - 1 < 64:
                read_w(idx + R - 1)
            tags = [f'w{idx}']
            if j == 0 and n == 0:
                tags.append(f'q{m}')
            st.need(*tags)
            q = v(Q + 2*n)
            wre, wim = v(4*(idx % R)), v(4*(idx % R) + 2)
            if n == 0:
                st.emit(f'v_mul_f64 {v(ZZ)}, {q}, {wre}')
                st.emit(f'v_mul_f64 {v(ZZ + 2)}, {q}, {wim}')
            else:
                st.emit(f'v_fma_f64 {v(ZZ)}, {q}, {wre}, {v(ZZ)}')
                st.emit(f'v_fma_f64 {v(ZZ + 2)}, {q}, {wim}, {v(ZZ + 2)}')
        if j == 3 and m + 1 < 4:
            read_q(m + 1)        # the row's q are dead now; the next row's arrive during the 20 ops below
        # z = psi zz
        if kk == 0:
            st.need('psi')
        st.emit(f'v_mul_f64 {v(Z)}, {v(PSI)}, {v(ZZ)}')
        st.emit(f'v_mul_f64 {v(Z + 2)}, {v(PSI)}, {v(ZZ + 2)}')
        st.emit(f'v_fma_f64 {v(Z)}, -{v(PSI + 2)}, {v(ZZ + 2)}, {v(Z)}')
        st.emit(f'v_fma_f64 {v(Z + 2)}, {v(PSI + 2)}, {v(ZZ)}, {v(Z + 2)}')
        # Y[i][j] += conj(T[m][i]) z : first terms of all eight chains, then the second terms
        for term in range(2):
            for i in range(4):
                yre = Y0 + 4*(4*i + j)
                yim = yre + 2
                tre, tim = T0 + 16*m + 4*i, T0 + 16*m + 4*i + 2
                if term == 0:
                    st.emit(f'v_fma_f64 {v(yre)}, {s(tre)}, {v(Z)}, {v(yre)}')
                    st.emit(f'v_fma_f64 {v(yim)}, {s(tre)}, {v(Z + 2)}, {v(yim)}')
                else:
                    st.emit(f'v_fma_f64 {v(yre)}, {s(tim)}, {v(Z + 2)}, {v(yre)}')
                    st.emit(f'v_fma_f64 {v(yim)}, -{s(tim)}, {v(Z)}, {v(yim)}')
    assert not st.fifo, st.fifo
    n_valu = sum(1 for ln in st.lines if ln.startswith('v_'))
    n_lds = sum(1 for ln in st.lines if ln.startswith('ds_'))
    print('// GENERATED by tools/gen_pc_consumer.py -- do not edit; see that file for the layout.')
    print(f'// ring of {R}: {n_valu} vector instructions, {n_lds} LDS reads per segment; temporaries v[0:{NTMP - 1}],')
    print(f'// accumulators v[{Y0}:{Y0 + 63}].')
    print(f'#define FFK_PC_CONSUMER_Y0 {Y0}')
    print('#define FFK_PC_CONSUMER_ASM \\')
    for ln in st.lines:
        print(f'    "{ln}\\n\\t" \\')
    print('    ""')
    print('#define FFK_PC_CONSUMER_Y_OPERANDS(Y0, Y1, Y2, Y3, Y4, Y5, Y6, Y7) \\')
    ops = ', '.join(f'"+{{v[{Y0 + 8*k}:{Y0 + 8*k + 7}]}}"(Y{k})' for k in range(8))
    print(f'    {ops}')
    print('#define FFK_PC_CONSUMER_CLOBBERS \\')
    regs = ', '.join(f'"v{r}"' for r in range(NTMP))
    print(f'    {regs}, "memory"')


if __name__ == '__main__':
    main()
