"""Generates filter_functions_amd/csrc/ctrl_pq_consumer.inc: a consumer wavefront's work on one tile
(segment) of the d = 4 matrix-core accumulate kernel (ctrl_pq.hip, three operators per block) as ONE
inline-asm block with an explicit instruction order and exact s_waitcnt counts.

    python tools/gen_pq_consumer.py > filter_functions_amd/csrc/ctrl_pq_consumer.inc

Why generated assembly (profiles/r05_b_*): the tile is 64 vector + 18 matrix instructions fed by 22 LDS
reads, two consumers per SIMD.  To keep the SIMD busy the NEXT tile's operands must be requested from
inside the current tile -- after the last vector instruction that reads the operand registers, before the
last set's matrix instructions -- and the flag of the tile after that must be read a tile ahead.  hipcc
moves the vector work of the last set behind those requests (two live copies of the operands: 168 VGPRs,
accumulators spilled inside the loop), sinks the reads to their first use, or reorders the interleaved
chains back into operator-by-operator order; with __builtin_amdgcn_sched_barrier it keeps the order only
inside a basic block.  Here every LDS operation of the consumer loop is in the block, so the in-order LDS
queue is known exactly and every wait names the number of younger operations that may still fly.

Mathematics per tile, for the set s (four frequencies, one per 4x4x4 block of the matrix instruction)
and the operator a (ctrl_pq.hip):
    c    = psi conj(T[m][i])                               (per lane (i, block, m): the A operand)
    z_a  = sum_n q[m][n] W_a[m][n][j]                      (per lane (m, block, j): the B operand)
    P1_a += cr^T zr_a,  P2_a += ci^T zi_a,  P3_a += (cr + ci)^T (zr_a + zi_a)

Registers (fixed; `{v[a:b]}` constraints on the C++ side):
    v[0:23]     temporaries: zr_a v[4a], zi_a v[4a+2], zs_a v[12+2a], cr v18, ci v20, cs v22
    v[24:71]    W_a[n] = (re, im) at v[24 + 4 (4 a + n)]
    v[72:87]    q01 v72 (q[m][0], q[m][1]), q23 v76, psi v80 (re, im), T v84 (tr, ti)
    v[88:123]   accumulators P_k of (operator a, set s) at v[88 + 2 (3 (2 a + s) + k)]
LDS operations, in queue order: flag of tile it + 2, partner's progress | set 1's q01, q23, psi |
next tile's T, psi, q01, W[.][0], W[.][1], q23, W[.][2], W[.][3] | done counter, own progress.
"""
import os

NC = 3
# tuning builds (never shipped): GEN_PQ_DROP=mfma|valu|next|set1 leaves that part of the block out
DROP = os.environ.get("GEN_PQ_DROP", "")
# sets of four frequencies a consumer owns (2: eight consumers per block, two per SIMD -- the shipped kernel;
# 4: four consumers, one per SIMD, 72 accumulator registers -- ctrl_pq.hip -DFFK_PQ_SETS=4)
NSETS = int(os.environ.get("GEN_PQ_SETS", "2"))
ZR = [4*a for a in range(NC)]
ZI = [4*a + 2 for a in range(NC)]
ZS = [12 + 2*a for a in range(NC)]
CR, CI, CS = 18, 20, 22
NTMP = 24
W0 = 24
Q01, Q23, PSI, T = 72, 76, 80, 84
ACC0 = 88
W_BYTES = 1024          # per operator: [n][m][j] complex
T_OFF = NC*W_BYTES      # (tr, ti) pairs behind the operators' W, same lane index


def v(r):
    return f'v[{r}:{r + 1}]'


def v4(r):
    return f'v[{r}:{r + 3}]'


def wre(a, n):
    return W0 + 4*(4*a + n)


def acc(a, s, k):
    return ACC0 + 2*(3*(NSETS*a + s) + k)


# the LDS queue at block entry: what the previous block left in flight (after the C++ prologue or a spin on
# a flag everything is complete and the waits are satisfied at once)
ENTRY = ['T', 'psi', 'q01'] + ['w0']*NC + ['w1']*NC + ['q23'] + ['w2']*NC + ['w3']*NC + ['done', 'prog']


class Stream:
    def __init__(self, fifo):
        self.lines = []
        self.fifo = list(fifo)      # tags of LDS operations in flight, oldest first

    def emit(self, text):
        if DROP == 'mfma' and 'mfma' in text:
            return
        if DROP == 'valu' and text.startswith('v_') and 'mfma' not in text:
            return
        self.lines.append(text)

    def lds(self, text, tag):
        self.lines.append(text)
        self.fifo.append(tag)

    def need(self, *tags):
        """wait until the operations tagged `tags` are done (the LDS queue is in order)"""
        last = max((i for i, t in enumerate(self.fifo) if t in tags), default=None)
        if last is None:
            return
        younger = len(self.fifo) - 1 - last
        self.lines.append(f's_waitcnt lgkmcnt({min(younger, 15)})')
        if younger <= 15:
            self.fifo = self.fifo[last + 1:]
        else:                       # the wait covered more than asked for
            self.fifo = self.fifo[len(self.fifo) - 15:]


def next_tile_requests(st):
    st.lds(f'ds_read_b128 {v4(T)}, %[a_w] offset:{T_OFF}', 'T')
    st.lds(f'ds_read_b128 {v4(PSI)}, %[a_p]', 'psi')
    st.lds(f'ds_read_b128 {v4(Q01)}, %[a_q0]', 'q01')
    for n in (0, 1):
        for a in range(NC):
            st.lds(f'ds_read_b128 {v4(wre(a, n))}, %[a_w] offset:{a*W_BYTES + n*256}', f'w{n}')
    st.lds(f'ds_read_b128 {v4(Q23)}, %[a_q0] offset:4096', 'q23')
    for n in (2, 3):
        for a in range(NC):
            st.lds(f'ds_read_b128 {v4(wre(a, n))}, %[a_w] offset:{a*W_BYTES + n*256}', f'w{n}')


def vector_part(st, s):
    q = [Q01, Q01 + 2, Q23, Q23 + 2]
    pr, pi = PSI, PSI + 2
    tr, ti = T, T + 2
    # stage 1
    st.need('T', 'psi', 'q01', 'w0')
    st.emit(f'v_mul_f64 {v(CR)}, {v(pi)}, {v(ti)}')
    st.emit(f'v_mul_f64 {v(CI)}, {v(pi)}, {v(tr)}')
    for a in range(NC):
        st.emit(f'v_mul_f64 {v(ZR[a])}, {v(q[0])}, {v(wre(a, 0))}')
        st.emit(f'v_mul_f64 {v(ZI[a])}, {v(q[0])}, {v(wre(a, 0) + 2)}')
    # stage 2
    st.need('w1')
    st.emit(f'v_fma_f64 {v(CR)}, {v(pr)}, {v(tr)}, {v(CR)}')
    st.emit(f'v_fma_f64 {v(CI)}, -{v(pr)}, {v(ti)}, {v(CI)}')
    for a in range(NC):
        st.emit(f'v_fma_f64 {v(ZR[a])}, {v(q[1])}, {v(wre(a, 1))}, {v(ZR[a])}')
        st.emit(f'v_fma_f64 {v(ZI[a])}, {v(q[1])}, {v(wre(a, 1) + 2)}, {v(ZI[a])}')
    # stage 3
    st.need('q23', 'w2')
    st.emit(f'v_add_f64 {v(CS)}, {v(CR)}, {v(CI)}')
    for a in range(NC):
        st.emit(f'v_fma_f64 {v(ZR[a])}, {v(q[2])}, {v(wre(a, 2))}, {v(ZR[a])}')
        st.emit(f'v_fma_f64 {v(ZI[a])}, {v(q[2])}, {v(wre(a, 2) + 2)}, {v(ZI[a])}')
    # stage 4
    st.need('w3')
    for a in range(NC):
        st.emit(f'v_fma_f64 {v(ZR[a])}, {v(q[3])}, {v(wre(a, 3))}, {v(ZR[a])}')
        st.emit(f'v_fma_f64 {v(ZI[a])}, {v(q[3])}, {v(wre(a, 3) + 2)}, {v(ZI[a])}')


def matrix_part(st, s):
    for a in range(NC):
        st.emit(f'v_add_f64 {v(ZS[a])}, {v(ZR[a])}, {v(ZI[a])}')
    for a in range(NC):
        st.emit(f'v_mfma_f64_4x4x4_4b_f64 {v(acc(a, s, 0))}, {v(CR)}, {v(ZR[a])}, {v(acc(a, s, 0))}')
        st.emit(f'v_mfma_f64_4x4x4_4b_f64 {v(acc(a, s, 1))}, {v(CI)}, {v(ZI[a])}, {v(acc(a, s, 1))}')
        st.emit(f'v_mfma_f64_4x4x4_4b_f64 {v(acc(a, s, 2))}, {v(CS)}, {v(ZS[a])}, {v(acc(a, s, 2))}')


def build(last):
    """one tile; `last`: no next tile to request (the last tile of the block)"""
    # what the previous block (or the C++ prologue, all of it complete) left in the queue
    st = Stream(ENTRY)
    st.lds('ds_read_b32 %[flag], %[a_flag]', 'flag')
    st.lds('ds_read_b32 %[partner], %[a_partner]', 'partner')
    for s_ in range(NSETS - 1):
        vector_part(st, s_)
        # the next set's q, psi into the registers this set is done with; they arrive during its nine matrix
        # instructions
        st.lds(f'ds_read_b128 {v4(PSI)}, %[a_p1] offset:{64*s_}', 'psi')
        st.lds(f'ds_read_b128 {v4(Q01)}, %[a_q{s_ + 1}]', 'q01')
        st.lds(f'ds_read_b128 {v4(Q23)}, %[a_q{s_ + 1}] offset:4096', 'q23')
        matrix_part(st, s_)
        # T and W are still this tile's: only q01, q23, psi have to arrive
        st.fifo = [t if t in ('q01', 'q23', 'psi', 'flag', 'partner') else 'old' for t in st.fifo]
    vector_part(st, NSETS - 1)
    # every operand register is dead: the next tile's operands fly during the matrix instructions and the hand-over
    assert not st.fifo, st.fifo
    if not last and DROP != 'next':
        next_tile_requests(st)
    elif DROP == 'next':
        st.fifo = list(ENTRY[:-2])
    matrix_part(st, NSETS - 1)
    # hand the slot back: lane 0 counts this consumer in and publishes its progress (LDS operations of a
    # wavefront execute in order: both are behind the tile's reads without a wait)
    st.emit('s_mov_b64 exec, 1')
    st.lds('ds_add_u32 %[a_done], %[one]', 'done')
    st.lds('ds_write_b32 %[a_prog], %[progress]', 'prog')
    st.emit('s_mov_b64 exec, -1')
    if last:
        st.emit('s_waitcnt lgkmcnt(0)')
    else:
        assert st.fifo == ENTRY, st.fifo
    return st


def dump(name, st):
    n_valu = sum(1 for ln in st.lines if ln.startswith('v_') and 'mfma' not in ln)
    n_mfma = sum(1 for ln in st.lines if 'mfma' in ln)
    n_lds = sum(1 for ln in st.lines if ln.startswith('ds_'))
    print(f'// {name}: {n_valu} vector, {n_mfma} matrix instructions, {n_lds} LDS operations')
    print(f'#define {name} \\')
    for ln in st.lines:
        print(f'    "{ln}\\n\\t" \\')
    print('    ""')


def prologue():
    """the first tile's operands (the compiler then has no LDS read of its own pending on the fixed registers,
    and does not put a full wait in front of the loop's block)"""
    st = Stream([])
    next_tile_requests(st)
    st.emit('s_waitcnt lgkmcnt(0)')
    return st


def main():
    print('// GENERATED by tools/gen_pq_consumer.py -- do not edit; see that file for the register map.')
    dump('FFK_PQ_CONSUMER_ASM', build(False))
    dump('FFK_PQ_CONSUMER_PROLOGUE_ASM', prologue())
    print('#define FFK_PQ_CONSUMER_CLOBBERS \\')
    print('    ' + ', '.join(f'"v{r}"' for r in range(NTMP)) + ', "memory"')


if __name__ == '__main__':
    main()
