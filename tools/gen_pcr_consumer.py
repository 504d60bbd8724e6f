"""Generates filter_functions_amd/csrc/ctrl_pcr_consumer.inc: one segment of a consumer wavefront of the d = 8
matrix-core accumulate kernel (ctrl_pcr.hip) as ONE inline-asm block with an explicit instruction order, fixed
registers and exact s_waitcnt counts.

    python tools/gen_pcr_consumer.py > filter_functions_amd/csrc/ctrl_pcr_consumer.inc

(tests/test_abi.py::test_generated_consumers_are_current asserts that the committed file is what this script
prints.)

Why (profiles/r06_a_*): the compiler-scheduled consumer reads every entry of W' once per set of four frequencies
(32 KB of LDS traffic per wavefront and segment of 73 KB in all; an ablation build that reads half of those bytes
is 10 % faster), and holding an entry for TWO sets needs 32 registers of products next to 64 accumulators at 128
registers per wavefront -- hipcc spills (scratch reloads inside a 6 us step) or, with one column group at a time,
serialises read -> wait -> four matrix instructions.  With the registers assigned by hand the two-set form fits in
125, every read is requested one or two steps ahead of its use, and the operands of the second product (T, psi)
and the next pair's first operands arrive in registers that the phase before has just finished with.  Every LDS
instruction (reads, the rotation's ds_bpermute) sits in a gap behind a matrix instruction -- it issues in the
matrix instruction's shadow; a first form with the rotation as a phase of its own (70 instructions without a
matrix instruction, three wavefronts of a SIMD in step) was 5 % SLOWER than the compiler's schedule.

Mathematics per segment, operator and pair p of sets of four frequencies (ctrl_pcr.hip, lane (c4, b, q) =
(lane & 3, (lane >> 2) & 3, lane >> 4)):
    first product   P_u[n = 4 ng + b][i = 4 ig + q](w_c4)  = sum_m W'[m][n][i] q[m][n](w),  m = 4 s + q: two steps s
    psi, rotation   P_u <- psi(w_c4) P_u, then lane (c4, b, q) takes the value of lane (b, q, c4)
    second product  Y_set[i = 4 ig + q][j = 4 jg + c4](w_b) += sum_n P_u[n = 4 ng + q][i] T[n][j]
in the order of the C++ consumer it replaces (s, then ng for every accumulator: bit-identical results).

Registers:
    v[0:63]     Y[set][ig][jg] (re, im) at 2 (((set 2 + ig) 2 + jg) 2 + reim)          pinned across segments
    v[64:95]    P[u][ng][ig] (re, im) at 64 + 2 (((u 2 + ng) 2 + ig) 2 + reim)
    v[96:111]   S0..S3: staging of W' (S0, S1) and q (S2, S3) in the first product; then T of ng = 0 (S0, S1) and of
                ng = 1 (S2, S3); S3 is the rotation's temporary in between
    v[112:119]  X0, X1: the pair's first W' entry and q, then psi of the two sets
    v120..v124  LDS addresses of this lane's W', q, psi, T (of the segment's buffer; moved on at the end) and
                the byte index of the rotation's source lane
"""
NS = 2                   # 4 x 4 blocks per dimension
ACC0, P0, S0, X0, A0 = 0, 64, 96, 112, 120
S = [S0, S0 + 4, S0 + 8, S0 + 12]
X = [X0, X0 + 4]
A_W, A_Q, A_PSI, A_T, ROT = (f'v{A0 + k}' for k in range(5))
VEND = A0 + 5
Q_S, Q_NG, Q_SET = 17920, 2176, 32     # byte strides of q: row group s, column group ng, set (ctrl_pcr.hip pcr_slot)
W_STEP = 1024                          # bytes between (s, ng, ig) entries of W'
T_S, T_G = 512, 64                     # bytes between T[4 s + q][.] and T[.][4 g + c4]
PSI_SET = 64


def v(r):
    return f'v[{r}:{r + 1}]'


def v4(r):
    return f'v[{r}:{r + 3}]'


def acc(st, ig, jg, reim):
    return ACC0 + 2*(((st*2 + ig)*2 + jg)*2 + reim)


def prod(u, ng, ig, reim):
    return P0 + 2*(((u*2 + ng)*2 + ig)*2 + reim)


import os
# tuning builds (wrong results, never shipped): GEN_PCR_DROP=rot|valu|reads|waits leaves that part of the block out
DROP = os.environ.get('GEN_PCR_DROP', '').split(',')


class Stream:
    def __init__(self):
        self.lines = []
        self.fifo = []              # tags of LDS operations in flight, oldest first

    def emit(self, text):
        if 'valu' in DROP and text.startswith('v_') and 'mfma' not in text and not text.startswith('v_add_u32'):
            return
        self.lines.append(text)

    def lds(self, text, tag):
        if ('rot' in DROP and 'bpermute' in text) or ('reads' in DROP and text.startswith('ds_read')):
            return
        self.lines.append(text)
        self.fifo.append(tag)

    def need(self, *tags):
        """wait until the operations tagged `tags` are done (the LDS queue is in order)"""
        last = max((i for i, t in enumerate(self.fifo) if t in tags), default=None)
        if last is None:
            return
        younger = len(self.fifo) - 1 - last
        if 'waits' not in DROP:
            self.lines.append(f's_waitcnt lgkmcnt({min(younger, 15)})')
        if younger <= 15:
            self.fifo = self.fifo[last + 1:]
        else:
            self.fifo = self.fifo[len(self.fifo) - 15:]


def off(base, offset):
    assert 0 <= offset < 65536
    return f'{base} offset:{offset}' if offset else base


def read_w(st, pair, k, dst):
    st.lds(f'ds_read_b128 {v4(dst)}, {off(A_W, k*W_STEP)}', f'w{pair}.{k}')


def read_q(st, pair, j, dst):
    """q[4 s + q][4 ng + b] of the pair's two sets, (s, ng) = (j >> 1, j & 1): u = 0 in dst, u = 1 in dst + 2"""
    s, ng = j >> 1, j & 1
    for u in range(2):
        st.lds(f'ds_read_b64 {v(dst + 2*u)}, {off(A_Q, s*Q_S + ng*Q_NG + (2*pair + u)*Q_SET)}', f'q{pair}.{j}')


def mfma(dst, a, b, c, neg=False):
    return f'v_mfma_f64_4x4x4_4b_f64 {v(dst)}, {v(a)}, {v(b)}, {c if isinstance(c, str) else v(c)}' + (' neg:[1,0,0]' if neg else '')


def deal(st, mfmas, sides, every=1):
    """matrix instructions with side work dealt out into the gaps behind them, one item per `every` gaps: an LDS
    instruction issues in the shadow of the wavefront's own matrix instruction -- up to one ds_read per gap and one
    ds_bpermute_b32 per two gaps cost about a cycle each, denser ones 11-15 (ds_bpermute) or 4-7 (ds_read_b128) and a
    ds_write_b128 35-45 (tools/lds_issue_probe.py, profiles/r06_b_*).  A side item is a function of the stream."""
    sides = list(sides)
    for i, m in enumerate(mfmas):
        st.emit(m)
        if sides and i % every == every - 1:
            item = sides.pop(0)
            if item is not None:                # (None: a gap left empty)
                item(st)
    for f in sides:                 # (more items than gaps: the rest behind the group)
        if f is not None:
            f(st)


# FP steps in the order (ng, s, ig): the products of the column group ng = 0 are finished after four steps
def step_index(k):
    ng, s, ig = k >> 2, (k >> 1) & 1, k & 1
    return s, ng, ig


WSLOT = lambda k: X[0] if k == 0 else S[(k - 1) % 2]
QSLOT = lambda j: X[1] if j == 0 else S[2 + (j - 1) % 2]
TSLOT = {(0, 0): S[0], (0, 1): S[1], (1, 0): S[2], (1, 1): S[3]}     # T[4 ng + q][4 jg + c4] of (ng, jg)


def read_w_step(st, pair, k):
    s, ng, ig = step_index(k)
    st.lds(f'ds_read_b128 {v4(WSLOT(k))}, {off(A_W, ((s*2 + ng)*2 + ig)*W_STEP)}', f'w{pair}.{k}')


def read_q_step(st, pair, j):
    """q[4 s + q][4 ng + b] of the pair's two sets for the steps 2 j, 2 j + 1: u = 0 in the slot, u = 1 behind it"""
    s, ng, _ = step_index(2*j)
    for u in range(2):
        st.lds(f'ds_read_b64 {v(QSLOT(j) + 2*u)}, {off(A_Q, s*Q_S + ng*Q_NG + (2*pair + u)*Q_SET)}', f'q{pair}.{j}')


def burst(st, pair, ng, tmp):
    """P <- psi P for the four products of a column group, in place (the products and their order are ffk::cmul's);
    vector FP64 instructions grouped: dealt out between matrix instructions each costs ~12 cycles instead of 4.25"""
    st.emit('s_nop 7')              # (the last step's matrix instructions wrote the products read here)
    st.emit('s_nop 7')
    for u in range(2):
        st.need(f'psi{pair}.{u}')
    for ig in range(NS):
        for u in range(2):
            pr, pi = prod(u, ng, ig, 0), prod(u, ng, ig, 1)
            psr, psi = X[u], X[u] + 2
            st.emit(f'v_mul_f64 {v(tmp)}, {v(psi)}, -{v(pi)}')
            st.emit(f'v_mul_f64 {v(tmp + 2)}, {v(psi)}, {v(pr)}')
            st.emit(f'v_fma_f64 {v(pr)}, {v(psr)}, {v(pr)}, {v(tmp)}')
            st.emit(f'v_fma_f64 {v(pi)}, {v(psr)}, {v(pi)}, {v(tmp + 2)}')


def rotate_items(pair, u, ng):
    """the lane rotation of the set's two products of a column group, in place: eight side items"""
    items = []
    for ig in range(NS):
        for h in range(4):
            r = prod(u, ng, ig, 0) + h
            items.append(lambda st, r=r: st.lds(f'ds_bpermute_b32 v{r}, {ROT}, v{r}', f'r{pair}.{ng}.{u}'))
    return items


def read_t(pair, ng, jg):
    return lambda st: st.lds(f'ds_read_b128 {v4(TSLOT[ng, jg])}, {off(A_T, ng*T_S + jg*T_G)}', f't{pair}.{ng}')


def fp_step(st, pair, k, sides, every=1):
    s, ng, ig = step_index(k)
    j = k >> 1
    st.need(f'w{pair}.{k}', f'q{pair}.{j}')
    w, q = WSLOT(k), QSLOT(j)
    mfmas = []
    for u in range(2):
        for reim in range(2):
            c = prod(u, ng, ig, reim) if s else '0'
            mfmas.append(mfma(prod(u, ng, ig, reim), w + 2*reim, q + 2*u, c))
    deal(st, mfmas, sides, every)


def sp_mfmas(pair, ng, u):
    """the 16 matrix instructions of (set u, column group ng), in the C++ consumer's order"""
    st_ = 2*pair + u
    out = []
    for ig in range(NS):
        for jg in range(NS):
            pr, t = prod(u, ng, ig, 0), TSLOT[ng, jg]
            out.append(mfma(acc(st_, ig, jg, 0), pr, t, acc(st_, ig, jg, 0)))
            out.append(mfma(acc(st_, ig, jg, 1), pr, t + 2, acc(st_, ig, jg, 1)))
    for ig in range(NS):
        for jg in range(NS):
            pi, t = prod(u, ng, ig, 1), TSLOT[ng, jg]
            out.append(mfma(acc(st_, ig, jg, 0), pi, t + 2, acc(st_, ig, jg, 0), neg=True))
            out.append(mfma(acc(st_, ig, jg, 1), pi, t, acc(st_, ig, jg, 1)))
    return out


def pair_block(st, pair, first_pair, last_pair):
    """One pair of sets.  W(0), q(0) are in X0, X1 and W(1), W(2) on their way (requested by the previous pair's
    second product, or at the top of the block)."""
    if first_pair:
        read_w_step(st, pair, 1)
        read_w_step(st, pair, 2)
    read_q_step(st, pair, 1)
    # ---- first product, column group 0 (steps 0..3); requests into the registers a step has finished with
    for k in range(4):
        sides = []
        if k >= 2:
            sides.append(lambda st, k=k: read_w_step(st, pair, k + 1))
        if k == 2:
            sides.append(lambda st: read_q_step(st, pair, 2))
            for u in range(2):
                sides.append(lambda st, u=u: st.lds(f'ds_read_b128 {v4(X[u])}, {off(A_PSI, (2*pair + u)*PSI_SET)}',
                                                    f'psi{pair}.{u}'))
        fp_step(st, pair, k, sides)
    burst(st, pair, 0, prod(0, 1, 0, 0))          # (temporaries: the second column group's registers, not yet written)
    # ---- first product, column group 1 (steps 4..7), carrying the rotation of set 0's first column group
    rot = rotate_items(pair, 0, 0)
    for k in range(4, 8):
        sides = []
        if k + 1 < 8:
            sides.append(lambda st, k=k: read_w_step(st, pair, k + 1))
        if k == 4:
            sides.append(lambda st: read_q_step(st, pair, 3))
        sides += [rot.pop(0), rot.pop(0)]
        fp_step(st, pair, k, sides)
    for jg in range(NS):
        read_t(pair, 0, jg)(st)                   # (arrive during the vector products below)
    burst(st, pair, 1, S[2])
    # ---- second product, column group 0
    st.need(f'r{pair}.0.0', f't{pair}.0')
    deal(st, sp_mfmas(pair, 0, 0), rotate_items(pair, 1, 0), every=2)
    st.need(f'r{pair}.0.1')
    rot = rotate_items(pair, 0, 1)
    sides = []
    for i in range(8):              # a rotation every second gap, the second column group's T in between
        sides += [read_t(pair, 1, i) if i < 2 else None, rot[i]]
    deal(st, sp_mfmas(pair, 0, 1), sides)
    # ---- second product, column group 1
    st.need(f'r{pair}.1.0', f't{pair}.1')
    rot = rotate_items(pair, 1, 1)
    nxt = [None]*8 if last_pair else [lambda st: read_w(st, pair + 1, 0, X[0]), lambda st: read_q(st, pair + 1, 0, X[1])] + [None]*6
    sides = []
    for i in range(8):
        sides += [nxt[i], rot[i]]
    deal(st, sp_mfmas(pair, 1, 0), sides)
    st.need(f'r{pair}.1.1')
    sides = []
    if not last_pair:
        sides = [lambda st: read_w_step(st, pair + 1, 1), lambda st: read_w_step(st, pair + 1, 2)]
    deal(st, sp_mfmas(pair, 1, 1), sides, every=4)


def build():
    st = Stream()
    read_w(st, 0, 0, X[0])
    read_q(st, 0, 0, X[1])
    for pair in range(2):
        pair_block(st, pair, pair == 0, pair == 1)
    assert not st.fifo, st.fifo
    # on to the other buffer (the segment after this one)
    for a in (A_W, A_Q, A_PSI, A_T):
        st.emit(f'v_add_u32_e32 {a}, %[delta], {a}')
    return st


def main():
    st = build()
    n_mfma = sum('mfma' in ln for ln in st.lines)
    n_valu = sum(ln.startswith('v_') and 'mfma' not in ln for ln in st.lines)
    n_lds = sum(ln.startswith('ds_') for ln in st.lines)
    print('// GENERATED by tools/gen_pcr_consumer.py -- do not edit; see that file for the register map.')
    print('// Included by ctrl_pcr.hip inside namespace ffk::{anonymous}.')
    print(f'// one segment: {n_mfma} matrix, {n_valu} vector instructions, {n_lds} LDS operations; {VEND} vector registers')
    print('struct PcrConsumer {')
    print(f'    static constexpr int kVgprs = {VEND};')
    print('    // Y0..Y3: the accumulators of the wavefront\'s four sets, Y[(ig 2 + jg) 2 + reim]; a_*: LDS addresses of this')
    print('    // lane\'s W\', q, psi, T in the segment\'s buffer (moved by `delta` bytes at the end: the next segment\'s buffer);')
    print('    // rot: byte index of the lane whose product this lane takes')
    print('    static __device__ __forceinline__ void segment(double8_t& Y0, double8_t& Y1, double8_t& Y2, double8_t& Y3,')
    print('                                                   unsigned& a_w, unsigned& a_q, unsigned& a_psi, unsigned& a_t,')
    print('                                                   unsigned rot, int delta) {')
    print('        asm volatile(')
    for ln in st.lines:
        print(f'            "{ln}\\n\\t"')
    print('            : "+{v[0:15]}"(Y0), "+{v[16:31]}"(Y1), "+{v[32:47]}"(Y2), "+{v[48:63]}"(Y3),')
    print(f'              "+{{{A_W}}}"(a_w), "+{{{A_Q}}}"(a_q), "+{{{A_PSI}}}"(a_psi), "+{{{A_T}}}"(a_t)')
    print(f'            : "{{{ROT}}}"(rot), [delta] "s"(delta)')
    print('            : ' + ', '.join(f'"v{r}"' for r in range(P0, A0)) + ', "memory");')
    print('    }')
    print('};')


if __name__ == '__main__':
    main()
