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
# trace build (ctrl_pq.hip -DFFK_PQ_CLOCK, tools/trace_pq.py): s_memtime into the SGPR pairs %[t0] .. %[t7] at eight
# points of the block; the stamps ride in the same in-order count as the LDS operations
STAMPS = os.environ.get("GEN_PQ_STAMPS", "") == "1"
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
ENTRY = ['w0']*NC + ['T', 'psi', 'q01'] + ['w1']*NC + ['w2']*NC + ['q23'] + ['w3']*NC + ['done', 'prog']


# logical operands of the block: asm operands in the per-tile form, fixed registers in the whole-loop form
TILE_OPERANDS = {k: f'%[{k}]' for k in ('a_w', 'a_p', 'a_q0', 'a_q1', 'a_q2', 'a_q3', 'a_p1', 'a_flag', 'a_partner', 'a_done', 'a_prog',
                                        'progress', 'one', 'flag', 'partner')}
LOOP_V = dict(a_w='v124', a_q0='v125', a_p='v126', a_q1='v127', a_p1='v128', a_flag='v129', a_done='v130',
              progress='v131', a_partner='v132', a_prog='v133', one='v134', flag='v135', partner='v136',
              b_w='v138', b_q0='v139', b_q1='v140', b_p='v141', spin='v137')
LOOP_S = dict(nit='s36', limit='s37', flags='s38', me='s39', it='s40', cur='s41', nxt='s42', t='s43', fnext='s44',
              p='s45', prio='s46', want='s47', fault='s48', spin='s49', t2='s50', has='s51')
# a consumer raises its priority while its SIMD partner has finished more than `it + LOCKSTEP` tiles (it = the tile this
# consumer works on; the partner's count is read at the top of the tile).  0: the one behind by a tile or more --
# 2 (the first shipped value): step +1.2 %, 5: +4 % (profiles/r05_v_*)
LOCKSTEP = int(os.environ.get('GEN_PQ_LOCKSTEP', '0'))
MERGE_WAITS = os.environ.get('GEN_PQ_MERGE_WAITS', '') == '1'      # one wait per set instead of one per stage
PRIO_EVERY = int(os.environ.get('GEN_PQ_PRIO_EVERY', '1'))          # partner priority every n-th tile (unrolled form)
PRIO_LOW, PRIO_HIGH = (int(x) for x in os.environ.get('GEN_PQ_PRIOS', '0,1').split(','))   # consumer priorities: normal, behind its partner
PRIO_EARLY = os.environ.get('GEN_PQ_PRIO_EARLY', '') == '1'        # tuning: the partner rule decided behind set 0's matrix instructions (half a tile fresher)
PHASE_PRIO = os.environ.get('GEN_PQ_PHASE_PRIO', '')    # tuning: 'v,m' = static priorities of the vector / matrix part of a set instead of the partner rule
ROLLED = os.environ.get('GEN_PQ_ROLLED', '') == '1'                # the whole-loop block with run-time slot arithmetic (A/B)
LOOP_CLOCK = os.environ.get('GEN_PQ_LOOP_CLOCK', '') == '1'     # s_memtime in s[52:53] / s[54:55] around the whole-loop block
TILE_BYTES = (1152 + NC*128 + 32)*8          # ctrl_pq.hip: pq_tile_doubles(NC) * 8 (checked there)
OPS = dict(TILE_OPERANDS)


OFFS = {}                                     # unrolled loop form: the slot's byte offset rides in the instruction


def O(name):
    return OPS[name]


def A(name, offset=0):
    """address operand of an LDS instruction: register and immediate offset"""
    off = offset + OFFS.get(name, 0)
    assert 0 <= off < 65536
    return f'{OPS[name]} offset:{off}' if off else OPS[name]


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

    def stamp(self, k):
        if STAMPS:
            self.lds(f's_memtime %[t{k}]', 'stamp')

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


def next_tile_requests(st, stage):
    """The next tile's operands, each requested right behind the LAST use of the registers it lands in (the last
    set's vector stages): W[.][n] is dead after stage n + 1, T, psi and q01 after stage 2, q23 after stage 4.
    (All 16 in one burst behind stage 4 cost 3.7 us of the consumers' 54.7: profiles/r05_b_*.)"""
    if DROP == 'next':
        return
    def w(n):
        for a in range(NC):
            st.lds(f'ds_read_b128 {v4(wre(a, n))}, {A("a_w", a*W_BYTES + n*256)}', f'w{n}')
    if stage == 1:
        w(0)
    elif stage == 2:
        st.lds(f'ds_read_b128 {v4(T)}, {A("a_w", T_OFF)}', 'T')
        st.lds(f'ds_read_b128 {v4(PSI)}, {A("a_p")}', 'psi')
        st.lds(f'ds_read_b128 {v4(Q01)}, {A("a_q0")}', 'q01')
        w(1)
    elif stage == 3:
        w(2)
    elif stage == 4:
        st.lds(f'ds_read_b128 {v4(Q23)}, {A("a_q0", 4096)}', 'q23')
        w(3)


def vector_part(st, s, after_stage=None):
    q = [Q01, Q01 + 2, Q23, Q23 + 2]
    pr, pi = PSI, PSI + 2
    tr, ti = T, T + 2
    # stage 1
    if MERGE_WAITS:
        # everything the set reads (requested most of a tile ago): the later waits then find nothing to wait for
        st.need('T', 'psi', 'q01', 'q23', 'w0', 'w1', 'w2', 'w3')
    else:
        st.need('T', 'psi', 'q01', 'w0')
    if NSETS == 2:
        st.stamp(1 if s == 0 else 4)
    st.emit(f'v_mul_f64 {v(CR)}, {v(pi)}, {v(ti)}')
    st.emit(f'v_mul_f64 {v(CI)}, {v(pi)}, {v(tr)}')
    for a in range(NC):
        st.emit(f'v_mul_f64 {v(ZR[a])}, {v(q[0])}, {v(wre(a, 0))}')
        st.emit(f'v_mul_f64 {v(ZI[a])}, {v(q[0])}, {v(wre(a, 0) + 2)}')
    if after_stage:
        after_stage(1)
    # stage 2
    st.need('w1')
    st.emit(f'v_fma_f64 {v(CR)}, {v(pr)}, {v(tr)}, {v(CR)}')
    st.emit(f'v_fma_f64 {v(CI)}, -{v(pr)}, {v(ti)}, {v(CI)}')
    for a in range(NC):
        st.emit(f'v_fma_f64 {v(ZR[a])}, {v(q[1])}, {v(wre(a, 1))}, {v(ZR[a])}')
        st.emit(f'v_fma_f64 {v(ZI[a])}, {v(q[1])}, {v(wre(a, 1) + 2)}, {v(ZI[a])}')
    if after_stage:
        after_stage(2)
    # stage 3
    st.need('q23', 'w2')
    st.emit(f'v_add_f64 {v(CS)}, {v(CR)}, {v(CI)}')
    for a in range(NC):
        st.emit(f'v_fma_f64 {v(ZR[a])}, {v(q[2])}, {v(wre(a, 2))}, {v(ZR[a])}')
        st.emit(f'v_fma_f64 {v(ZI[a])}, {v(q[2])}, {v(wre(a, 2) + 2)}, {v(ZI[a])}')
    if after_stage:
        after_stage(3)
    # stage 4
    st.need('w3')
    for a in range(NC):
        st.emit(f'v_fma_f64 {v(ZR[a])}, {v(q[3])}, {v(wre(a, 3))}, {v(ZR[a])}')
        st.emit(f'v_fma_f64 {v(ZI[a])}, {v(q[3])}, {v(wre(a, 3) + 2)}, {v(ZI[a])}')
    if after_stage:
        after_stage(4)


def matrix_part(st, s, glue=()):
    """the nine matrix instructions of a set; `glue`: scalar / address instructions of the loop form, dealt out
    behind the matrix instructions (they issue while the matrix pipe works)"""
    if NSETS == 2:
        st.stamp(2 if s == 0 else 5)
    for a in range(NC):
        st.emit(f'v_add_f64 {v(ZS[a])}, {v(ZR[a])}, {v(ZI[a])}')
    glue = list(glue)
    per = -(-len(glue)//(3*NC)) if glue else 0
    for a in range(NC):
        for k, (src_a, src_b) in enumerate(((CR, ZR[a]), (CI, ZI[a]), (CS, ZS[a]))):
            st.emit(f'v_mfma_f64_4x4x4_4b_f64 {v(acc(a, s, k))}, {v(src_a)}, {v(src_b)}, {v(acc(a, s, k))}')
            for _ in range(per):
                if glue:
                    item = glue.pop(0)          # a tuple: instructions with branches among them, kept together
                    for line in ((item,) if isinstance(item, str) else item):
                        st.emit(line)
    assert not glue
    if NSETS == 2:
        st.stamp(3 if s == 0 else 6)


def build(last):
    """one tile; `last`: no next tile to request (the last tile of the block)"""
    # what the previous block (or the C++ prologue, all of it complete) left in the queue
    st = Stream(ENTRY)
    st.stamp(0)
    st.lds(f'ds_read_b32 {O("flag")}, {O("a_flag")}', 'flag')
    st.lds(f'ds_read_b32 {O("partner")}, {O("a_partner")}', 'partner')
    for s_ in range(NSETS - 1):
        # the next set's psi, q01 behind stage 2 (their registers' last use), q23 behind stage 4: they arrive during
        # the rest of the set and its nine matrix instructions
        def inner(stage, s_=s_):
            if stage == 2:
                st.lds(f'ds_read_b128 {v4(PSI)}, {O("a_p1")} offset:{64*s_}', 'psi')
                st.lds(f'ds_read_b128 {v4(Q01)}, {O("a_q" + str(s_ + 1))}', 'q01')
            elif stage == 4:
                st.lds(f'ds_read_b128 {v4(Q23)}, {O("a_q" + str(s_ + 1))} offset:4096', 'q23')
        vector_part(st, s_, inner)
        matrix_part(st, s_)
        # T and W are still this tile's: only q01, q23, psi have to arrive
        st.fifo = [t if t in ('q01', 'q23', 'psi', 'flag', 'partner', 'stamp') else 'old' for t in st.fifo]
    vector_part(st, NSETS - 1, None if last else (lambda stage: next_tile_requests(st, stage)))
    if DROP == 'next':
        st.fifo = list(ENTRY[:-2])
    matrix_part(st, NSETS - 1)
    # hand the slot back: lane 0 counts this consumer in and publishes its progress (LDS operations of a
    # wavefront execute in order: both are behind the tile's reads without a wait)
    st.emit('s_mov_b64 exec, 1')
    st.lds(f'ds_add_u32 {O("a_done")}, {O("one")}', 'done')
    st.lds(f'ds_write_b32 {O("a_prog")}, {O("progress")}', 'prog')
    st.emit('s_mov_b64 exec, -1')
    st.stamp(7)
    if last:
        st.emit('s_waitcnt lgkmcnt(0)')
    elif STAMPS:
        st.emit('s_waitcnt lgkmcnt(0)')     # (trace build: the stamps must have landed; the next block finds nothing in flight)
    else:
        assert st.fifo == ENTRY, st.fifo
    return st


def build_loop():
    """The whole tile loop of a consumer as one block: no compiler-generated code between the tiles.  (With the tile
    as the unit, ~35 scalar / address instructions sat between two blocks; beside a partner issuing 16-cycle matrix
    instructions each of them waits its turn: ~580 of a tile's ~2000 cycles, profiles/r05_b_trace_6_*.)  The address
    arithmetic of the next tile rides behind the matrix instructions; what is left between two tiles is the loop
    branch.  Inputs s[36:39] = (tiles, spin limit, LDS address of the flags, consumer number), v[138:141] = LDS
    addresses of this lane's W/T, q (set 0), q (set 1), psi in slot 0; the first tile's operands are in their
    registers (prologue block).  Output s48: 0, or the fault code of a flag wait that ran out."""
    assert NSETS == 2 and not STAMPS
    global OPS
    OPS = dict(LOOP_V)
    V, S = LOOP_V, LOOP_S
    st = Stream([])
    e = st.emit
    # ---- preamble ----
    for r in ('it', 'cur', 'prio', 'fault', 'fnext'):
        e(f's_mov_b32 {S[r]}, 0')
    e(f'v_mov_b32 {V["flag"]}, 0')
    e(f'v_mov_b32 {V["partner"]}, 0')
    e(f'v_mov_b32 {V["one"]}, 1')
    e(f's_xor_b32 {S["t"]}, {S["me"]}, 4')
    e(f's_lshl_b32 {S["t"]}, {S["t"]}, 2')
    e(f's_add_i32 {S["t"]}, {S["t"]}, {S["flags"]}')
    e(f's_add_i32 {S["t"]}, {S["t"]}, 64')
    e(f'v_mov_b32 {V["a_partner"]}, {S["t"]}')
    e(f's_lshl_b32 {S["t"]}, {S["me"]}, 2')
    e(f's_add_i32 {S["t"]}, {S["t"]}, {S["flags"]}')
    e(f's_add_i32 {S["t"]}, {S["t"]}, 64')
    e(f'v_mov_b32 {V["a_prog"]}, {S["t"]}')
    e(f'v_mov_b32 {V["a_q1"]}, {V["b_q1"]}')
    e(f'v_mov_b32 {V["a_p1"]}, {V["b_p"]}')
    e(f's_add_i32 {S["t"]}, {S["flags"]}, 8')
    e(f'v_mov_b32 {V["a_flag"]}, {S["t"]}')
    if LOOP_CLOCK:
        e('s_memtime s[52:53]')
        e('s_waitcnt lgkmcnt(0)')
    e('L_tile%=:')
    st.fifo = list(ENTRY)
    st.lds(f'ds_read_b32 {V["flag"]}, {V["a_flag"]}', 'flag')
    st.lds(f'ds_read_b32 {V["partner"]}, {V["a_partner"]}', 'partner')

    def inner(stage):
        if stage == 2:
            st.lds(f'ds_read_b128 {v4(PSI)}, {V["a_p1"]} offset:64', 'psi')
            st.lds(f'ds_read_b128 {v4(Q01)}, {V["a_q1"]}', 'q01')
        elif stage == 4:
            st.lds(f'ds_read_b128 {v4(Q23)}, {V["a_q1"]} offset:4096', 'q23')
    vector_part(st, 0, inner)
    glue0 = [
        f's_add_i32 {S["t2"]}, {S["it"]}, 1',
        f's_cmp_lt_i32 {S["t2"]}, {S["nit"]}',
        f's_cselect_b32 {S["has"]}, 1, 0',
        f's_and_b32 {S["t"]}, {S["t2"]}, 7',
        f's_mul_i32 {S["t"]}, {S["t"]}, {TILE_BYTES}',
        f's_cmp_eq_u32 {S["has"]}, 1',
        f's_cselect_b32 {S["nxt"]}, {S["t"]}, {S["cur"]}',       # the last tile requests its own slot once more
        f'v_add_u32_e32 {V["a_w"]}, {S["nxt"]}, {V["b_w"]}',
        f'v_add_u32_e32 {V["a_q0"]}, {S["nxt"]}, {V["b_q0"]}',
        f'v_add_u32_e32 {V["a_p"]}, {S["nxt"]}, {V["b_p"]}',
        # tile it + 1 published?  (its flag was read a tile ago; the slow path spins, bounded)
        f's_add_i32 {S["t"]}, {S["it"]}, 2',
        (f's_cmp_ge_i32 {S["fnext"]}, {S["t"]}',
         's_cbranch_scc1 L_ready%=',
         f's_cmp_eq_u32 {S["has"]}, 0',
         's_cbranch_scc1 L_ready%=',
         's_branch L_spin%=',
         'L_ready%=:'),
    ]
    matrix_part(st, 0, glue0)
    st.fifo = [t if t in ('q01', 'q23', 'psi', 'flag', 'partner') else 'old' for t in st.fifo]
    vector_part(st, 1, lambda stage: next_tile_requests(st, stage))
    glue1 = [
        f'v_readfirstlane_b32 {S["fnext"]}, {V["flag"]}',        # flag of tile it + 2: the next tile's "next"
        f'v_readfirstlane_b32 {S["p"]}, {V["partner"]}',
        # whoever is behind its SIMD partner raises its priority (the partner's count was read at the top of this
        # tile: the same distance as the per-tile form's, which compared a tile-old sample with it + 1)
        f's_add_i32 {S["t"]}, {S["it"]}, {LOCKSTEP}',
        f's_cmp_gt_i32 {S["p"]}, {S["t"]}',
        f's_cselect_b32 {S["want"]}, 1, 0',
        (f's_cmp_eq_u32 {S["want"]}, {S["prio"]}',
         's_cbranch_scc1 L_prio_done%=',
         f's_mov_b32 {S["prio"]}, {S["want"]}',
         f's_cmp_eq_u32 {S["want"]}, 1',
         's_cbranch_scc1 L_prio_hi%=',
         's_setprio 0',
         's_branch L_prio_done%=',
         'L_prio_hi%=:',
         's_setprio 1',
         'L_prio_done%=:'),
        # the hand-over's operands
        f's_and_b32 {S["t"]}, {S["it"]}, 7',
        f's_lshl_b32 {S["t"]}, {S["t"]}, 2',
        f's_add_i32 {S["t"]}, {S["t"]}, {S["flags"]}',
        f's_add_i32 {S["t"]}, {S["t"]}, 32',
        f'v_mov_b32 {V["a_done"]}, {S["t"]}',
        f's_add_i32 {S["t"]}, {S["it"]}, 1',
        f'v_mov_b32 {V["progress"]}, {S["t"]}',
        # the next iteration: flag of tile it + 3, this iteration's "next" becomes "current"
        f's_add_i32 {S["t"]}, {S["it"]}, 3',
        f's_and_b32 {S["t"]}, {S["t"]}, 7',
        f's_lshl_b32 {S["t"]}, {S["t"]}, 2',
        f's_add_i32 {S["t"]}, {S["t"]}, {S["flags"]}',
        f'v_mov_b32 {V["a_flag"]}, {S["t"]}',
        f'v_add_u32_e32 {V["a_q1"]}, {S["nxt"]}, {V["b_q1"]}',
        f'v_mov_b32 {V["a_p1"]}, {V["a_p"]}',
        f's_mov_b32 {S["cur"]}, {S["nxt"]}',
    ]
    matrix_part(st, 1, glue1)
    e('s_mov_b64 exec, 1')
    st.lds(f'ds_add_u32 {V["a_done"]}, {V["one"]}', 'done')
    st.lds(f'ds_write_b32 {V["a_prog"]}, {V["progress"]}', 'prog')
    e('s_mov_b64 exec, -1')
    assert st.fifo == ENTRY, st.fifo
    e(f's_add_i32 {S["it"]}, {S["it"]}, 1')
    e(f's_cmp_lt_i32 {S["it"]}, {S["nit"]}')
    e('s_cbranch_scc1 L_tile%=')
    e('s_waitcnt lgkmcnt(0)')
    e('s_setprio 0')
    if LOOP_CLOCK:
        e('s_memtime s[54:55]')
        e('s_waitcnt lgkmcnt(0)')
    e('s_nop 15')
    e('s_nop 15')
    e('s_branch L_end%=')
    # ---- slow path: tile it + 1 is not published yet ----
    e('L_spin%=:')
    e(f's_cmp_eq_u32 {S["limit"]}, 0')
    e('s_cbranch_scc1 L_ready%=')                                # a wait ran out earlier: no more waiting
    e(f's_and_b32 {S["t"]}, {S["t2"]}, 7')
    e(f's_lshl_b32 {S["t"]}, {S["t"]}, 2')
    e(f's_add_i32 {S["t"]}, {S["t"]}, {S["flags"]}')
    e(f'v_mov_b32 {V["spin"]}, {S["t"]}')
    e(f's_mov_b32 {S["spin"]}, 0')
    e('L_spin_loop%=:')
    e(f'ds_read_b32 {V["spin"]}, {V["spin"]}')                    # (address register reused for the value: reloaded below)
    e('s_waitcnt lgkmcnt(0)')
    e(f'v_readfirstlane_b32 {S["p"]}, {V["spin"]}')
    e(f'v_mov_b32 {V["spin"]}, {S["t"]}')
    e(f's_add_i32 {S["want"]}, {S["it"]}, 2')
    e(f's_cmp_ge_i32 {S["p"]}, {S["want"]}')
    e('s_cbranch_scc1 L_ready%=')
    e('s_sleep 1')
    e(f's_add_i32 {S["spin"]}, {S["spin"]}, 1')
    e(f's_cmp_lt_u32 {S["spin"]}, {S["limit"]}')
    e('s_cbranch_scc1 L_spin_loop%=')
    e(f's_mov_b32 {S["fault"]}, 2')                               # kFaultPcConsumerWait
    e(f's_mov_b32 {S["limit"]}, 0')
    e('s_branch L_ready%=')
    e('L_end%=:')
    OPS = dict(TILE_OPERANDS)
    return st


def build_loop_unrolled():
    """The whole-loop block unrolled over the ring's eight slots: every LDS address of a tile is a per-lane base
    register (slots 0-3: the inputs v[138:141]; slots 4-7: the same plus four tiles) and an immediate offset, flags
    and counters are immediates too.  What is left of the loop's bookkeeping per tile: the test of the next tile's
    flag, the partner priority, the progress counter and the loop count -- ~16 instructions instead of 48 (a
    wavefront does not issue in the shadow of its own matrix instructions, tools/fp64_issue_probe.py: every one of
    them lengthens the consumer's chain).  Same inputs and outputs as build_loop()."""
    assert NSETS == 2 and not STAMPS
    global OPS, OFFS
    V, S = LOOP_V, LOOP_S
    v_flags = V['a_flag']
    half = [dict(a_w=V['b_w'], a_q0=V['b_q0'], a_q1=V['b_q1'], a_p=V['b_p']),
            dict(a_w=V['a_w'], a_q0=V['a_q0'], a_q1=V['a_q1'], a_p=V['a_p'])]
    st = Stream([])
    e = st.emit
    for r in ('it', 'prio', 'fault', 'fnext'):
        e(f's_mov_b32 {S[r]}, 0')
    for r in ('flag', 'partner', 'progress'):
        e(f'v_mov_b32 {V[r]}, 0')
    e(f'v_mov_b32 {V["one"]}, 1')
    e(f'v_mov_b32 {v_flags}, {S["flags"]}')
    e(f's_xor_b32 {S["t"]}, {S["me"]}, 4')
    e(f'v_lshl_add_u32 {V["a_partner"]}, {S["t"]}, 2, {v_flags}')
    e(f'v_lshl_add_u32 {V["a_prog"]}, {S["me"]}, 2, {v_flags}')
    for name in ('a_w', 'a_q0', 'a_q1', 'a_p'):
        e(f'v_add_u32_e32 {half[1][name]}, {4*TILE_BYTES}, {half[0][name]}')
    if LOOP_CLOCK:
        e('s_memtime s[52:53]')
        e('s_waitcnt lgkmcnt(0)')
    if PRIO_LOW:
        e(f's_setprio {PRIO_LOW}')
    for k in range(8):
        cur, nxt = k, (k + 1) & 7
        e(f'L_slot{k}_%=:')
        prio_here, prio_before = k % PRIO_EVERY == 0, ((k - 1) & 7) % PRIO_EVERY == 0
        st.fifo = list(ENTRY if prio_before else ENTRY[:-1])
        OPS = dict(LOOP_V)
        OPS.update(a_q1=half[cur >> 2]['a_q1'], a_p1=half[cur >> 2]['a_p'],
                   a_w=half[nxt >> 2]['a_w'], a_q0=half[nxt >> 2]['a_q0'], a_p=half[nxt >> 2]['a_p'])
        OFFS = dict(a_q1=(cur & 3)*TILE_BYTES, a_p1=(cur & 3)*TILE_BYTES,
                    a_w=(nxt & 3)*TILE_BYTES, a_q0=(nxt & 3)*TILE_BYTES, a_p=(nxt & 3)*TILE_BYTES)
        st.lds(f'ds_read_b32 {V["flag"]}, {v_flags} offset:{4*((k + 2) & 7)}', 'flag')
        if prio_here:
            st.lds(f'ds_read_b32 {V["partner"]}, {V["a_partner"]} offset:64', 'partner')

        def inner(stage):
            if stage == 2:
                st.lds(f'ds_read_b128 {v4(PSI)}, {A("a_p1", 64)}', 'psi')
                st.lds(f'ds_read_b128 {v4(Q01)}, {A("a_q1")}', 'q01')
            elif stage == 4:
                st.lds(f'ds_read_b128 {v4(Q23)}, {A("a_q1", 4096)}', 'q23')
        if PHASE_PRIO:
            e(f's_setprio {PHASE_PRIO.split(",")[0]}')
        vector_part(st, 0, inner)
        if PHASE_PRIO:
            e(f's_setprio {PHASE_PRIO.split(",")[1]}')
        # tile it + 1 published?  (its flag was read a tile ago; the slow path spins, bounded; no wait behind the last tile)
        glue0 = [f's_add_i32 {S["t"]}, {S["it"]}, 2',
                 (f's_cmp_ge_i32 {S["fnext"]}, {S["t"]}',
                  f's_cbranch_scc1 L_ready{k}_%=',
                  f's_add_i32 {S["t2"]}, {S["it"]}, 1',
                  f's_cmp_ge_i32 {S["t2"]}, {S["nit"]}',
                  f's_cbranch_scc1 L_ready{k}_%=',
                  f's_branch L_spin{k}_%=',
                  f'L_ready{k}_%=:')]
        if PRIO_EARLY and prio_here and not PHASE_PRIO:
            st.need('partner')
            glue0 = glue0 + [
                 f'v_readfirstlane_b32 {S["p"]}, {V["partner"]}',
                 f's_add_i32 {S["t"]}, {S["it"]}, {LOCKSTEP}',
                 (f's_cmp_gt_i32 {S["p"]}, {S["t"]}',
                  f's_cselect_b32 {S["want"]}, 1, 0',
                  f's_cmp_eq_u32 {S["want"]}, {S["prio"]}',
                  f's_cbranch_scc1 L_prio_done{k}_%=',
                  f's_mov_b32 {S["prio"]}, {S["want"]}',
                  f's_cmp_eq_u32 {S["want"]}, 1',
                  f's_cbranch_scc1 L_prio_hi{k}_%=',
                  f's_setprio {PRIO_LOW}',
                  f's_branch L_prio_done{k}_%=',
                  f'L_prio_hi{k}_%=:',
                  f's_setprio {PRIO_HIGH}',
                  f'L_prio_done{k}_%=:')]
        matrix_part(st, 0, glue0)
        st.fifo = [t if t in ('q01', 'q23', 'psi', 'flag', 'partner') else 'old' for t in st.fifo]
        if PHASE_PRIO:
            e(f's_setprio {PHASE_PRIO.split(",")[0]}')
        vector_part(st, 1, lambda stage: next_tile_requests(st, stage))
        if PHASE_PRIO:
            e(f's_setprio {PHASE_PRIO.split(",")[1]}')
        glue1 = [f'v_readfirstlane_b32 {S["fnext"]}, {V["flag"]}',
                 f'v_add_u32_e32 {V["progress"]}, 1, {V["progress"]}']
        if prio_here and not PHASE_PRIO and not PRIO_EARLY:
            glue1 += [
                 f'v_readfirstlane_b32 {S["p"]}, {V["partner"]}',
                 f's_add_i32 {S["t"]}, {S["it"]}, {LOCKSTEP}',
                 (f's_cmp_gt_i32 {S["p"]}, {S["t"]}',
                  f's_cselect_b32 {S["want"]}, 1, 0',
                  f's_cmp_eq_u32 {S["want"]}, {S["prio"]}',
                  f's_cbranch_scc1 L_prio_done{k}_%=',
                  f's_mov_b32 {S["prio"]}, {S["want"]}',
                  f's_cmp_eq_u32 {S["want"]}, 1',
                  f's_cbranch_scc1 L_prio_hi{k}_%=',
                  f's_setprio {PRIO_LOW}',
                  f's_branch L_prio_done{k}_%=',
                  f'L_prio_hi{k}_%=:',
                  f's_setprio {PRIO_HIGH}',
                  f'L_prio_done{k}_%=:')]
        matrix_part(st, 1, glue1)
        e('s_mov_b64 exec, 1')
        st.lds(f'ds_add_u32 {v_flags}, {V["one"]} offset:{32 + 4*k}', 'done')
        if prio_here:
            st.lds(f'ds_write_b32 {V["a_prog"]}, {V["progress"]} offset:64', 'prog')
        e('s_mov_b64 exec, -1')
        assert st.fifo == (ENTRY if prio_here else ENTRY[:-1]), st.fifo
        e(f's_add_i32 {S["it"]}, {S["it"]}, 1')
        e(f's_cmp_ge_i32 {S["it"]}, {S["nit"]}')
        e('s_cbranch_scc1 L_exit%=')
        if k == 7:
            e('s_branch L_slot0_%=')
    e('L_exit%=:')
    e('s_waitcnt lgkmcnt(0)')
    e('s_setprio 0')
    if LOOP_CLOCK:
        e('s_memtime s[54:55]')
        e('s_waitcnt lgkmcnt(0)')
    e('s_nop 15')
    e('s_nop 15')
    e('s_branch L_end%=')
    for k in range(8):
        # ---- slow path of slot k: tile it + 1 (slot k + 1) is not published yet ----
        e(f'L_spin{k}_%=:')
        e(f's_cmp_eq_u32 {S["limit"]}, 0')
        e(f's_cbranch_scc1 L_ready{k}_%=')                        # a wait ran out earlier: no more waiting
        e(f's_mov_b32 {S["spin"]}, 0')
        e(f's_add_i32 {S["want"]}, {S["it"]}, 2')
        e(f'L_spin_loop{k}_%=:')
        e(f'ds_read_b32 {V["spin"]}, {v_flags} offset:{4*((k + 1) & 7)}')
        e('s_waitcnt lgkmcnt(0)')
        e(f'v_readfirstlane_b32 {S["p"]}, {V["spin"]}')
        e(f's_cmp_ge_i32 {S["p"]}, {S["want"]}')
        e(f's_cbranch_scc1 L_ready{k}_%=')
        e('s_sleep 1')
        e(f's_add_i32 {S["spin"]}, {S["spin"]}, 1')
        e(f's_cmp_lt_u32 {S["spin"]}, {S["limit"]}')
        e(f's_cbranch_scc1 L_spin_loop{k}_%=')
        e(f's_mov_b32 {S["fault"]}, 2')                           # kFaultPcConsumerWait
        e(f's_mov_b32 {S["limit"]}, 0')
        e(f's_branch L_ready{k}_%=')
    e('L_end%=:')
    OPS = dict(TILE_OPERANDS)
    OFFS = {}
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
    for stage in (1, 2, 3, 4):
        next_tile_requests(st, stage)
    st.emit('s_waitcnt lgkmcnt(0)')
    return st


def main():
    print('// GENERATED by tools/gen_pq_consumer.py -- do not edit; see that file for the register map.')
    dump('FFK_PQ_CONSUMER_ASM', build(False))
    dump('FFK_PQ_CONSUMER_PROLOGUE_ASM', prologue())
    if NSETS == 2 and not STAMPS and not DROP:
        dump('FFK_PQ_CONSUMER_LOOP_ASM', build_loop() if ROLLED else build_loop_unrolled())
        if LOOP_CLOCK:
            print('#define FFK_PQ_LOOP_CLOCK 1')
        print(f'#define FFK_PQ_TILE_BYTES {TILE_BYTES}')
        print('#define FFK_PQ_LOOP_CLOBBERS \\')
        regs = [f'"v{r}"' for r in range(NTMP)] + [f'"v{r}"' for r in range(124, 138)]
        regs += [f'"s{r}"' for r in list(range(40, 48)) + [49, 50, 51]] + ['"scc"', '"memory"']
        print('    ' + ', '.join(regs))
    print('#define FFK_PQ_CONSUMER_CLOBBERS \\')
    print('    ' + ', '.join(f'"v{r}"' for r in range(NTMP)) + ', "memory"')


if __name__ == '__main__':
    main()
