"""Generates filter_functions_amd/csrc/ctrl_pq_consumer.inc: the consumer wavefront's WHOLE tile loop of
the d = 4 matrix-core accumulate kernel (ctrl_pq.hip) as one inline-asm block per operator count
NC = 1, 2, 3, with an explicit instruction order and exact s_waitcnt counts.

    python tools/gen_pq_consumer.py > filter_functions_amd/csrc/ctrl_pq_consumer.inc

(tests/test_abi.py::test_generated_consumers_are_current asserts that the committed file is what this
script prints.)

Why generated assembly (profiles/r05_b_*, r05_l_*, r05_m_*): a tile is 8 NC + 13 (+ 3 NC adds) vector and 6 NC
matrix instructions fed by 4 NC + 10 LDS reads, two consumers per SIMD.  To keep the SIMD busy the NEXT tile's
operands must be requested from inside the current tile -- behind the last vector instruction that reads the
registers they land in, before the last set's matrix instructions -- and the flag of the tile after that must
be read a tile ahead.  hipcc moves the vector work of the last set behind those requests (two live copies of
the operands, accumulators spilled inside the loop), sinks the reads to their first use, or reorders the
interleaved chains.  Here every LDS operation of the loop is in the block, so the in-order LDS queue is known
exactly and every wait names the number of younger operations that may still fly; the loop is unrolled over
the ring's eight slots so that every LDS address is a per-lane base register plus an immediate (a wavefront
does not issue in the shadow of its own matrix instructions: every bookkeeping instruction lengthens the
consumer's chain, tools/fp64_issue_probe.py).  The round-5 generator with its measured-and-dropped options
(rolled loop, per-tile block, merged waits, sparse priority, four sets per consumer, clocks) is
tools/tuning/gen_pq_consumer_r5.py.

Mathematics per tile, for the set s (four frequencies, one per 4x4x4 block of the matrix instruction) and the
operator a (ctrl_pq.hip):
    c    = psi conj(T[m][i])                               (per lane (i, block, m): the A operand)
    z_a  = sum_n q[m][n] W_a[m][n][j]                      (per lane (m, block, j): the B operand)
    P1_a += cr^T zr_a,  P2_a += ci^T zi_a,  P3_a += (cr + ci)^T (zr_a + zi_a)

Registers of the NC-operator block (fixed; `{v[a:b]}` constraints in the generated C++ wrapper):
    temporaries  zr_a v[4a], zi_a v[4a+2], zs_a v[4 NC + 2a], cr, ci, cs behind them     6 NC + 6
    W_a[n] = (re, im) at W0 + 4 (4 a + n)                                                 16 NC
    q01 (q[m][0], q[m][1]), q23, psi (re, im), T (tr, ti)                                 16
    accumulators P_k of (operator a, set s) at ACC0 + 2 (3 (2 a + s) + k)                 12 NC
    loop registers (LDS addresses of both ring halves, flags, counters)                   18
i.e. 142 / 108 / 74 for NC = 3 / 2 / 1.  Scalars: inputs s[36:39] = (tiles, spin limit, LDS address of the
flags, consumer number), output s48 = 0 or the fault code of a flag wait that ran out; s[40:51] scratch.
LDS operations of a tile, in queue order: flag of tile it + 2, partner's progress | set 1's psi, q01, q23 |
next tile's W[.][0], T, psi, q01, W[.][1], W[.][2], q23, W[.][3] | done counter, own progress.
"""
RING = 8
SETS = 2                # sets of four frequencies per consumer
W_BYTES = 1024          # per operator: [n][m][j] complex
FAULT_CONSUMER_WAIT = 2     # ffk_internal.h kFaultPcConsumerWait


def v(r):
    return f'v[{r}:{r + 1}]'


def v4(r):
    return f'v[{r}:{r + 3}]'


class Map:
    """register map and tile geometry of the NC-operator block"""

    def __init__(self, nc):
        self.nc = nc
        self.zr = [4*a for a in range(nc)]
        self.zi = [4*a + 2 for a in range(nc)]
        self.zs = [4*nc + 2*a for a in range(nc)]
        self.cr, self.ci, self.cs = 6*nc, 6*nc + 2, 6*nc + 4
        self.ntmp = 6*nc + 6
        self.w0 = self.ntmp
        self.q01 = self.w0 + 16*nc
        self.q23, self.psi, self.t = self.q01 + 4, self.q01 + 8, self.q01 + 12
        self.acc0 = self.q01 + 16
        self.loop0 = self.acc0 + 6*SETS*nc
        names = ('a_w', 'a_q0', 'a_p', 'a_q1', 'unused', 'a_flag', 'unused2', 'progress', 'a_partner', 'a_prog', 'one',
                 'flag', 'partner', 'spin', 'b_w', 'b_q0', 'b_q1', 'b_p')
        self.V = {n: f'v{self.loop0 + k}' for k, n in enumerate(names)}
        self.vend = self.loop0 + len(names)
        self.S = dict(nit='s36', limit='s37', flags='s38', me='s39', it='s40', t='s43', fnext='s44', p='s45', prio='s46',
                      want='s47', fault='s48', spin='s49', t2='s50')
        self.t_off = nc*W_BYTES                       # (tr, ti) pairs behind the operators' W, same lane index
        self.tile_bytes = (1152 + nc*128 + 32)*8      # ctrl_pq.hip: pq_tile_doubles(NC) * 8 (checked there)
        # the LDS queue at the top of a tile: what the previous tile left in flight
        self.entry = ['w0']*nc + ['T', 'psi', 'q01'] + ['w1']*nc + ['w2']*nc + ['q23'] + ['w3']*nc + ['done', 'prog']

    def wre(self, a, n):
        return self.w0 + 4*(4*a + n)

    def acc(self, a, s, k):
        return self.acc0 + 2*(3*(SETS*a + s) + k)


class Stream:
    def __init__(self, fifo):
        self.lines = []
        self.fifo = list(fifo)      # tags of LDS operations in flight, oldest first

    def emit(self, text):
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


def next_tile_requests(st, M, adr, stage):
    """The next tile's operands, each requested right behind the LAST use of the registers it lands in (the last
    set's vector stages): W[.][n] is dead after stage n + 1, T, psi and q01 after stage 2, q23 after stage 4.
    (All of them in one burst behind stage 4 cost 3.7 us of the consumers' 54.7: profiles/r05_b_*.)"""
    def w(n):
        for a in range(M.nc):
            st.lds(f'ds_read_b128 {v4(M.wre(a, n))}, {adr("a_w", a*W_BYTES + n*256)}', f'w{n}')
    if stage == 1:
        w(0)
    elif stage == 2:
        st.lds(f'ds_read_b128 {v4(M.t)}, {adr("a_w", M.t_off)}', 'T')
        st.lds(f'ds_read_b128 {v4(M.psi)}, {adr("a_p")}', 'psi')
        st.lds(f'ds_read_b128 {v4(M.q01)}, {adr("a_q0")}', 'q01')
        w(1)
    elif stage == 3:
        w(2)
    elif stage == 4:
        st.lds(f'ds_read_b128 {v4(M.q23)}, {adr("a_q0", 4096)}', 'q23')
        w(3)


def vector_part(st, M, after_stage):
    q = [M.q01, M.q01 + 2, M.q23, M.q23 + 2]
    pr, pi = M.psi, M.psi + 2
    tr, ti = M.t, M.t + 2
    # stage 1
    st.need('T', 'psi', 'q01', 'w0')
    st.emit(f'v_mul_f64 {v(M.cr)}, {v(pi)}, {v(ti)}')
    st.emit(f'v_mul_f64 {v(M.ci)}, {v(pi)}, {v(tr)}')
    for a in range(M.nc):
        st.emit(f'v_mul_f64 {v(M.zr[a])}, {v(q[0])}, {v(M.wre(a, 0))}')
        st.emit(f'v_mul_f64 {v(M.zi[a])}, {v(q[0])}, {v(M.wre(a, 0) + 2)}')
    after_stage(1)
    # stage 2
    st.need('w1')
    st.emit(f'v_fma_f64 {v(M.cr)}, {v(pr)}, {v(tr)}, {v(M.cr)}')
    st.emit(f'v_fma_f64 {v(M.ci)}, -{v(pr)}, {v(ti)}, {v(M.ci)}')
    for a in range(M.nc):
        st.emit(f'v_fma_f64 {v(M.zr[a])}, {v(q[1])}, {v(M.wre(a, 1))}, {v(M.zr[a])}')
        st.emit(f'v_fma_f64 {v(M.zi[a])}, {v(q[1])}, {v(M.wre(a, 1) + 2)}, {v(M.zi[a])}')
    after_stage(2)
    # stage 3
    st.need('q23', 'w2')
    st.emit(f'v_add_f64 {v(M.cs)}, {v(M.cr)}, {v(M.ci)}')
    for a in range(M.nc):
        st.emit(f'v_fma_f64 {v(M.zr[a])}, {v(q[2])}, {v(M.wre(a, 2))}, {v(M.zr[a])}')
        st.emit(f'v_fma_f64 {v(M.zi[a])}, {v(q[2])}, {v(M.wre(a, 2) + 2)}, {v(M.zi[a])}')
    after_stage(3)
    # stage 4
    st.need('w3')
    for a in range(M.nc):
        st.emit(f'v_fma_f64 {v(M.zr[a])}, {v(q[3])}, {v(M.wre(a, 3))}, {v(M.zr[a])}')
        st.emit(f'v_fma_f64 {v(M.zi[a])}, {v(q[3])}, {v(M.wre(a, 3) + 2)}, {v(M.zi[a])}')
    after_stage(4)


def matrix_part(st, M, s, glue):
    """the 3 NC matrix instructions of a set; `glue`: scalar / address instructions of the loop, dealt out behind
    the matrix instructions"""
    for a in range(M.nc):
        st.emit(f'v_add_f64 {v(M.zs[a])}, {v(M.zr[a])}, {v(M.zi[a])}')
    glue = list(glue)
    per = -(-len(glue)//(3*M.nc)) if glue else 0
    for a in range(M.nc):
        for k, (src_a, src_b) in enumerate(((M.cr, M.zr[a]), (M.ci, M.zi[a]), (M.cs, M.zs[a]))):
            st.emit(f'v_mfma_f64_4x4x4_4b_f64 {v(M.acc(a, s, k))}, {v(src_a)}, {v(src_b)}, {v(M.acc(a, s, k))}')
            for _ in range(per):
                if glue:
                    item = glue.pop(0)          # a tuple: instructions with branches among them, kept together
                    for line in ((item,) if isinstance(item, str) else item):
                        st.emit(line)
    assert not glue


def priority_rule(M, k):
    """whoever is behind its SIMD partner by a tile or more raises its priority (the partner's count was read at
    the top of this tile); looser rules are slower, profiles/r05_v_*"""
    S, V = M.S, M.V
    return [f'v_readfirstlane_b32 {S["p"]}, {V["partner"]}',
            f's_add_i32 {S["t"]}, {S["it"]}, 0',
            (f's_cmp_gt_i32 {S["p"]}, {S["t"]}',
             f's_cselect_b32 {S["want"]}, 1, 0',
             f's_cmp_eq_u32 {S["want"]}, {S["prio"]}',
             f's_cbranch_scc1 L_prio_done{k}_%=',
             f's_mov_b32 {S["prio"]}, {S["want"]}',
             f's_cmp_eq_u32 {S["want"]}, 1',
             f's_cbranch_scc1 L_prio_hi{k}_%=',
             's_setprio 0',
             f's_branch L_prio_done{k}_%=',
             f'L_prio_hi{k}_%=:',
             's_setprio 1',
             f'L_prio_done{k}_%=:')]


def build_loop(M):
    """The consumer's whole tile loop, unrolled over the ring's eight slots: every LDS address of a tile is a
    per-lane base register (slots 0-3: the inputs b_*; slots 4-7: the same plus four tiles) and an immediate
    offset, flags and counters are immediates too.  What is left of the loop's bookkeeping per tile: the test of
    the next tile's flag, the partner priority, the progress counter and the loop count.  The first tile's
    operands are in their registers (prologue block)."""
    V, S = M.V, M.S
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
        e(f'v_add_u32_e32 {half[1][name]}, {4*M.tile_bytes}, {half[0][name]}')
    for k in range(RING):
        cur, nxt = k, (k + 1) % RING
        e(f'L_slot{k}_%=:')
        st.fifo = list(M.entry)
        regs = dict(a_q1=half[cur >> 2]['a_q1'], a_p1=half[cur >> 2]['a_p'],
                    a_w=half[nxt >> 2]['a_w'], a_q0=half[nxt >> 2]['a_q0'], a_p=half[nxt >> 2]['a_p'])
        offs = dict(a_q1=(cur & 3)*M.tile_bytes, a_p1=(cur & 3)*M.tile_bytes,
                    a_w=(nxt & 3)*M.tile_bytes, a_q0=(nxt & 3)*M.tile_bytes, a_p=(nxt & 3)*M.tile_bytes)

        def adr(name, offset=0):
            off = offset + offs[name]
            assert 0 <= off < 65536
            return f'{regs[name]} offset:{off}' if off else regs[name]

        st.lds(f'ds_read_b32 {V["flag"]}, {v_flags} offset:{4*((k + 2) % RING)}', 'flag')
        st.lds(f'ds_read_b32 {V["partner"]}, {V["a_partner"]} offset:64', 'partner')

        def second_set(stage):
            # set 1's psi, q01 behind stage 2 (their registers' last use), q23 behind stage 4: they arrive during
            # the rest of the set and its matrix instructions
            if stage == 2:
                st.lds(f'ds_read_b128 {v4(M.psi)}, {adr("a_p1", 64)}', 'psi')
                st.lds(f'ds_read_b128 {v4(M.q01)}, {adr("a_q1")}', 'q01')
            elif stage == 4:
                st.lds(f'ds_read_b128 {v4(M.q23)}, {adr("a_q1", 4096)}', 'q23')
        vector_part(st, M, second_set)
        # tile it + 1 published?  (its flag was read a tile ago; the slow path spins, bounded; no wait behind the last tile)
        glue0 = [f's_add_i32 {S["t"]}, {S["it"]}, 2',
                 (f's_cmp_ge_i32 {S["fnext"]}, {S["t"]}',
                  f's_cbranch_scc1 L_ready{k}_%=',
                  f's_add_i32 {S["t2"]}, {S["it"]}, 1',
                  f's_cmp_ge_i32 {S["t2"]}, {S["nit"]}',
                  f's_cbranch_scc1 L_ready{k}_%=',
                  f's_branch L_spin{k}_%=',
                  f'L_ready{k}_%=:')]
        matrix_part(st, M, 0, glue0)
        # T and W are still this tile's: only q01, q23, psi have to arrive
        st.fifo = [t if t in ('q01', 'q23', 'psi', 'flag', 'partner') else 'old' for t in st.fifo]
        vector_part(st, M, lambda stage: next_tile_requests(st, M, adr, stage))
        glue1 = [f'v_readfirstlane_b32 {S["fnext"]}, {V["flag"]}',
                 f'v_add_u32_e32 {V["progress"]}, 1, {V["progress"]}'] + priority_rule(M, k)
        matrix_part(st, M, 1, glue1)
        # hand the slot back: lane 0 counts this consumer in and publishes its progress (LDS operations of a
        # wavefront execute in order: both are behind the tile's reads without a wait)
        e('s_mov_b64 exec, 1')
        st.lds(f'ds_add_u32 {v_flags}, {V["one"]} offset:{4*RING + 4*k}', 'done')
        st.lds(f'ds_write_b32 {V["a_prog"]}, {V["progress"]} offset:64', 'prog')
        e('s_mov_b64 exec, -1')
        assert st.fifo == M.entry, st.fifo
        e(f's_add_i32 {S["it"]}, {S["it"]}, 1')
        e(f's_cmp_ge_i32 {S["it"]}, {S["nit"]}')
        e('s_cbranch_scc1 L_exit%=')
        if k == RING - 1:
            e('s_branch L_slot0_%=')
    e('L_exit%=:')
    e('s_waitcnt lgkmcnt(0)')
    e('s_setprio 0')
    e('s_nop 15')
    e('s_nop 15')
    e('s_branch L_end%=')
    for k in range(RING):
        # ---- slow path of slot k: tile it + 1 (slot k + 1) is not published yet ----
        e(f'L_spin{k}_%=:')
        e(f's_cmp_eq_u32 {S["limit"]}, 0')
        e(f's_cbranch_scc1 L_ready{k}_%=')                        # a wait ran out earlier: no more waiting
        e(f's_mov_b32 {S["spin"]}, 0')
        e(f's_add_i32 {S["want"]}, {S["it"]}, 2')
        e(f'L_spin_loop{k}_%=:')
        e(f'ds_read_b32 {V["spin"]}, {v_flags} offset:{4*((k + 1) % RING)}')
        e('s_waitcnt lgkmcnt(0)')
        e(f'v_readfirstlane_b32 {S["p"]}, {V["spin"]}')
        e(f's_cmp_ge_i32 {S["p"]}, {S["want"]}')
        e(f's_cbranch_scc1 L_ready{k}_%=')
        e('s_sleep 1')
        e(f's_add_i32 {S["spin"]}, {S["spin"]}, 1')
        e(f's_cmp_lt_u32 {S["spin"]}, {S["limit"]}')
        e(f's_cbranch_scc1 L_spin_loop{k}_%=')
        e(f's_mov_b32 {S["fault"]}, {FAULT_CONSUMER_WAIT}')
        e(f's_mov_b32 {S["limit"]}, 0')
        e(f's_branch L_ready{k}_%=')
    e('L_end%=:')
    return st


def prologue(M):
    """the first tile's operands (the compiler then has no LDS read of its own pending on the fixed registers,
    and does not put a full wait in front of the loop's block)"""
    st = Stream([])

    def adr(name, offset=0):
        return f'%[{name}] offset:{offset}' if offset else f'%[{name}]'
    for stage in (1, 2, 3, 4):
        next_tile_requests(st, M, adr, stage)
    st.emit('s_waitcnt lgkmcnt(0)')
    return st


def counts(st):
    n_valu = sum(1 for ln in st.lines if ln.startswith('v_') and 'mfma' not in ln)
    n_mfma = sum(1 for ln in st.lines if 'mfma' in ln)
    n_lds = sum(1 for ln in st.lines if ln.startswith('ds_'))
    return f'{n_valu} vector, {n_mfma} matrix instructions, {n_lds} LDS operations'


def asm_string(st, indent):
    return '\n'.join(f'{indent}"{ln}\\n\\t"' for ln in st.lines)


def chunks(first, count):
    """`count` consecutive registers from `first` as (first, size) pieces of 16, 8 or 4 registers"""
    out = []
    while count:
        size = next(s for s in (16, 8, 4) if s <= count)
        out.append((first, size))
        first += size
        count -= size
    return out


VEC = {16: 'double8_t', 8: 'double4_t', 4: 'double2_t'}


def wrapper(M):
    nc = M.nc
    w_regs = [(M.w0 + 16*a, 16) for a in range(nc)]
    q_reg = (M.q01, 16)
    acc_regs = chunks(M.acc0, 6*SETS*nc)

    def rng(r):
        return f'v[{r[0]}:{r[0] + r[1] - 1}]'
    loop, pro = build_loop(M), prologue(M)
    operands = [f'W{a}' for a in range(nc)] + ['Q']
    lines = []
    a = lines.append
    a(f'// ---- {nc} operator{"s" if nc > 1 else ""} per block: {M.vend} vector registers; one tile = {counts(loop)} / {RING}')
    a('template <>')
    a(f'struct PqConsumer<{nc}> {{')
    a(f'    static constexpr int kTileBytes = {M.tile_bytes};')
    a(f'    static constexpr int kVgprs = {M.vend};')
    a('    // b_*: LDS addresses of this lane\'s W / T, q (set 0), q (set 1), psi in slot 0; flags_b: LDS address of the')
    a('    // flags; P[3 (2 a + s) + k]: the sums P_k of (operator a, set s).  Returns 0 or the fault code of a wait that ran out.')
    a('    static __device__ __forceinline__ int run(unsigned b_w, unsigned b_q0, unsigned b_q1, unsigned b_p, int n_it,')
    a(f'                                              int spin_limit, unsigned flags_b, int me, double (&P)[{6*nc}]) {{')
    for name, r in zip(operands, w_regs + [q_reg]):
        a(f'        {VEC[r[1]]} {name};        // {rng(r)}')
    for k, r in enumerate(acc_regs):
        a(f'        {VEC[r[1]]} A{k} = 0.0;  // {rng(r)}')
    a('        asm volatile(')
    a(asm_string(pro, '            '))
    outs = ', '.join(f'"=&{{{rng(r)}}}"({n})' for n, r in zip(operands, w_regs + [q_reg]))
    a(f'            : {outs}')
    a('            : [a_w] "v"(b_w), [a_p] "v"(b_p), [a_q0] "v"(b_q0)')
    a('            : "memory");')
    a('        int4_t sarg = {n_it, spin_limit, static_cast<int>(flags_b), me};')
    a('        const int4_t varg = {static_cast<int>(b_w), static_cast<int>(b_q0), static_cast<int>(b_q1), static_cast<int>(b_p)};')
    a('        int fault_code;')
    a('        asm volatile(')
    a(asm_string(loop, '            '))
    inouts = [f'"+{{{rng(r)}}}"({n})' for n, r in zip(operands, w_regs + [q_reg])]
    inouts += [f'"+{{{rng(r)}}}"(A{k})' for k, r in enumerate(acc_regs)]
    a('            : ' + ', '.join(inouts) + ',')
    a('              "+{s[36:39]}"(sarg), "={s48}"(fault_code)')
    b = int(M.V['b_w'][1:])
    a(f'            : "{{v[{b}:{b + 3}]}}"(varg)')
    clob = [f'"v{r}"' for r in range(M.ntmp)] + [f'"v{r}"' for r in range(M.loop0, b)]
    clob += [f'"s{r}"' for r in list(range(40, 48)) + [49, 50, 51]] + ['"scc"', '"memory"']
    a('            : ' + ', '.join(clob) + ');')
    a('        // (the compiler does not know that matrix instructions wrote the accumulators: keep the vector')
    a('        // instructions that read them next out of their shadow)')
    a('        asm volatile("s_waitcnt lgkmcnt(0)\\n\\ts_nop 15\\n\\ts_nop 15"')
    a('                     : ' + ', '.join(f'"+{{{rng(r)}}}"({n})' for n, r in zip(operands, w_regs + [q_reg])))
    a('                     :')
    a('                     : "memory");')
    idx = 0
    for k, r in enumerate(acc_regs):
        for c in range(r[1]//2):
            a(f'        P[{idx}] = A{k}[{c}];')
            idx += 1
    assert idx == 6*nc
    a('        return fault_code;')
    a('    }')
    a('};')
    return '\n'.join(lines)


def main():
    print('// GENERATED by tools/gen_pq_consumer.py -- do not edit; see that file for the register maps.')
    print('// Included by ctrl_pq.hip inside namespace ffk::{anonymous}, behind the declaration of PqConsumer<NC>.')
    for nc in (1, 2, 3):
        print(wrapper(Map(nc)))


if __name__ == '__main__':
    main()
