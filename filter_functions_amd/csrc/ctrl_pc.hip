// ctrl_pc.hip -- K3p: the control-matrix accumulation for d = 4 with SPECIALISED wavefronts.
// Same mathematics, inputs, output layout and per-lane arithmetic as ctrl.hip (one frequency per
// lane, 64 per wave; Y_a(w) = sum_g T_g^dag [Bbar_a o E_g(w)] T_g), different division of labour:
// every sub-chunk of a block consists of ONE producer wavefront, which generates the whole integral
// tile e^{i w t_g} I^(g)(w) of the next segment (and moves the segment's operands and table rows
// from global memory into LDS), and NC consumer wavefronts, one per noise operator, which do
// nothing but the two d x d products on the tile of the current segment.  In ctrl.hip every wave
// alternates between the two phases and all waves of a block are in the same phase at the same
// time; the latency-bound generation and the FMA-bound contraction never overlap there
// (profiles/r01_c_*: generation alone 43 us, contraction alone 80 us, together 111 us).  Here they
// run side by side on every SIMD.  Producers sit on different SIMDs (sub-chunk s -> wave s of its
// group): with four sub-chunks of one producer and three consumers every SIMD hosts exactly one
// producer and three consumers (checked with HW_ID stamps, profiles/r04_a_*).
// The consumers take T_g through SCALAR loads (wave-uniform addresses, issued one segment ahead at
// the end of the previous one): 32 doubles in SGPRs feed v_fma_f64 directly instead of occupying
// 64 VGPRs, which brings the kernel to 104 VGPRs, i.e. four wavefronts per SIMD.
//
// Round 4 (profiles/r04_*): three changes, each from a measurement.
//  * REAL tile.  E = psi e^{ib} q with q = 2 sin(a + b)/x real and psi = e^{i w t_g} e^{ia} per
//    (segment, frequency) (ffk_math.h).  The producer publishes the 13 distinct q and psi only; the
//    phase e^{ib}, Bbar and the first T are folded into W_a[m][n][j] = Bbar_a[m][n] e^{i b_mn} T[n][j]
//    (frequency independent, one element per producer lane).  Consumers: Z[m][j] = sum_n q[m][n]
//    W[m][n][j] (real x complex), z = psi Z, Y[i][j] += conj(T[m][i]) z: 448 instead of 576 vector
//    instructions per consumer and segment, ~250 instead of ~440 for the producer.
//  * Queue instead of barrier.  The per-wavefront timeline (tools/trace_pc.py) of the round-3
//    kernel: wavefronts of a SIMD finish one after the other (the arbiter serves the oldest), the
//    block waits ~400 cycles for the release of its 16-wavefront s_barrier and every consumer starts
//    with the same burst of LDS reads: ~1000 of a step's ~9700 cycles without issue.  Now each
//    sub-chunk is a single-producer / NC-consumer queue over its two tile buffers, with a `ready`
//    count written by the producer and a `done` count per consumer in LDS: a consumer that finds its
//    tile ready (the usual case) never waits, the producer sleeps until its buffer is free.  The
//    sub-chunks drift apart, so a SIMD always has wavefronts in other phases to issue from; because
//    the arbiter would let the oldest sub-chunk run away, a consumer that is k tiles behind the most
//    advanced sub-chunk raises its priority to min(k, 2) (soft lockstep: 71.1 -> 67.8 us per step).
//  * The consumer's segment is ONE generated asm block (ctrl_pc_consumer.inc): the folded operands
//    have no reuse, so their LDS reads must fly ahead of their use inside a 112-VGPR budget; hipcc
//    serialised them (read, s_waitcnt 0, two FMAs) or spilled the accumulators.
// Same box, bench schedule: round-3 kernel 78.1 us per step, this one 67.4-70.7.
// Dropped (lost their A/B, records in profiles/r03_c_*, r03_u_*, r04_*): matrix-core consumers (two
// forms, also under sustained load), Bbar folded into T with a complex tile, T_g from LDS, four
// wavefronts meeting at an LDS counter, one s_barrier per segment (70.1 vs 67.4 us with this tile).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "ffk_internal.h"
#ifndef FFK_PC_CONSUMER_INC   /* tuning builds: another ring depth */
#define FFK_PC_CONSUMER_INC "ctrl_pc_consumer.inc"
#endif
#include FFK_PC_CONSUMER_INC   // generated: tools/gen_pc_consumer.py

namespace ffk {
namespace {

constexpr int kPcSub = 4;               // sub-chunks per block: 4 x (1 + 3) = 16 waves = 4 per SIMD
// Bound of every flag wait.  A wait that runs out is a FAULT, not a result: the wavefront stores a code
// in the library's sticky fault word (mapped host memory, ffk_internal.h::kernel_fault_word; every host
// entry point reads it after its synchronisation and returns FFK_EKERNEL, `_dev` callers ask
// ffk_kernel_fault_status), stops waiting for the rest of the launch and runs to the end so that the
// grid drains.  -DFFK_PC_SPIN_LIMIT=n -DFFK_PC_FAULT_INJECT: the test build whose producers stop
// publishing after their second tile (tests/test_gpu_parity.py::test_flag_wait_timeout_is_an_error).
#ifndef FFK_PC_SPIN_LIMIT
#define FFK_PC_SPIN_LIMIT (1 << 21)
#endif
constexpr int kPcSpinLimit = FFK_PC_SPIN_LIMIT;
// (the word's address sits in a device global, read on the fault path only: as a kernel argument it cost
// two scalar registers that the consumers' loop spills -- 111 -> 114 VGPRs, past the 112 at which a
// second pass's small kernels still fit beside this one)
__device__ int* g_pc_fault_word = nullptr;
__device__ __noinline__ void pc_report_fault(int code) {
    int* fault = g_pc_fault_word;
    if (fault != nullptr && (threadIdx.x & 63) == 0)
        __hip_atomic_store(fault, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

#ifdef FFK_PC_CLOCK   /* tuning build: per-wavefront timeline of the d = 4 kernel (tools/trace_pc.py) */
// per wavefront kPcTraceLen words: [0] HW_ID | XCC_ID << 32, [1] role (0 producer, 1.. consumer),
// [2] shader clock at kernel entry, [3] 100 MHz ticks at entry, [4] first loop top, then per step
// (top = tile available / barrier passed, done = work finished), then loop end, kernel end (shader
// clock), kernel end (100 MHz ticks)
constexpr int kPcTraceSteps = 64, kPcTraceLen = 8 + 2*kPcTraceSteps;
__device__ unsigned long long g_pc_trace[1024*16*kPcTraceLen];
#define FFK_PC_STAMP(slot) \
    do { if (pc_tr != nullptr && lane == 0) pc_tr[slot] = __builtin_amdgcn_s_memtime(); } while (0)
#define FFK_PC_STEP_TOP(it) do { if ((it) < kPcTraceSteps) FFK_PC_STAMP(5 + 2*(it)); } while (0)
#define FFK_PC_STEP_DONE(it) do { if ((it) < kPcTraceSteps) FFK_PC_STAMP(6 + 2*(it)); } while (0)
#else
#define FFK_PC_STAMP(slot)
#define FFK_PC_STEP_TOP(it)
#define FFK_PC_STEP_DONE(it)
#endif

template <int D>
struct PcEntries {          // every entry once, the (coinciding) diagonal entries as slot 0
    static constexpr int build(int* out) {
        int n = 0;
        for (int e = 0; e < D*D; ++e) {
            if (e != 0 && e / D == e % D) continue;
            if (out) out[n] = e;
            ++n;
        }
        return n;
    }
    static constexpr int count = build(nullptr);
    struct Slots {
        int v[D*D];
    };
    static constexpr Slots make() {
        Slots s{};
        build(s.v);
        return s;
    }
    static constexpr Slots slots = make();
    // position of entry e = m*D + n in the tile: the list index (diagonal entries: 0)
    static constexpr Slots make_table() {
        Slots t{};
        for (int e = 0; e < D*D; ++e) {
            int k = 0;
            for (int x = 0; x < e; ++x)
                if (x == 0 || x / D != x % D) ++k;
            t.v[e] = (e / D == e % D) ? 0 : k;
        }
        return t;
    }
    static constexpr Slots slot_table = make_table();
};

// ---- LDS flags of the queue mode (every word is written by exactly one wavefront) ---------------
// LDS operations of one wavefront execute in order, so "data, s_waitcnt lgkmcnt(0), flag" on the
// writing side and "flag, then data" on the reading side is all the ordering there is to keep; the
// asm memory clobbers keep the compiler from moving LDS accesses across the flag accesses.
// Layout (ints): ready[sub] at [sub] (tiles of the sub-chunk that are complete), done of consumer c
// of a sub-chunk at [4 + 4 sub + c] (tiles it has finished reading).
typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double8_t __attribute__((ext_vector_type(8)));
constexpr int kPcFlagWords = 4 + 4*kPcSub;
typedef __attribute__((address_space(3))) int lds_int_t;    // (a generic volatile access would be a flat_ one)
__device__ __forceinline__ void lds_publish(int* flag, int value, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) *(volatile lds_int_t*)(flag) = value;
}
__device__ __forceinline__ int lds_peek(const int* flag) {
    return __builtin_amdgcn_readfirstlane(*(const volatile lds_int_t*)(flag));
}
// doubles of the LDS body: the four sub-chunks' double-buffered tiles, or (epilogue) the six
// accumulator sets parked by the tree reduction, whichever is larger; the queue flags follow
constexpr int kPcPlanes = 15;   // tile planes of 64 doubles: 13 distinct q | psi.re | psi.im
__host__ __device__ constexpr int pc_lds_body_doubles(int d, int nc) {
    const int buffers = kPcSub*2*(kPcPlanes*64 + 2*nc*d*d*d);
    const int parked = 2*nc*d*d*64*2;
    return buffers > parked ? buffers : parked;
}

template <int D, int NC>
__global__ __launch_bounds__((NC + 1)*kPcSub*64, kPcSub) void ctrl_accumulate_pc_kernel(
    const double* __restrict__ omega, int W, const double* __restrict__ segtab,
    const cplx* __restrict__ ops, int G, int A, int chunk_len, cplx* __restrict__ Ypart) {
    constexpr int GS = kPcSub, NWS = NC + 1;
    constexpr int S = seg_stride(D), DD = D*D, NE = PcEntries<D>::count;
    static_assert(D == 4 && NE + 2 == kPcPlanes, "lane <-> (m, n, j) of the folded operand needs D^3 = 64");
    // One buffer of a sub-chunk, in doubles: kPcPlanes planes of 64 (the REAL factors q of the NE
    // distinct integral entries, psi.re, psi.im) | the folded operands W_a[m][n][j] (NC x 64 complex).
    // (With complex entries and Bbar, T in LDS the block held 119 KiB; a kernel of another pass that
    // needs more than what is left beside it waits for the whole accumulate kernel to retire,
    // tools/corun.hip and profiles/r02_q_*.)
    constexpr int QT = kPcPlanes*64, WS = 2*NC*D*DD;
    constexpr int BUFD = QT + WS;                     // doubles per buffer
    constexpr int SUBD = 2*BUFD;                      // per sub-chunk: 2 buffers
    constexpr int YSZ = DD*64;                        // cplx of one consumer's accumulators
    constexpr int BODY = pc_lds_body_doubles(D, NC);  // max(buffers, reduction slots), in doubles
    static_assert(GS*SUBD <= BODY && 2*NC*YSZ*2 <= BODY, "flags live behind both uses of the body");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

    const int lane = threadIdx.x & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sub = wave_all / NWS, wl = wave_all % NWS;
    const int pw = sub % NWS;                         // producer's position inside the group
    const bool producer = wl == pw;
    const int cidx = wl < pw ? wl : wl - 1;           // consumer index 0..NC-1
    const int alpha0 = blockIdx.y*NC;
    const int alpha = alpha0 + cidx;
    const bool active = !producer && alpha < A;
    const int n_alpha = min(NC, A - alpha0);
    double* ldsd = reinterpret_cast<double*>(lds_raw) + static_cast<size_t>(sub)*SUBD;
    // queue flags behind the body
    int* flags = reinterpret_cast<int*>(reinterpret_cast<double*>(lds_raw) + BODY);
    int* my_ready = flags + sub;
    int* my_done = flags + 4 + 4*sub;
    const int iw = blockIdx.x*64 + lane;
    const int sub_len = (chunk_len + GS - 1)/GS;
    const int g0 = blockIdx.z*chunk_len + sub*sub_len;
    const int g1 = min(min(G, static_cast<int>(blockIdx.z + 1)*chunk_len), g0 + sub_len);
    const int n_it = max(0, g1 - g0);                 // tiles of this sub-chunk
    int spin_limit = kPcSpinLimit;                    // 0 after a wait of this wavefront has run out
#ifdef FFK_PC_CLOCK
    unsigned long long* pc_tr = nullptr;
    {
        const unsigned bl = blockIdx.x + gridDim.x*(blockIdx.y + gridDim.y*blockIdx.z);
        if (bl < 1024 && wave_all < 16) {
            pc_tr = g_pc_trace + (static_cast<size_t>(bl)*16 + wave_all)*kPcTraceLen;
            if (lane == 0) {
                pc_tr[0] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |
                           (static_cast<unsigned long long>(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11))) << 32);
                pc_tr[1] = producer ? 0 : 1 + cidx;
                pc_tr[2] = __builtin_amdgcn_s_memtime();
                pc_tr[3] = __builtin_amdgcn_s_memrealtime();
            }
        }
    }
#endif

    // The q planes of one segment for the entries k0, k0 + kstep, ...: the table row is read through
    // SCALAR loads (its address is wave-uniform: dE, sin b, cos b feed the arithmetic as SGPR
    // operands, no LDS staging of rows), all entries on the straight-line path first -- thirteen
    // independent chains the hardware can interleave -- and the rare near-resonance lanes patched
    // afterwards under one branch.  (Round 3 walked the entries one by one with an LDS read and a
    // divergent branch each: a chain of ~6 k cycles per segment, which became the critical path once
    // the consumers' work shrank, profiles/r04_c_*.)
    auto generate = [&](double* buf, int g, double om, int k0, int kstep, bool with_psi) __attribute__((always_inline)) {
        const double* st = segtab + static_cast<size_t>(g)*S;
        double* qt = buf + lane;
        const double dtg = st[0];
        cplx ph;
        sincos_pi<false>(om*st[1], &ph.im, &ph.re);
        double sa, ca;
        sincos_pi<false>(0.5*(om*dtg), &sa, &ca);
        const PhasedFrequency pf = phased_frequency(om, dtg, ph, sa, ca);
        if (with_psi) {
            qt[NE*64] = pf.pr;
            qt[(NE + 1)*64] = pf.pi;
        }
        constexpr auto& sl = PcEntries<D>::slots;
        double qv[NE];
        unsigned near = 0u;
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            if (k % kstep != k0 % kstep && kstep != 1) continue;
            const double* r = st + seg_rec(sl.v[k]);
            const double x = om + r[0];
            qv[k] = fma(pf.sa2, r[2], pf.ca2*r[1])*rcp_fast(x);
            near |= (fabs(x) < pf.thr ? 1u : 0u) << k;
        }
        if (near != 0u) {
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                if (k % kstep != k0 % kstep && kstep != 1) continue;
                const double* r = st + seg_rec(sl.v[k]);
                if ((near >> k) & 1u) qv[k] = phased_q(pf, r[0], r[1], r[2]);
            }
        }
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            if (k % kstep != k0 % kstep && kstep != 1) continue;
            qt[k*64] = qv[k];
        }
    };

    if (producer) {
        // ---- producer: tile and folded operands of the next segment ------------------------------
        // Static issue priority: the producer's work is chains of dependent operations (argument
        // reduction -> polynomial -> reciprocal -> entries); at equal priority it competes with three
        // consumers' independent FMAs for every issue slot, finishes last and everybody waits for it.
        // One s_setprio before the loop, no per-segment flips (priority 1: the same; 0: 74 instead of
        // 67.8 us per step, profiles/r04_d_*).
        __builtin_amdgcn_s_setprio(3);
        const double om = omega[iw < W ? iw : W - 1];
        const int n_ops = (1 + n_alpha)*DD;           // <= 64: one element per lane
        struct Staged {
            cplx o;           // lane l: element l of [T | Bbar_0 | Bbar_1 ..] of the segment
            double sb, cb;    // sin b, cos b of entry (m, n) = l >> 2
        };
        auto load_ops = [&](int g) __attribute__((always_inline)) -> Staged {
            const cplx* src = ops + static_cast<size_t>(g)*(1 + A)*DD;
            const double* r = segtab + static_cast<size_t>(g)*S + seg_rec(lane >> 2);
            Staged t;
            t.o = lane < n_ops ? src[lane < DD ? lane : lane + alpha0*DD] : cplx{0.0, 0.0};
            t.sb = r[1];
            t.cb = r[2];
            return t;
        };
        // W_a[m][n][j] = Bbar_a[m][n] e^{i b_mn} T[n][j], (m, n, j) = this lane's index: the factors
        // come from the lanes that loaded them
        auto fold_ops = [&](double* buf, const Staged& t) __attribute__((always_inline)) {
            const int src_t = lane & (DD - 1);                        // T[n][j]
            const cplx tv = {__shfl(t.o.re, src_t, 64), __shfl(t.o.im, src_t, 64)};
            const cplx et = cmul(cplx{t.cb, t.sb}, tv);
            cplx* wb = reinterpret_cast<cplx*>(buf + QT);
#pragma unroll
            for (int a = 0; a < NC; ++a) {
                const int src_b = DD + a*DD + (lane >> 2);            // Bbar_a[m][n]
                const cplx bv = {__shfl(t.o.re, src_b, 64), __shfl(t.o.im, src_b, 64)};
                wb[a*D*DD + lane] = cmul(bv, et);
            }
        };
        static_assert((1 + NC)*DD <= 64, "one staging element per lane");
        if (lane < 4) {                  // tile 0 is complete after the barrier
            if (lane == 0) *(volatile lds_int_t*)(my_ready) = 1;
            else *(volatile lds_int_t*)(my_done + (lane - 1)) = 0;
        }
        // tile g0: the planes by all wavefronts of the sub-chunk together (every NWS-th each; left to
        // the producer alone its chains kept the consumers waiting), the operands folded here
        if (g0 < g1) {
            const Staged t0 = load_ops(g0);
            generate(ldsd, g0, om, wl, NWS, true);
            fold_ops(ldsd, t0);
        }
        __syncthreads();
        FFK_PC_STAMP(4);
        {
            // queue mode: tile it+1 goes into buffer (it+1) & 1, whose previous tenant was tile it-1
            for (int it = 0; it + 1 < n_it; ++it) {
                const int g = g0 + it;
                const int nb = (it + 1) & 1;
                const Staged t = load_ops(g + 1);
                if (it > 0) {
                    int spin = 0;
                    for (; spin < spin_limit; ++spin) {
                        int dn = lds_peek(my_done);
                        if (n_alpha > 1) dn = min(dn, lds_peek(my_done + 1));
                        if (n_alpha > 2) dn = min(dn, lds_peek(my_done + 2));
                        if (dn >= it) break;
                        __builtin_amdgcn_s_sleep(2);
                    }
                    if (spin == spin_limit && spin_limit != 0) {
                        pc_report_fault(kFaultPcProducerWait);
                        spin_limit = 0;
                    }
                    asm volatile("" ::: "memory");
                }
                FFK_PC_STEP_TOP(it);
                generate(ldsd + nb*BUFD, g + 1, om, 0, 1, true);
                fold_ops(ldsd + nb*BUFD, t);
#ifdef FFK_PC_FAULT_INJECT
                if (it < 1)
#endif
                lds_publish(my_ready, it + 2, lane);
                FFK_PC_STEP_DONE(it);
            }
        }
        FFK_PC_STAMP(5 + 2*kPcTraceSteps);
    } else {
        // ---- consumers: Y += psi T^dag Z,  Z[m][j] = sum_n q[m][n] W[m][n][j]  ------------------
        // q is REAL (E = psi e^{ib} q, the phase e^{ib} folded into W by the producer): the first
        // product costs 2 instead of 4 multiply-adds per term and Bbar o E is never formed:
        // 448 instead of 576 vector instructions per consumer and segment.
        // Accumulators and T_g live in FIXED registers, the segment's work is one generated asm
        // block (ctrl_pc_consumer.inc, tools/gen_pc_consumer.py): Y[i][j] = v[Y0 + 4 (4 i + j) : +3],
        // T_g[m][i] = s[36 + 16 m + 4 i : +3].  T_g comes through scalar loads (wave-uniform addresses
        // -> SGPR operands of v_fma_f64), issued for segment g+1 at the end of segment g.
        double4_t Y0 = 0.0, Y1 = 0.0, Y2 = 0.0, Y3 = 0.0, Y4 = 0.0, Y5 = 0.0, Y6 = 0.0, Y7 = 0.0;
        double8_t T0 = 0.0, T1 = 0.0, T2 = 0.0, T3 = 0.0;
        auto load_T = [&](int g) {
            const double8_t* tg = reinterpret_cast<const double8_t*>(ops + static_cast<size_t>(g)*(1 + A)*DD);
            T0 = tg[0];
            T1 = tg[1];
            T2 = tg[2];
            T3 = tg[3];
        };
        const unsigned lds_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(ldsd));
        auto contract = [&](int it) __attribute__((always_inline)) {
            const unsigned buf = lds_base + static_cast<unsigned>((it & 1)*BUFD*sizeof(double));
            const unsigned vq = buf + lane*8;                                   // plane[k][lane]
            const unsigned vw = buf + QT*8 + cidx*(D*DD*16);                    // W_cidx[m][n][j]
            asm volatile(FFK_PC_CONSUMER_ASM
                         : FFK_PC_CONSUMER_Y_OPERANDS(Y0, Y1, Y2, Y3, Y4, Y5, Y6, Y7)
                         : [vq] "v"(vq), [vw] "v"(vw), "{s[36:51]}"(T0), "{s[52:67]}"(T1), "{s[68:83]}"(T2),
                           "{s[84:99]}"(T3)
                         : FFK_PC_CONSUMER_CLOBBERS);
        };
        if (g0 < g1) {
            load_T(g0);
            const double om = omega[iw < W ? iw : W - 1];
            generate(ldsd, g0, om, wl, NWS, wl == 0);
        }
        __syncthreads();
        FFK_PC_STAMP(4);
        if (active) {
            int prio = 0;
            for (int it = 0; it < n_it; ++it) {
                // tile `it` published?  (usually yes: the producer works one tile ahead)
                int lead = 0;
                int spin = 0;
                for (; spin < spin_limit; ++spin) {
                    const int r0 = lds_peek(flags), r1 = lds_peek(flags + 1), r2 = lds_peek(flags + 2),
                              r3 = lds_peek(flags + 3);
                    const int mine = sub == 0 ? r0 : sub == 1 ? r1 : sub == 2 ? r2 : r3;
                    lead = max(max(r0, r1), max(r2, r3)) - (it + 1);
                    if (mine >= it + 1) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                if (spin == spin_limit && spin_limit != 0) {
                    pc_report_fault(kFaultPcConsumerWait);
                    spin_limit = 0;
                }
                asm volatile("" ::: "memory");
                // tiles the most advanced sub-chunk is ahead of this one -> issue priority
                lead = min(2, max(0, lead));
                if (lead != prio) {
                    prio = lead;
                    if (lead == 0) __builtin_amdgcn_s_setprio(0);
                    else if (lead == 1) __builtin_amdgcn_s_setprio(1);
                    else __builtin_amdgcn_s_setprio(2);
                }
                FFK_PC_STEP_TOP(it);
                contract(it);
                lds_publish(my_done + cidx, it + 1, lane);
                FFK_PC_STEP_DONE(it);
                // the scalar loads of the next T_g behind the flag (an opaque index: read-only
                // __restrict__ data could otherwise be fetched early)
                int gn = g0 + it + 1;
                asm volatile("" : "+s"(gn));
                if (it + 1 < n_it) load_T(gn);
            }
            __builtin_amdgcn_s_setprio(0);
        }
        FFK_PC_STAMP(5 + 2*kPcTraceSteps);
        __syncthreads();                              // every tile consumed: the buffers are free
        // The four sub-chunks' accumulators are summed through LDS (the tiles are dead now) as a
        // tree: sub-chunks 2, 3 -> 0, 1, then 1 -> 0, which writes the chunk's partial sum.
        cplx* red = reinterpret_cast<cplx*>(lds_raw);
        cplx Y[D][D];
        {
            const double4_t yv[8] = {Y0, Y1, Y2, Y3, Y4, Y5, Y6, Y7};
#pragma unroll
            for (int e = 0; e < DD; ++e) Y[e / D][e % D] = {yv[e >> 1][2*(e & 1)], yv[e >> 1][2*(e & 1) + 1]};
        }
        auto park = [&](int slot) {
            cplx* dst = red + static_cast<size_t>(slot)*YSZ + lane;
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) dst[(i*D + j)*64] = Y[i][j];
        };
        auto fetch_add = [&](int slot) {
            const cplx* srcy = red + static_cast<size_t>(slot)*YSZ + lane;
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    const cplx v = srcy[(i*D + j)*64];
                    Y[i][j].re += v.re;
                    Y[i][j].im += v.im;
                }
        };
        if (sub >= 2) park((sub - 2)*NC + cidx);
        __syncthreads();
        if (sub < 2) fetch_add(sub*NC + cidx);
        __syncthreads();
        if (sub == 1) park(cidx);
        __syncthreads();
        if (sub == 0) {
            fetch_add(cidx);
            if (active && iw < W) {
                cplx* out = Ypart + ((static_cast<size_t>(blockIdx.z)*A + alpha)*DD)*W + iw;
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) out[static_cast<size_t>(i*D + j)*W] = Y[i][j];
            }
        }
#ifdef FFK_PC_CLOCK
        if (pc_tr != nullptr && lane == 0) {
            pc_tr[6 + 2*kPcTraceSteps] = __builtin_amdgcn_s_memtime();
            pc_tr[7 + 2*kPcTraceSteps] = __builtin_amdgcn_s_memrealtime();
        }
#endif
        return;
    }
    // producers keep the consumers' epilogue barriers company
    __syncthreads();
    __syncthreads();
    __syncthreads();
    __syncthreads();
#ifdef FFK_PC_CLOCK
    if (pc_tr != nullptr && lane == 0) {
        pc_tr[6 + 2*kPcTraceSteps] = __builtin_amdgcn_s_memtime();
        pc_tr[7 + 2*kPcTraceSteps] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

template <int D, int NC>
hipError_t launch_pc(const double* omega, int W, const double* segtab, const cplx* ops, int G, int A,
                     int chunks, int chunk_len, cplx* Ypart, hipStream_t stream) {
    const int lds = pc_accumulate_lds_bytes(D, NC);
    (void)kernel_fault_word();
    auto kern = ctrl_accumulate_pc_kernel<D, NC>;
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (err != hipSuccess) return err;
    const dim3 grid((W + 63)/64, (A + NC - 1)/NC, chunks);
    hipLaunchKernelGGL(kern, grid, dim3((NC + 1)*kPcSub*64), lds, stream, omega, W, segtab, ops, G, A,
                       chunk_len, Ypart);
    return hipGetLastError();
}

}  // namespace

// called once, from kernel_fault_word(): the d = 4 kernel finds the word through a device global
hipError_t pc_bind_fault_word(int* device_pointer) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_pc_fault_word), &device_pointer, sizeof device_pointer);
}

bool pc_accumulate_supported(int d, int A) { return d == 4 && A >= 1; }
int pc_accumulate_ops_per_block(int A) { return A >= 3 ? 3 : A; }
int pc_accumulate_subchunks() { return kPcSub; }
int pc_accumulate_lds_bytes(int d, int nc) {
    return pc_lds_body_doubles(d, nc)*static_cast<int>(sizeof(double)) + kPcFlagWords*static_cast<int>(sizeof(int));
}

hipError_t launch_accumulate_pc(const double* omega, int W, const double* segtab, const cplx* ops,
                                int G, int d, int A, int nc, int chunks, int chunk_len, cplx* Ypart,
                                hipStream_t stream) {
    if (d != 4) return hipErrorInvalidValue;
    switch (nc) {
        case 1: return launch_pc<4, 1>(omega, W, segtab, ops, G, A, chunks, chunk_len, Ypart, stream);
        case 2: return launch_pc<4, 2>(omega, W, segtab, ops, G, A, chunks, chunk_len, Ypart, stream);
        case 3: return launch_pc<4, 3>(omega, W, segtab, ops, G, A, chunks, chunk_len, Ypart, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ffk

#ifdef FFK_PC_CLOCK
// (tuning build only, not in include/ffk.h) the last launch's per-wavefront timeline
extern "C" int ffk_debug_pc_trace(unsigned long long* out, int n_waves) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ffk::g_pc_trace),
                               sizeof(unsigned long long)*ffk::kPcTraceLen*static_cast<size_t>(n_waves)) != hipSuccess;
}
extern "C" int ffk_debug_pc_trace_len(void) { return ffk::kPcTraceLen; }
#endif
